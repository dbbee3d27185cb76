#!/bin/bash
# VGPRs / spills / scratch / LDS of the kernels in an object file: bash profiles/kernel_regs.sh <file.o> [grep pattern]
B=/opt/rocm/lib/llvm/bin
$B/llvm-objcopy -O binary --only-section=.hip_fatbin $1 /tmp/_fat.bin || exit 1
$B/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=/tmp/_fat.bin --output=/tmp/_k.co --unbundle || exit 1
$B/llvm-readelf --notes /tmp/_k.co | awk '/\.name:/{n=$2} /\.vgpr_count:/{v=$2} /\.vgpr_spill_count:/{s=$2} /\.private_segment_fixed_size:/{p=$2} /\.group_segment_fixed_size:/{l=$2} /\.wavefront_size:/{print n, "vgpr", v, "spill", s, "scratch", p, "lds", l}' | grep -E "${2:-.}"
