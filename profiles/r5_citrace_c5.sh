#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; out=gpurun_out; mkdir -p $out; rm -f $out/r5_citrace_c5.txt
for L in 200 400; do
echo "== k_collide_items launch $L of configs[4]" >> $out/r5_citrace_c5.txt
SFFGPU_LIB=libsffgpu_ci$L.so SFFGPU_PROFILE=1 timeout 300 python3 profiles/c5_probe.py 2000000 8192 2>&1 | grep -A17 "k_collide_items trace" | head -18 >> $out/r5_citrace_c5.txt
done
cat $out/r5_citrace_c5.txt
