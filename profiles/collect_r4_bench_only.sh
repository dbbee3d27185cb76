#!/bin/bash
# the bench-job part of collect_r4.sh (kernel stats, FETCH / WRITE, SQ and L2 counters, full line, phase clocks, force-dist)
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
bash profiles/collect.sh r4_bench
bash profiles/collect_sq.sh r4
bash profiles/collect_counters.sh r4_query "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "TCC_EA0_RDREQ_LEVEL TCC_EA0_RDREQ_DRAM TCC_TAG_STALL_sum TCC_BUSY_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM"
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --force-dist --cpu-iters 0 --no-sweep-micro --no-wave-sweep --no-extra-legs 2>/dev/null | tail -1 > $out/r4_force_dist_line.json
SFFGPU_PROFILE=1 timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-iters 0 --no-sweep-micro --no-wave-sweep --no-extra-legs 2>&1 | grep -E "^\[sffgpu" | tail -6 > $out/r4_phase_clocks.txt
echo done
