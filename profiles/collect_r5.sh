#!/bin/bash
# the round-5 profile set (run through gpurun from the repo root): bash profiles/collect_r5.sh
# -> gpurun_out/r5_*: kernel stats + FETCH / WRITE per kernel + bench lines of the driver's command, SQ counters, L2 /
# instruction counters of the query kernel (with and without the wave's spatial order), the stand-alone sweep's trace and
# counters, configs[4] kernel stats, per-round multi-GPU budget, phase clocks, the small-wave / RRT / priority probes
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
bash profiles/collect.sh r5_bench
bash profiles/collect_sq.sh r5
bash profiles/collect_counters.sh r5_query "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "TCC_EA0_RDREQ_LEVEL TCC_EA0_RDREQ_DRAM TCC_TAG_STALL_sum TCC_BUSY_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM" "FETCH_SIZE"
SFFGPU_NO_ORDER=1 bash profiles/collect_counters.sh r5_query_no_order "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE"
bash profiles/collect_sweep.sh r5 > /dev/null 2>&1
bash profiles/r5_c5_trace.sh r5_c5 > $out/r5_c5_trace_top.txt 2>&1
timeout 300 python3 profiles/c5_probe.py 2>/dev/null | tail -1 > $out/r5_c5_probe.json
bash profiles/r5_tail_ab.sh > $out/r5_tail_ab.txt 2>&1
SFFGPU_PROFILE=1 timeout 300 python3 profiles/c5_probe.py 2>&1 | grep -E "k_star_tail|k_star_knn" > $out/r5_c5_star_clocks.txt
bash profiles/r5_star_chain.sh > /dev/null 2>&1
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --force-dist --cpu-iters 0 --no-sweep-micro --no-wave-sweep --no-extra-legs 2>/dev/null | tail -1 > $out/r5_force_dist_line.json
SFFGPU_PROFILE=1 timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-iters 0 --no-sweep-micro --no-wave-sweep --no-extra-legs 2>&1 | grep -E "^\[sffgpu" | tail -8 > $out/r5_phase_clocks.txt
timeout 600 python3 profiles/small_wave_probe.py > $out/r5_small_waves.txt 2>&1
for m in star rrt multi; do timeout 300 python3 profiles/rrt_probe.py 150000 $m 2>/dev/null | tail -1; done > $out/r5_rrt_probe.jsonl
timeout 300 python3 profiles/priority_probe.py 300000 1024 8192 16384 2>/dev/null | grep -E "^\{" > $out/r5_priority_probe.jsonl
SFFGPU_NO_ORDER=1 bash profiles/bench_lean.sh r5_no_order > $out/r5_bench_lean_no_order.txt 2>&1
bash profiles/bench_lean.sh r5_order > $out/r5_bench_lean_order.txt 2>&1
echo done
