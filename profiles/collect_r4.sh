#!/bin/bash
# the round-4 profile set (run through gpurun from the repo root): bash profiles/collect_r4.sh
# -> gpurun_out/r4_*: kernel stats + FETCH / WRITE per kernel + bench lines of the driver's command, SQ counters, memory-side
# counters of the query kernel, configs[4] and priority-mode kernel stats, small-wave probe, per-round multi-GPU budget,
# RRT legs (kernel stats, per-call k-nearest split), the k-NN and heap micro-benchmarks
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
bash profiles/collect.sh r4_bench
bash profiles/collect_sq.sh r4
bash profiles/collect_counters.sh r4_query "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "TCC_EA0_RDREQ_LEVEL TCC_EA0_RDREQ_DRAM TCC_TAG_STALL_sum TCC_BUSY_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM"
SFFGPU_NO_GRAPH=1 bash profiles/trace_cmd.sh r4_c5 profiles/c5_probe.py > $out/r4_c5_trace_top.txt 2>&1
timeout 300 python3 profiles/c5_probe.py 2>/dev/null | tail -1 > $out/r4_c5_probe.json
bash profiles/trace_cmd.sh r4_priority profiles/priority_probe.py 300000 8192 > $out/r4_priority_trace_top.txt 2>&1
timeout 300 python3 profiles/priority_probe.py 300000 1024 8192 16384 2>/dev/null | grep -E "^\{" > $out/r4_priority_probe.jsonl
SFFGPU_PRIO_DEVICE=0 timeout 600 python3 profiles/priority_probe.py 100000 1024 8192 2>/dev/null | grep -E "^\{" > $out/r4_priority_probe_host_engine.jsonl
timeout 600 python3 profiles/small_wave_probe.py > $out/r4_small_waves.txt 2>&1
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --force-dist --cpu-iters 0 --no-sweep-micro --no-wave-sweep --no-extra-legs 2>/dev/null | tail -1 > $out/r4_force_dist_line.json
SFFGPU_PROFILE=1 timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-iters 0 --no-sweep-micro --no-wave-sweep --no-extra-legs 2>&1 | grep -E "^\[sffgpu" | tail -6 > $out/r4_phase_clocks.txt
bash profiles/trace_cmd.sh r4_rrt_star profiles/rrt_probe.py 150000 star > $out/r4_rrt_star_trace_top.txt 2>&1
for m in star rrt multi; do timeout 300 python3 profiles/rrt_probe.py 150000 $m 2>/dev/null | tail -1; done > $out/r4_rrt_probe.jsonl
bash profiles/rrt_knn_split.sh 60000 > $out/r4_rrt_knn_split.txt 2>&1
timeout 900 python3 profiles/knn_microbench.py 2>/dev/null | grep -E "^\{" > $out/r4_knn_microbench.jsonl
( cd profiles/micro && [ -x ./heap_microbench ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w -I../../space_filling_forest_star_amd/csrc -I../../include -I../../include/sff heap_microbench.hip -o heap_microbench; ./heap_microbench 90 30000 90; ./heap_microbench 90 300000 90 ) > $out/r4_heap_microbench.txt 2>&1
echo done
