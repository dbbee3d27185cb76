#!/bin/bash
# the round-6 profile set (run through gpurun from the repo root): bash profiles/collect_r6.sh
# -> gpurun_out/r6_*: kernel stats + FETCH / WRITE per kernel + the driver's bench line (collect.sh), the wave-1 probe of the
# speculative kernel with the leader's / workers' phase clocks, the one-rank exchange three times through both drivers, the
# scaled-wave budget, the RRT legs, configs[4]
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
bash profiles/collect.sh r6_bench
SFFGPU_PROFILE=1 PROBE_ITERS=8000,100000,600000 timeout 600 python3 profiles/spec_probe.py > $out/r6_spec_probe.txt 2>&1
for d in 1 2 3 4; do for s in 1 2; do echo "depth $d sets $s"; SFFGPU_SPEC_DEPTH=$d SFFGPU_SPEC_SETS=$s PROBE_ITERS=8000 timeout 300 python3 profiles/spec_probe.py 2>&1 | grep " spec "; done; done > $out/r6_spec_shapes.txt 2>&1
SFFGPU_SPEC_PIPE=0 PROBE_ITERS=8000,100000 timeout 300 python3 profiles/spec_probe.py 2>&1 | grep " spec " > $out/r6_spec_no_pipe.txt
bash profiles/r6_force_dist.sh > /dev/null 2>&1
timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 --force-dist --scaled-wave 65536 --cpu-iters 0 --no-sweep-micro --no-wave-sweep --no-extra-legs 2>/dev/null | tail -1 > $out/r6_wave_scaled_line.json
for m in star rrt multi; do SFFGPU_PROFILE=1 timeout 300 python3 profiles/rrt_probe.py 150000 $m 2>&1 | grep -E "run_wave|iterations_per_s" | tail -2; done > $out/r6_rrt_probe.txt
bash profiles/r6_rrt_repair.sh; bash profiles/r6_rrt_dry.sh
for m in star rrt multi; do SFFGPU_RRT_CHAIN=0 timeout 300 python3 profiles/rrt_probe.py 150000 $m 2>/dev/null | tail -1; done > $out/r6_rrt_probe_no_chain.jsonl
timeout 300 python3 profiles/c5_probe.py 2>/dev/null | tail -1 > $out/r6_c5_probe.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/r6_spec_trace -o t -- python3 $root/profiles/spec_probe.py > /dev/null 2>&1
cp $out/r6_spec_trace/t_kernel_stats.csv $out/r6_spec_kernel_stats.csv; rm -rf $out/r6_spec_trace
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/r6_rrt_trace -o t -- python3 $root/profiles/rrt_probe.py 150000 star > /dev/null 2>&1
cp $out/r6_rrt_trace/t_kernel_stats.csv $out/r6_rrt_star_kernel_stats.csv; rm -rf $out/r6_rrt_trace
echo done
