#!/bin/bash
# Collects the profile set of the driver's bench command on the GPU box (run through gpurun from the repo root):
#   bash profiles/collect.sh <tag> [extra bench args]
# Writes gpurun_out/<tag>_{kernel_stats.csv,pmc_summary.json,bench_line.json}; copy them into profiles/.
# The profiled command is the driver's `bench.py --gpus 1 --steps 20 --warmup 5` without the legs that run AFTER
# the timed region (CPU baseline, stand-alone sweep, small-wave legs: they launch the same kernels on other
# workloads and would pollute the per-kernel averages).  Kernel trace and each PMC counter are separate rocprofv3
# passes (the pool refuses --pmc with trace domains other than kernel-trace, and FETCH_SIZE / WRITE_SIZE do not
# share a pass reliably).
set -u
tag=${1:-r2}
shift || true
extra="$*"
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
BENCH="$root/bench.py --gpus 1 --steps 20 --warmup 5 $extra"
LEAN="--cpu-iters 0 --no-sweep-micro --no-wave-sweep --no-extra-legs"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace -o t -- python3 $BENCH $LEAN > $out/${tag}_trace.log 2>&1
cp $out/${tag}_trace/t_kernel_stats.csv $out/${tag}_kernel_stats.csv
grep -E "^\{\"metric\"" $out/${tag}_trace.log | tail -1 > $out/${tag}_bench_line_under_trace.json
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $out/${tag}_pmc_$c -o p -- python3 $BENCH $LEAN > $out/${tag}_pmc_$c.log 2>&1
done
python3 $root/profiles/summarize_pmc.py --kernel-stats $out/${tag}_kernel_stats.csv --bench-args "--steps 20 --warmup 5 $extra" \
  $out/${tag}_pmc_FETCH_SIZE/p_counter_collection.csv $out/${tag}_pmc_WRITE_SIZE/p_counter_collection.csv > $out/${tag}_pmc_summary.json
cd $root && timeout 600 python3 $BENCH > $out/${tag}_bench_full.log 2>&1
tail -1 $out/${tag}_bench_full.log > $out/${tag}_bench_line.json
rm -rf $out/${tag}_trace/t_kernel_trace.csv $out/${tag}_pmc_FETCH_SIZE $out/${tag}_pmc_WRITE_SIZE
echo done
