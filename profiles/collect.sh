#!/bin/bash
# Collects the profile set of one bench configuration on the GPU box (run through gpurun from the repo root):
#   bash profiles/collect.sh <tag>
# Writes gpurun_out/<tag>_{kernel_stats.csv,pmc_summary.json,bench_line.json}; copy them into profiles/.
# Kernel trace and each PMC counter are separate rocprofv3 passes (the pool refuses --pmc with trace domains
# other than kernel-trace, and FETCH_SIZE / WRITE_SIZE do not share a pass reliably).
set -u
tag=${1:-r1}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace -o t -- python3 $root/bench.py --cpu-iters 0 > $out/${tag}_trace.log 2>&1
cp $out/${tag}_trace/t_kernel_stats.csv $out/${tag}_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/${tag}_pmc_$c -o p -- python3 $root/bench.py --cpu-iters 0 > $out/${tag}_pmc_$c.log 2>&1
done
python3 $root/profiles/summarize_pmc.py $out/${tag}_pmc_FETCH_SIZE/p_counter_collection.csv $out/${tag}_pmc_WRITE_SIZE/p_counter_collection.csv > $out/${tag}_pmc_summary.json
cd $root && python3 bench.py > $out/${tag}_bench_full.log 2>&1
tail -1 $out/${tag}_bench_full.log > $out/${tag}_bench_line.json
rm -rf $out/${tag}_trace/t_kernel_trace.csv
echo done
