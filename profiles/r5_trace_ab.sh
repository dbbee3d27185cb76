#!/bin/bash
# kernel stats of the lean bench command with / without a toggle: bash profiles/r5_trace_ab.sh <tag> [ENV=1]
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out; mkdir -p $out
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace -o t -- python3 $root/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-iters 0 --no-sweep-micro --no-wave-sweep --no-extra-legs > $out/${tag}_trace.log 2>&1
cp $out/${tag}_trace/t_kernel_stats.csv $out/${tag}_kernel_stats.csv
rm -rf $out/${tag}_trace
python3 - $out/${tag}_kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-34s calls %6s avg %8.2f us  min %7.2f max %8.2f  %5.1f %%" % (r["Name"].split("(")[0][:34], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3, float(r["Percentage"])))
PY
