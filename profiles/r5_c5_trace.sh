#!/bin/bash
# kernel stats of configs[4] (building, SFF*, waves of 8192 slots; eager launches: rocprofv3 crashes on replays of the long graph)
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out; mkdir -p $out
tag=${1:-r5_c5}
cd /tmp && export TMPDIR=/tmp
SFFGPU_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace -o t -- python3 $root/profiles/c5_probe.py 2000000 8192 > $out/${tag}_trace.log 2>&1
cp $out/${tag}_trace/t_kernel_stats.csv $out/${tag}_kernel_stats.csv
rm -rf $out/${tag}_trace
python3 - $out/${tag}_kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:20]:
    print("%-34s calls %6s avg %8.2f us  min %7.2f max %8.2f  %5.1f %%" % (r["Name"].split("(")[0][:34], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3, float(r["Percentage"])))
PY
