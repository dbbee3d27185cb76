#!/bin/bash
# kernel averages of an arbitrary python command of this repo (run through gpurun from the repo root):
#   bash profiles/trace_cmd.sh <tag> <script.py> [args]      -> gpurun_out/<tag>_kernel_stats.csv + a printed top list
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${tag}_trace -o t -- python3 $root/$@ > $root/gpurun_out/${tag}_trace.log 2>&1
cp $root/gpurun_out/${tag}_trace/t_kernel_stats.csv $root/gpurun_out/${tag}_kernel_stats.csv
rm -rf $root/gpurun_out/${tag}_trace
tail -1 $root/gpurun_out/${tag}_trace.log | cut -c1-400
python3 - <<PY
import csv
for r in list(csv.DictReader(open('$root/gpurun_out/${tag}_kernel_stats.csv')))[:${TOP:-24}]:
    print(r['Name'][:44].ljust(44), r['Calls'].rjust(7), ('%.1f' % (float(r['AverageNs'])/1e3)).rjust(8), ('%.1f' % (float(r['MaxNs'])/1e3)).rjust(8), r['Percentage'])
PY
