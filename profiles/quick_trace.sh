#!/bin/bash
# quick look: kernel averages of the default bench job + the plain bench line (run through gpurun from the repo root)
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/q_trace -o t -- python3 $root/bench.py --cpu-iters 0 --no-sweep-micro --no-wave-sweep --no-extra-legs > /dev/null 2>&1
rm -f $root/gpurun_out/q_trace/t_kernel_trace.csv
python3 - <<PY
import csv
for r in list(csv.DictReader(open('$root/gpurun_out/q_trace/t_kernel_stats.csv')))[:12]:
    print(r['Name'][:36].ljust(36), r['Calls'], '%.1f' % (float(r['AverageNs'])/1e3), r['Percentage'])
PY
cd $root; timeout 200 python3 bench.py --cpu-iters 0 --no-sweep-micro --no-wave-sweep --no-extra-legs 2>&1 | tail -1 | cut -c1-200
