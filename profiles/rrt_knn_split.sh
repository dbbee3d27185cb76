#!/bin/bash
# RRT* leg: per-call durations of k_knn_grid split by call parity (the nearest-node query, k = 1, and the k_max-nearest
# query alternate inside a speculative wave) - run through gpurun: bash profiles/rrt_knn_split.sh [iterations]
root=${GRAFT_REPO_ROOT:-$(pwd)}
it=${1:-60000}
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/rrt_split_trace -o t -- python3 $root/profiles/rrt_probe.py $it > $root/gpurun_out/rrt_split.log 2>&1
tail -1 $root/gpurun_out/rrt_split.log | cut -c1-300
python3 - <<PY
import csv, glob
f = glob.glob('$root/gpurun_out/rrt_split_trace/*kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
import collections
d = collections.defaultdict(list)
seq = []
for r in rows:
    n = r['Kernel_Name'].split('(')[0]
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    d[n].append(dur)
    if 'k_knn_grid' in n: seq.append((dur, int(r.get('Grid_Size', r.get('Grid_Size_X', 0)) or 0)))
tot = sum(sum(v) for v in d.values())
print('total kernel ms', tot / 1e3)
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:10]:
    print(n[:50].ljust(50), len(v), '%.1f us avg' % (sum(v) / len(v)), '%.1f ms' % (sum(v) / 1e3))
# knn calls: pairs
a = seq[0::2]; b = seq[1::2]
for name, s in (('even calls', a), ('odd calls', b)):
    if s: print(name, len(s), 'avg %.1f us' % (sum(x[0] for x in s) / len(s)), 'max %.1f' % max(x[0] for x in s), 'avg grid', sum(x[1] for x in s) / len(s))
# by decile of the run
n = len(seq)
for q in range(10):
    s = seq[q * n // 10:(q + 1) * n // 10]
    ev = [x[0] for i, x in enumerate(s) if i % 2 == 0]; od = [x[0] for i, x in enumerate(s) if i % 2 == 1]
    print('decile', q, 'even %.0f us' % (sum(ev) / max(1, len(ev))), 'odd %.0f us' % (sum(od) / max(1, len(od))), 'grid', sum(x[1] for x in s) / max(1, len(s)))
PY
rm -rf $root/gpurun_out/rrt_split_trace
