#!/bin/bash
# per-kernel averages of arbitrary PMC counters of the lean bench command, one rocprofv3 pass per quoted group:
#   bash profiles/collect_counters.sh <tag> "CNT_A CNT_B" "CNT_C ..."   ->  gpurun_out/<tag>_counters.json
set -u
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
k=0
files=""
for grp in "$@"; do
  k=$((k+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $out/${tag}_cnt$k -o p -- python3 $root/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-iters 0 --no-sweep-micro --no-wave-sweep --no-extra-legs > $out/${tag}_cnt$k.log 2>&1
  files="$files $out/${tag}_cnt$k/p_counter_collection.csv"
done
python3 - $files > $out/${tag}_counters.json <<'PY'
import collections, csv, json, sys, os
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.Counter())
for path in sys.argv[1:]:
    if not os.path.exists(path):
        continue
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k][r["Counter_Name"]] += 1
out = {k: {m: v / max(1, calls[k][m]) for m, v in c.items()} for k, c in acc.items()}
json.dump(out, sys.stdout, indent=1)
PY
for i in $(seq 1 $k); do rm -rf $out/${tag}_cnt$i; done
echo done
