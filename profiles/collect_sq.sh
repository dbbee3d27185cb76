#!/bin/bash
# SQ counters of the collision kernels of the driver's bench command (one rocprofv3 --pmc pass, kernel-trace only):
#   bash profiles/collect_sq.sh <tag>   ->  gpurun_out/<tag>_sq_summary.json   (copy into profiles/)
# SQ_* cycle counters tick in quad-cycles (MI355X_MICROARCH.md); WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES.
set -u
tag=${1:-r2}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT \
  --output-format csv -d $out/${tag}_sq -o p -- python3 $root/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-iters 0 --no-sweep-micro --no-wave-sweep --no-extra-legs > $out/${tag}_sq.log 2>&1
python3 - "$out/${tag}_sq/p_counter_collection.csv" > $out/${tag}_sq_summary.json <<'PY'
import collections, csv, json, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES":
        calls[k] += 1
out = {}
for k, c in acc.items():
    n = max(1, calls[k])
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    out[k] = {"launches": calls[k], "per_launch": {m: v / n for m, v in c.items()},
              "wait_any_frac": c.get("SQ_WAIT_ANY", 0) / wc if wc else None,
              "issue_stall_frac": c.get("SQ_WAIT_INST_ANY", 0) / wc if wc else None,
              "active_frac": c.get("SQ_ACTIVE_INST_ANY", 0) / wc if wc else None,
              "valu_frac": c.get("SQ_ACTIVE_INST_VALU", 0) / wc if wc else None,
              "lds_conflict_frac": c.get("SQ_LDS_BANK_CONFLICT", 0) / wc if wc else None}
json.dump(out, sys.stdout, indent=1)
PY
rm -rf $out/${tag}_sq
echo done
