#!/bin/bash
# rocprofv3 evidence for the bandwidth-shaped kernel (k_sweep on 16 M nodes, bench.py's sweep_kernel_roofline leg):
#   bash profiles/collect_sweep.sh <tag>   ->  gpurun_out/<tag>_sweep_kernel_stats.csv, <tag>_sweep_pmc_summary.json, <tag>_sweep_line.json
set -u
tag=${1:-r5}
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_sweep_trace -o t -- python3 $root/profiles/sweep_leg.py > $out/${tag}_sweep_trace.log 2>&1
cp $out/${tag}_sweep_trace/t_kernel_stats.csv $out/${tag}_sweep_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $out/${tag}_sweep_pmc_$c -o p -- python3 $root/profiles/sweep_leg.py > $out/${tag}_sweep_pmc_$c.log 2>&1
done
python3 - $out/${tag}_sweep_pmc_FETCH_SIZE/p_counter_collection.csv $out/${tag}_sweep_pmc_WRITE_SIZE/p_counter_collection.csv $out/${tag}_sweep_kernel_stats.csv > $out/${tag}_sweep_pmc_summary.json <<'PY'
import collections, csv, json, sys
out = {"nodes": 16000000, "kernel": "sffk::k_sweep"}
for path in sys.argv[1:3]:
    acc = [0, 0.0]; name = None
    for r in csv.DictReader(open(path)):
        if r["Kernel_Name"].split("(")[0] != "sffk::k_sweep": continue
        name = r["Counter_Name"]; acc[0] += 1; acc[1] += float(r["Counter_Value"])
    out[name] = {"launches": acc[0], "avg_KiB_per_launch": acc[1] / max(1, acc[0])}
for r in csv.DictReader(open(sys.argv[3])):
    if r["Name"].split("(")[0] == "sffk::k_sweep":
        out["kernel_trace"] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3}
json.dump(out, sys.stdout, indent=1)
PY
cd $root && python3 profiles/sweep_leg.py > $out/${tag}_sweep_line.json 2>/dev/null
rm -rf $out/${tag}_sweep_trace $out/${tag}_sweep_pmc_FETCH_SIZE $out/${tag}_sweep_pmc_WRITE_SIZE
cat $out/${tag}_sweep_pmc_summary.json; cat $out/${tag}_sweep_line.json
