"""Per-kernel sums of rocprofv3 --pmc passes (counter_collection.csv files) -> JSON on stdout.
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; bench.py applies the gfx950 correction
(FETCH_SIZE x 2, MI355X_MICROARCH.md) when it quotes `roofline.traffic`.  --bench-args records the bench
arguments of the profiled command: bench.py only quotes the traffic when its own arguments are the same."""
import argparse
import collections
import csv
import json
import sys

ap = argparse.ArgumentParser()
ap.add_argument("--bench-args", default="")
ap.add_argument("--query-kernel", default="sffk::k_query_block")
ap.add_argument("--kernel-stats", default="", help="rocprofv3 --kernel-trace --stats summary (t_kernel_stats.csv) of the same command: per-kernel average durations")
ap.add_argument("files", nargs="+")
a = ap.parse_args()

bp = argparse.ArgumentParser()
bp.add_argument("--gpus", type=int, default=1)
bp.add_argument("--steps", type=int, default=20)
bp.add_argument("--warmup", type=int, default=5)
bp.add_argument("--wave", type=int, default=16384)
bp.add_argument("--waves-per-step", type=int, default=9)
bp.add_argument("--budget", type=int, default=1000000)
bp.add_argument("--seed", type=int, default=1)
b, _ = bp.parse_known_args(a.bench_args.split())
out = {"bench_args": {"steps": b.steps, "warmup": b.warmup, "wave": b.wave, "waves_per_step": b.waves_per_step,
                      "budget": b.budget, "seed": b.seed, "gpus": b.gpus},
       "query_kernel": a.query_kernel}
for path in a.files:
    acc = collections.defaultdict(lambda: [0, 0.0])
    name = None
    for r in csv.DictReader(open(path)):
        name = r["Counter_Name"]
        k = r["Kernel_Name"].split("(")[0]
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
    out[name] = {k: {"launches": v[0], "sum_KiB": v[1], "avg_KiB_per_launch": v[1] / max(1, v[0])} for k, v in acc.items()}
if a.kernel_stats:
    out["kernel_trace_avg_us"] = {r["Name"].split("(")[0].replace("void ", ""): float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(a.kernel_stats))}
json.dump(out, sys.stdout, indent=1)
