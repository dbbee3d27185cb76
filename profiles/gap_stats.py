#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV, grouped by (previous, next) kernel.
usage: gap_stats.py t_kernel_trace.csv"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("(")[0].replace("sffk::", "")
gaps = collections.defaultdict(list)
busy = 0
for a, b in zip(rows, rows[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    gaps[(short(a["Kernel_Name"]), short(b["Kernel_Name"]))].append(g)
for r in rows:
    busy += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print("kernels %d busy %.1f ms span %.1f ms" % (len(rows), busy / 1e6, span / 1e6))
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    if len(v) < 20:
        continue
    v2 = sorted(v)
    print("%-22s -> %-22s n %5d  median %7.2f us  mean %7.2f us  total %7.2f ms" % (k[0], k[1], len(v), v2[len(v2) // 2] / 1e3, sum(v) / len(v) / 1e3, sum(v) / 1e6))
