"""Per-kernel sums of rocprofv3 --pmc passes (counter_collection.csv files) -> JSON on stdout.
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; bench.py applies the gfx950 correction
(FETCH_SIZE x 2, MI355X_MICROARCH.md) when it quotes `roofline.traffic`."""
import collections
import csv
import json
import sys

out = {}
for path in sys.argv[1:]:
    acc = collections.defaultdict(lambda: [0, 0.0])
    name = None
    for r in csv.DictReader(open(path)):
        name = r["Counter_Name"]
        k = r["Kernel_Name"].split("(")[0]
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
    out[name] = {k: {"launches": v[0], "sum_KiB": v[1], "avg_KiB_per_launch": v[1] / max(1, v[0])} for k, v in acc.items()}
json.dump(out, sys.stdout, indent=1)
