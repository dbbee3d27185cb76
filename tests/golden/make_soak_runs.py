#!/usr/bin/env python3
"""Regenerate tests/golden/soak_runs.json: more seeds and wave sizes of mid-size CPU-oracle runs (PORTABLE trig),
summarised like config_runs.json - a wider net for rare divergences (clearance-bit borders, list overflows,
wave-mate ordering).  tests/test_gpu_parity.py::test_more_seeds_equal_the_oracle replays them on the GPU."""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import common  # noqa: E402
import oracle_lib as O  # noqa: E402
from make_config_runs import summary  # noqa: E402

RUNS = [  # (scenario, roots, optimize, budget, wave, seed)
    ("triang", 5, False, 60000, 4096, 2), ("triang", 5, False, 60000, 1024, 3), ("triang", 6, True, 25000, 2048, 4),
    ("building", 12, False, 60000, 8192, 5), ("building", 8, True, 20000, 512, 6),
    ("dense3d", 10, False, 60000, 8192, 7), ("dense3d", 6, True, 20000, 4096, 8), ("dense2d", 3, False, 10000, 64, 9),
]

if __name__ == "__main__":
    out = {}
    for name, nroots, opt, budget, wave, seed in RUNS:
        sc = common.scenario(name)
        w = O.World(sc["env"], sc["robot"], O.TRIG_PORTABLE)
        roots = common.free_roots(w.collide, sc["limits"], nroots, seed=seed, dim=sc["dim"])
        f = O.Forest(w, roots, sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=sc["dim"],
                     optimize=opt, max_iterations=2**31 - 1, node_budget=budget, wave=wave, seed=seed)
        t0 = time.time()
        f.run(0)
        key = "%s/%d roots/%s/budget %d/wave %d/seed %d" % (name, nroots, "star" if opt else "plain", budget, wave, seed)
        out[key] = summary(f)
        out[key]["oracle_seconds"] = round(time.time() - t0, 1)
        print(key, out[key]["n_nodes"], out[key]["iterations"], out[key]["oracle_seconds"], flush=True)
    json.dump(out, open(os.path.join(HERE, "soak_runs.json"), "w"), indent=1)
