#!/usr/bin/env python3
"""Regenerate tests/golden/c5_full_run.json: the CPU oracle (PORTABLE trig) on BASELINE configs[4] run to its END -
building.obj, 6-DoF, 20 seeded roots, SFF* (optimize = true: choose-parent + rewire, src/forest.h:307-351), 2 M-node
budget.  The forest saturates (frontier empty, all trees connected: "solved") long before the budget, so the whole
job is pinned: fingerprint over every node, counters, checksums.  tests/test_gpu_parity.py replays it on the GPU.
C5_WAVE (default 4096) = frontier slots per wave."""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import common  # noqa: E402
import oracle_lib as O  # noqa: E402
from make_config_runs import summary  # noqa: E402

WAVE = int(os.environ.get("C5_WAVE", "4096"))

if __name__ == "__main__":
    sc = common.scenario("building")
    w = O.World(sc["env"], sc["robot"], O.TRIG_PORTABLE)
    roots = common.free_roots(w.collide, sc["limits"], 20, seed=1, dim=sc["dim"])
    f = O.Forest(w, roots, sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=sc["dim"],
                 optimize=True, max_iterations=2**31 - 1, node_budget=2000000, wave=WAVE, seed=1)
    t0 = time.time()
    f.run(0)
    out = summary(f)
    s = f.stats()
    out.update({"wave": WAVE, "waves": int(s["waves"]), "solved": int(s["solved"]), "frontier_size": int(s["frontier_size"]),
                "config": "building, 20 roots (seed 1), SFF* optimize, dist_tree %g, sampling_dist %g, budget 2000000, wave %d, seed 1"
                          % (sc["dist_tree"], sc["sampling_dist"], WAVE),
                "oracle_seconds": round(time.time() - t0, 1)})
    print(out)
    key = "c5_full_run.json" if WAVE == 4096 else "c5_full_run_w%d.json" % WAVE
    json.dump(out, open(os.path.join(HERE, key), "w"), indent=1)
