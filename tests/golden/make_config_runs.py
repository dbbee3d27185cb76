#!/usr/bin/env python3
"""Regenerate tests/golden/config_runs.json: CPU-oracle runs (PORTABLE trig) of the other BASELINE.json
configurations at sizes the oracle finishes in minutes - configs[0] (test_2D: dense.tri, 2-D, 3 roots, 10 k-node
budget) and configs[1] (triang.obj, 6-DoF, 5 roots, 100 k-node budget)
in full, configs[4] (building.obj, 20 roots, SFF* with rewire) cut to a 150 k-node budget - summarised like
full_size_run.json.  tests/test_gpu_parity.py::test_baseline_configs_equal_the_oracle replays them on the GPU."""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import common  # noqa: E402
import oracle_lib as O  # noqa: E402

CONFIGS = {
    # name: (scenario, roots, optimize, budget, wave, waves)
    "configs[0] dense2d 3 roots 10k": ("dense2d", 3, False, 10000, 256, 0),
    "configs[1] triang 5 roots 100k": ("triang", 5, False, 100000, 4096, 0),
    "configs[4] building 20 roots SFF* 150k": ("building", 20, True, 150000, 4096, 0),
}


def summary(f):
    s, n = f.stats(), f.nodes()
    return {"fingerprint": "%016x" % f.fingerprint(), "n_nodes": int(s["n_nodes"]), "iterations": int(s["iterations"]),
            "collide_calls": int(s["collide_calls"]), "path_free_calls": int(s["path_free_calls"]),
            "nn_queries": int(s["nn_queries"]), "n_borders": int(s["n_borders"]),
            "parent_sum": int(n["parent"].astype(np.int64).sum()), "cost_sum": float(n["cost"].sum()).hex()}


if __name__ == "__main__":
    out = {}
    for key, (name, nroots, opt, budget, wave, waves) in CONFIGS.items():
        sc = common.scenario(name)
        w = O.World(sc["env"], sc["robot"], O.TRIG_PORTABLE)
        roots = common.free_roots(w.collide, sc["limits"], nroots, seed=1, dim=sc["dim"])
        f = O.Forest(w, roots, sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=sc["dim"],
                     optimize=opt, max_iterations=2**31 - 1, node_budget=budget, wave=wave, seed=1)
        t0 = time.time()
        f.run(waves)
        out[key] = summary(f)
        out[key]["oracle_seconds"] = round(time.time() - t0, 1)
        print(key, out[key], flush=True)
    json.dump(out, open(os.path.join(HERE, "config_runs.json"), "w"), indent=1)
