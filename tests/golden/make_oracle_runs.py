#!/usr/bin/env python3
"""Regenerate tests/golden/oracle_runs.json: summary numbers of small oracle runs (PORTABLE trig),
used as a regression fixture for the oracle itself (tests/test_oracle_cpu.py)."""
import json, os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as O
from test_oracle_cpu import RUNS, small_run
out = {}
for name, wave, opt, rrt in RUNS:
    key = "%s/w%d/%s/%s" % (name, wave, "star" if opt else "plain", "rrt" if rrt else "sff")
    r = small_run(name, wave, opt, O.TRIG_PORTABLE, iters=1200, seed=13, rrt=rrt)
    s = r.stats(); n = r.nodes()
    out[key] = {"n_nodes": int(s["n_nodes"]), "iterations": int(s["iterations"]), "collide_calls": int(s["collide_calls"]),
                "parent_sum": int(n["parent"].astype(np.int64).sum()), "cost_sum": float(n["cost"].sum()).hex()}
    print(key, out[key])
json.dump(out, open(os.path.join(HERE, "oracle_runs.json"), "w"), indent=1)
