#!/usr/bin/env python3
"""Regenerate tests/golden/full_size_run.json: the CPU oracle (PORTABLE trig) on the headline configuration of
bench.py - dense_3D, 6-DoF, 10 seeded roots, circum 14 / dtree 18, 1 M-node budget, waves of 8192 slots, seed 1,
3 + 255 waves - summarised as fingerprint, counters and checksums.  The GPU run of the same configuration must
reproduce every number (tests/test_gpu_parity.py::test_full_size_headline_run_equals_the_oracle).
Takes ~10-40 minutes of one CPU core.  The roots are the ones bench.py draws (first 10 collision-free uniform
points of RandomState(1)), here tested with the oracle's collide."""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import common  # noqa: E402
import oracle_lib as O  # noqa: E402

WAVES = int(os.environ.get("FULL_SIZE_WAVES", "258"))     # 0 = until the node budget stops the run
WAVE = int(os.environ.get("FULL_SIZE_WAVE", "8192"))       # slots per wave; != 8192 writes full_size_run_w<WAVE>.json
# FULL_SIZE_TRIG=libm: the REFERENCE-PINNED sampling arithmetic (glibc cos / sin / acos, tests/golden/ref_primitives.json)
# instead of the kernels' portable trig; writes ..._libm.json (replayed on the GPU with libm_sampling = 1)
LIBM = os.environ.get("FULL_SIZE_TRIG", "portable") == "libm"
sc = common.scenario("dense3d")
w = O.World(sc["env"], sc["robot"], O.TRIG_PORTABLE)
roots = common.free_roots(w.collide, sc["limits"], 10, seed=1)
f = O.Forest(w, roots, sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=6,
             max_iterations=2**31 - 1, node_budget=1000000, wave=WAVE, seed=1,
             trig=O.TRIG_LIBM if LIBM else O.TRIG_PORTABLE)
t0 = time.time()
f.run(WAVES)
s = f.stats()
n = f.nodes()
out = {"config": "dense3d, 10 roots (seed 1), dist_tree %g, sampling_dist %g, budget 1000000, wave %d, seed 1, %d waves"
                 % (sc["dist_tree"], sc["sampling_dist"], WAVE, int(s["waves"])),
       "waves": int(s["waves"]), "wave": WAVE, "fingerprint": "%016x" % f.fingerprint(),
       "n_nodes": int(s["n_nodes"]), "iterations": int(s["iterations"]), "collide_calls": int(s["collide_calls"]),
       "path_free_calls": int(s["path_free_calls"]), "nn_queries": int(s["nn_queries"]), "n_borders": int(s["n_borders"]),
       "parent_sum": int(n["parent"].astype(np.int64).sum()), "cost_sum": float(n["cost"].sum()).hex(),
       "sampling_trig": "libm" if LIBM else "portable", "oracle_seconds": round(time.time() - t0, 1)}
print(out)
name = ("full_size_run" if WAVE == 8192 else "full_size_run_w%d" % WAVE) + ("_libm" if LIBM else "") + ".json"
json.dump(out, open(os.path.join(HERE, name), "w"), indent=1)
