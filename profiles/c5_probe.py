"""BASELINE configs[4] on one GPU: building.obj, 20 seeded roots, SFF* (optimize=true, choose-parent + rewire), node
budget from argv (default 2 M).  Prints throughput and the time split; `--check` adds the size-independent property
checks of tests/test_gpu_parity.py::test_c5_full_size_properties."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import common  # noqa: E402
import space_filling_forest_star_amd as S  # noqa: E402

budget = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
wave = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
sc = common.scenario("building")
ctx = S.Context(0)
ctx.upload_env(sc["env"])
ctx.upload_robot(sc["robot"])
roots = common.free_roots(lambda p: int(ctx.collide_poses(p[None, :])[0]), sc["limits"], 20, seed=1)
f = S.Forest(ctx, roots, sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=6, optimize=True,
             max_iterations=2**31 - 1, node_budget=budget, wave=wave, seed=1)
t0 = time.perf_counter()
f.run()
dt = time.perf_counter() - t0
st = f.stats()
print(json.dumps({"config": "building.obj 20 roots SFF* budget %d wave %d" % (budget, wave), "nodes": st["n_nodes"],
                  "iterations": st["iterations"], "solved": st["solved"], "seconds": dt,
                  "accepted_nodes_per_s": (st["n_nodes"] - 20) / dt, "collision_checks_per_s": st["collide_calls"] / dt,
                  "host_ms": st["host_ms"], "total_ms": st["total_ms"], "sweep_ms": st["sweep_ms"], "collide_ms": st["collide_ms"],
                  "device_engine": bool(f.device_engine()), "waves": st["waves"], "rounds": st["sweeps"],
                  "star_rounds": st["star_rounds"], "star_passes": st["star_passes"], "star_members": st["star_members"],
                  "star_rewires": st["star_rewires"], "host_fallback_waves": st["host_fallback_waves"]}))
