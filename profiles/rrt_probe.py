"""The RRT legs of bench.py on their own (for rocprofv3): dense_3D.obj, 6-DoF, adaptive speculative waves.
argv: iterations [star | rrt | multi]  (RRT* one root / RRT one root / Multi-T-RRT ten roots).  Prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common  # noqa: E402
import space_filling_forest_star_amd as S  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 150000
mode = sys.argv[2] if len(sys.argv) > 2 else "star"
nroot = 10 if mode == "multi" else 1
sc = common.scenario("dense3d")
ctx = S.Context(0)
ctx.upload_env(sc["env"])
ctx.upload_robot(sc["robot"])
roots = common.free_roots(lambda p: int(ctx.collide_poses(p[None, :])[0]), sc["limits"], 10, seed=1)
for rep in range(2):
    r = S.Rrt(ctx, roots[:nroot], sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=6, optimize=(mode == "star"),
              max_iterations=iters, wave=0, seed=1)
    t0 = time.perf_counter()
    r.run()
    dt = time.perf_counter() - t0
    st = r.stats()
    r.close()
print(json.dumps({"config": "dense_3D %s %d root(s) %d iterations" % ({"star": "RRT*", "rrt": "RRT", "multi": "Multi-T-RRT"}[mode], nroot, iters), "iterations_per_s": st["iterations"] / dt,
                  "accepted_nodes_per_s": (st["n_nodes"] - 1) / dt, "nodes": st["n_nodes"], "waves": st["waves"],
                  "speculated": st["speculated"], "committed": st["committed"], "seconds": dt,
                  "collision_checks_per_s": st["collide_calls"] / dt}))
