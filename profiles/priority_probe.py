"""The priority-frontier mode (priorityBias, src/forest.h:126-147,160-181,360-363; src/heap.h) on the headline map:
dense_3D.obj, 10 seeded roots, bias 0.95.  argv: node budget, wave sizes..."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, common
import space_filling_forest_star_amd as S

budget = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
waves = [int(x) for x in sys.argv[2:]] or [1024, 16384]
sc = common.scenario("dense3d")
ctx = S.Context(0)
ctx.upload_env(sc["env"]); ctx.upload_robot(sc["robot"])
roots = common.free_roots(lambda p: int(ctx.collide_poses(p[None, :])[0]), sc["limits"], 10, seed=1)
for wv in waves:
    for rep in range(2):
        f = S.Forest(ctx, roots, sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=6,
                     max_iterations=2**31 - 1, node_budget=budget, wave=wv, seed=1, priority_bias=0.95)
        t0 = time.perf_counter(); f.run(); dt = time.perf_counter() - t0
        st = f.stats(); dev = bool(f.device_engine()); f.close()
    print(json.dumps({"wave": wv, "budget": budget, "nodes": st["n_nodes"], "iterations": st["iterations"], "seconds": dt,
                      "accepted_nodes_per_s": (st["n_nodes"] - 10) / dt, "device_engine": dev, "host_ms": st["host_ms"],
                      "waves": st["waves"], "host_fallback_waves": st["host_fallback_waves"], "total_ms": st["total_ms"], "sweep_ms": st["sweep_ms"], "collide_ms": st["collide_ms"], "commit_ms": st.get("commit_ms", 0)}))
