import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, common
import space_filling_forest_star_amd as S
sc = common.scenario("dense3d")
ctx = S.Context(0); ctx.upload_env(sc["env"]); ctx.upload_robot(sc["robot"])
roots = common.free_roots(lambda p: int(ctx.collide_poses(p[None, :])[0]), sc["limits"], 10, seed=1)
for eng in ("host", "device"):
    os.environ["SFFGPU_ENGINE"] = eng
    for wv, iters in ((1, 8000), (8, 40000), (64, 150000), (256, 400000)):
        for rep in range(2):
            f = S.Forest(ctx, roots, sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=6, max_iterations=iters, wave=wv, seed=1)
            t = time.perf_counter(); f.run(); dt = time.perf_counter() - t
            st = f.stats(); f.close()
        print(eng, wv, "nodes/s %.0f it/s %.0f us/wave %.1f host_ms %.1f of %.1f" % ((st["n_nodes"] - 10) / dt, st["iterations"] / dt, 1e6 * dt / st["waves"], st["host_ms"], st["total_ms"]), flush=True)
