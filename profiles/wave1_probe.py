"""wave = 1 (the reference's loop order) on the device: k_seq_waves, plain SFF and SFF*, dense_3D, 10 roots."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, common
import space_filling_forest_star_amd as S
sc = common.scenario("dense3d")
ctx = S.Context(0); ctx.upload_env(sc["env"]); ctx.upload_robot(sc["robot"])
roots = common.free_roots(lambda p: int(ctx.collide_poses(p[None, :])[0]), sc["limits"], 10, seed=1)
for opt in (False, True):
    for iters in (8000, 8000, 100000):
        f = S.Forest(ctx, roots, sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=6, max_iterations=iters, wave=1, seed=1, optimize=opt)
        t = time.perf_counter(); f.run(); dt = time.perf_counter() - t
        st = f.stats(); f.close()
        print("SFF*" if opt else "SFF ", iters, "nodes/s %.0f it/s %.0f us/it %.2f" % ((st["n_nodes"] - 10) / dt, st["iterations"] / dt, 1e6 * dt / st["iterations"]), flush=True)
