import sys, time, os
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, common
import space_filling_forest_star_amd as S
sc = common.scenario("dense3d")
ctx = S.Context(0); ctx.upload_env(sc["env"]); ctx.upload_robot(sc["robot"])
roots = common.free_roots(lambda p: int(ctx.collide_poses(p[None, :])[0]), sc["limits"], 10, seed=1)
for wave in (1, 8, 64):
    for rep in range(2):
        f = S.Forest(ctx, roots, sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=6, max_iterations=8000 if wave == 1 else 40000, wave=wave, seed=1, priority_bias=0.95)
        t = time.perf_counter(); f.run(); dt = time.perf_counter() - t
        st = f.stats(); dev = f.device_engine(); f.close()
    print("priority wave", wave, "device engine", dev, "nodes/s %.0f it/s %.0f" % ((st["n_nodes"] - 10) / dt, st["iterations"] / dt), flush=True)
