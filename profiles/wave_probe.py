"""Plain SFF on dense_3D (10 roots) at one wave size, for rocprofv3: argv = wave, iterations.  Prints nodes/s and us per wave."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, common
import space_filling_forest_star_amd as S
wv = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 150000
sc = common.scenario("dense3d")
ctx = S.Context(0); ctx.upload_env(sc["env"]); ctx.upload_robot(sc["robot"])
roots = common.free_roots(lambda p: int(ctx.collide_poses(p[None, :])[0]), sc["limits"], 10, seed=1)
for rep in range(2):
    f = S.Forest(ctx, roots, sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=6, max_iterations=iters, wave=wv, seed=1)
    t = time.perf_counter(); f.run(); dt = time.perf_counter() - t
    st = f.stats(); f.close()
print("wave", wv, "nodes/s %.0f it/s %.0f us/wave %.1f rounds/wave %.2f host_ms %.1f of %.1f" % ((st["n_nodes"] - 10) / dt, st["iterations"] / dt, 1e6 * dt / st["waves"], st["rounds"] / max(1, st["waves"]), st["host_ms"], st["total_ms"]), flush=True)
