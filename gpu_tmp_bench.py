import sys, time, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, common
import space_filling_forest_star_amd as S
name = sys.argv[1]; optimize = int(sys.argv[2]); wave = int(sys.argv[3]); budget = int(sys.argv[4]); nroots = int(sys.argv[5])
sc = common.scenario(name)
ctx = S.Context(0); ctx.upload_env(sc["env"]); ctx.upload_robot(sc["robot"])
roots = common.free_roots(lambda p: int(ctx.collide_poses(p[None,:])[0]), sc["limits"], nroots, seed=1)
f = S.Forest(ctx, roots, sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=6, optimize=bool(optimize), max_iterations=2**31-1, node_budget=budget, wave=wave, seed=1)
f.run(3); s0 = f.stats(); t0 = time.perf_counter()
f.run(int(sys.argv[6])); t1 = time.perf_counter(); s1 = f.stats()
dt = t1 - t0
print(name, "optimize", optimize, "wave", wave, "nodes", s1["n_nodes"], "nodes/s %.0f" % ((s1["n_nodes"]-s0["n_nodes"])/dt), "iters", s1["iterations"], "checks/s %.3g" % ((s1["collide_calls"]-s0["collide_calls"])/dt), "time %.2f" % dt, "sweep_ms %.0f collide_ms %.0f host_ms %.0f" % (s1["sweep_ms"]-s0["sweep_ms"], s1["collide_ms"]-s0["collide_ms"], s1["host_ms"]-s0["host_ms"]))
