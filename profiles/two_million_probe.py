import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import common, space_filling_forest_star_amd as S
sc = common.scenario("dense3d")
ctx = S.Context(0); ctx.upload_env(sc["env"]); ctx.upload_robot(sc["robot"])
roots = common.free_roots(lambda p: int(ctx.collide_poses(p[None, :])[0]), sc["limits"], 10, seed=1)
kw = dict(dist_tree=14.0, sampling_dist=11.0, dim=6, optimize=True, max_iterations=2**31 - 1, wave=16384, seed=1)
f = S.Forest(ctx, roots, sc["limits"], node_budget=2000000, **kw)
f.run()
s = f.stats()
print({k: s[k] for k in ("n_nodes", "host_fallback_waves", "star_rounds", "waves", "total_ms")})
