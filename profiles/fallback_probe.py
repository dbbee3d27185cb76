"""why a wave went to the host path: SFFGPU_PROFILE=1 python3 profiles/fallback_probe.py <seed> (building, SFF*, waves of 8192 slots, the XML roots) prints the\nfault reasons; seeds 4 / 7 / 8: one sample overflows its hit / neighbour list in one round - a bounded list of the round engine."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import common, space_filling_forest_star_amd as S
sc = common.scenario("building")
ctx = S.Context(0); ctx.upload_env(sc["env"]); ctx.upload_robot(sc["robot"])
seed = int(sys.argv[1])
roots = sc["xml_points"][:20] if sc["xml_points"] is not None else common.free_roots(lambda p: int(ctx.collide_poses(p[None, :])[0]), sc["limits"], 20, seed=seed, dim=sc["dim"])
f = S.Forest(ctx, roots, sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=sc["dim"], optimize=True,
             max_iterations=2**31 - 1, node_budget=120000, wave=8192, seed=seed)
f.run(); st = f.stats(); print(seed, st["n_nodes"], st["host_fallback_waves"])
