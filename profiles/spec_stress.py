"""k_spec_waves against k_seq_waves (fingerprints + every counter) over maps, seeds, root counts, ThresholdMisses and both
solvers - no oracle in the loop, so many cases per second.  Prints one line per mismatch and a summary."""
import sys, os, time, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, common
import space_filling_forest_star_amd as S
KEYS = ("iterations", "solved", "n_nodes", "frontier_size", "closed_size", "n_borders", "collide_calls", "path_free_calls", "nn_queries", "waves")
bad = n = 0
t0 = time.time()
for name in ("dense3d", "triang", "building", "dense3d_coarse", "dense2d"):
    sc = common.scenario(name)
    ctx = S.Context(0); ctx.upload_env(sc["env"]); ctx.upload_robot(sc["robot"])
    for seed in range(1, int(os.environ.get("STRESS_SEEDS", "7"))):
        rs = np.random.RandomState(seed)
        nroots = int(rs.randint(2, 12)); tm = int(rs.choice([3, 5, 5, 5, 8])); opt = bool(rs.randint(0, 2)); iters = int(rs.choice([3000, 9000, 25000]))
        if sc["xml_points"] is not None and rs.randint(0, 2):
            roots = sc["xml_points"][:min(nroots, len(sc["xml_points"]))]
        else:
            roots = common.free_roots(lambda p: int(ctx.collide_poses(p[None, :])[0]), sc["limits"], nroots, seed=seed, dim=sc["dim"])
        res = {}
        for spec in ("1", "0"):
            os.environ["SFFGPU_SPEC"] = spec
            f = S.Forest(ctx, roots, sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=sc["dim"], max_iterations=iters,
                         wave=1, seed=seed, optimize=opt, threshold_misses=tm)
            f.run(); st = f.stats(); res[spec] = (f.fingerprint(), tuple(st[k] for k in KEYS), st["spec_committed"], st["host_fallback_waves"]); f.close()
        n += 1
        if res["1"][:2] != res["0"][:2]:
            bad += 1
            print("MISMATCH", name, "seed", seed, "roots", len(roots), "tm", tm, "opt", opt, "iters", iters, res["1"][1], res["0"][1], flush=True)
        elif res["1"][2] != res["1"][1][0]:
            print("note: fallback", name, seed, res["1"][2], res["1"][1][0], res["1"][3], flush=True)
    ctx.close()
print("cases %d, mismatches %d, %.1f s" % (n, bad, time.time() - t0))
