"""The RRT session's wave engine against itself: repaired slots + the dry walk (default) vs no repair vs the separate batch calls
(no chain) vs one iteration per round trip (wave = 1, small cases only) - nodes, parents, costs, links, counters and the stream
position over maps, seeds, root counts, goal bias, RRT / RRT*, wave sizes.  No oracle in the loop: many cases per second.
Prints one line per mismatch and a summary."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import common  # noqa: E402
import space_filling_forest_star_amd as S  # noqa: E402

SKIP = ("waves", "speculated", "committed", "total_ms", "host_ms", "sweep_ms", "collide_ms")
bad = n = 0
t0 = time.time()
for name in ("dense3d", "triang", "building", "dense3d_coarse", "dense2d"):
    sc = common.scenario(name)
    ctx = S.Context(0)
    ctx.upload_env(sc["env"])
    ctx.upload_robot(sc["robot"])
    for seed in range(1, int(os.environ.get("STRESS_SEEDS", "9"))):
        rs = np.random.RandomState(1000 + seed)
        nroots = int(rs.choice([1, 1, 2, 4, 9]))
        opt = bool(rs.randint(0, 2))
        iters = int(rs.choice([800, 2500, 7000]))
        wave = int(rs.choice([0, 0, 64, 700, 4096]))
        use_goal = nroots == 1 and bool(rs.randint(0, 2))
        bias = float(rs.choice([0.0, 0.1])) if use_goal else 0.0
        if sc["xml_points"] is not None and rs.randint(0, 2) and len(sc["xml_points"]) > nroots:
            pts = sc["xml_points"]
        else:
            pts = common.free_roots(lambda p: int(ctx.collide_poses(p[None, :])[0]), sc["limits"], nroots + 1, seed=seed, dim=sc["dim"])
        roots, goal = pts[:nroots], (pts[nroots] if use_goal else None)
        kw = dict(dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=sc["dim"], optimize=opt, goal=goal, priority_bias=bias,
                  max_iterations=iters, seed=seed)
        res = {}
        engines = [("repair", {}, wave), ("no_repair", {"SFFGPU_RRT_REPAIR": "0"}, wave), ("no_dry", {"SFFGPU_RRT_DRY": "0"}, wave),
                   ("no_chain", {"SFFGPU_RRT_CHAIN": "0"}, wave)]
        if iters <= 800:
            engines.append(("one_by_one", {}, 1))
        for tag, env, wv in engines:
            for k, v in env.items():
                os.environ[k] = v
            r = S.Rrt(ctx, roots, sc["limits"], wave=wv, **kw)
            for k in env:
                os.environ.pop(k)
            r.run()
            st, nd, lk = r.stats(), r.nodes(), r.links()
            key = tuple(sorted((k, v) for k, v in st.items() if k not in SKIP))
            res[tag] = (key, tuple(v.tobytes() for _, v in sorted(nd.items())), tuple(v.tobytes() for _, v in sorted(lk.items())), st["waves"])
            r.close()
        n += 1
        ref = res["no_chain"][:3]
        for tag in res:
            if res[tag][:3] != ref:
                bad += 1
                print("MISMATCH", name, "seed", seed, "roots", nroots, "opt", opt, "iters", iters, "wave", wave, "goal", use_goal, bias, tag, flush=True)
        if os.environ.get("STRESS_VERBOSE"):
            print(name, seed, nroots, opt, iters, wave, {t: res[t][3] for t in res}, flush=True)
    ctx.close()
print("cases %d, mismatches %d, %.1f s" % (n, bad, time.time() - t0))
