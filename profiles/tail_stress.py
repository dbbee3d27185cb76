"""SFF* with the passes after the first as one launch (k_star_tail) against one launch per pass (SFFGPU_STAR_TAIL=0): the same
forest - fingerprint and reference-equivalent counters - over many seeds, maps and wave sizes.  The tail's workgroups exchange
their words through memory inside ONE launch; a word read stale would show here as a different forest or as a round that
does not settle.  usage: python3 profiles/tail_stress.py [seeds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common  # noqa: E402
import space_filling_forest_star_amd as S  # noqa: E402

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
KEYS = ("iterations", "n_nodes", "n_borders", "collide_calls", "path_free_calls", "nn_queries", "frontier_size", "closed_size", "solved",
        "star_rounds", "star_members", "star_rewires")
bad = 0
runs = 0
for name, wave, budget, n_roots in (("building", 8192, 120000, 20), ("building", 1024, 40000, 20), ("dense3d", 16384, 400000, 10),
                                    ("dense3d", 2048, 60000, 10), ("triang", 2048, 30000, 5)):
    sc = common.scenario(name)
    ctx = S.Context(0)
    ctx.upload_env(sc["env"])
    ctx.upload_robot(sc["robot"])
    for seed in range(1, n_seeds + 1):
        roots = sc["xml_points"][:n_roots] if sc["xml_points"] is not None else \
            common.free_roots(lambda p: int(ctx.collide_poses(p[None, :])[0]), sc["limits"], n_roots, seed=seed, dim=sc["dim"])
        out = []
        for tail in ("1", "0"):
            os.environ["SFFGPU_STAR_TAIL"] = tail
            f = S.Forest(ctx, roots, sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=sc["dim"], optimize=True,
                         max_iterations=2**31 - 1, node_budget=budget, wave=wave, seed=seed)
            f.run()
            st = f.stats()
            out.append((f.fingerprint(), {k: int(st[k]) for k in KEYS}, int(st["host_fallback_waves"]), int(st["star_passes"])))
            f.close()
        runs += 1
        same = out[0][0] == out[1][0] and out[0][1] == out[1][1]
        if not same or out[0][2] != out[1][2]:   # (a fallback both ways is a bounded list of the round engine, not the tail)
            bad += 1
        print("%-9s wave %5d seed %2d nodes %7d rewires %6d passes tail %5d chain %5d fallbacks %d/%d  %s" % (
            name, wave, seed, out[0][1]["n_nodes"], out[0][1]["star_rewires"], out[0][3], out[1][3], out[0][2], out[1][2],
            "same forest" if same else "DIFFERENT"), flush=True)
    ctx.close() if hasattr(ctx, "close") else None
print("%d comparisons, %d bad" % (runs, bad))
sys.exit(1 if bad else 0)
