"""Scaling estimate on ONE GPU: world ranks of one sharded forest are run one after the other (one Context
each, records handed over in-process instead of the RCCL all-gather), every rank's round_begin / round_commit
is timed, and a round is charged max-over-ranks(begin) + max-over-ranks(commit), which is what a real
N-GPU run would wait for (minus the all-gather latency, tens of microseconds).  Same weak-scaling workload
definition as bench.py.  Usage: python profiles/emulate_ranks.py [world ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common  # noqa: E402
import space_filling_forest_star_amd as S  # noqa: E402


def run(world, wave=8192, budget=1000000, waves=120, warm=3, strong=False):
    sc = common.scenario("dense3d")
    dens = 1.0 if strong else world ** (-1.0 / 3.0)
    if strong:   # total work fixed: the same forest, the wave split over the ranks
        wave = max(1, wave // world)
        budget = max(1, budget // world)
    ctxs, fs = [], []
    roots = None
    for r in range(world):
        c = S.Context(0)
        c.upload_env(sc["env"])
        c.upload_robot(sc["robot"])
        if roots is None:
            roots = common.free_roots(lambda p: int(c.collide_poses(p[None, :])[0]), sc["limits"], 10, seed=1)
        fs.append(S.Forest(c, roots, sc["limits"], dist_tree=sc["dist_tree"] * dens,
                           sampling_dist=sc["sampling_dist"] * dens, dim=6, max_iterations=2**31 - 1,
                           node_budget=budget * world, wave=wave * world, seed=1, rank=r, world=world))
        ctxs.append(c)
    est = 0.0
    n0 = None
    w_done = 0
    while True:
        s = fs[0].stats()
        if s["waves"] >= warm and n0 is None and not fs[0].in_wave():
            n0, est = s["n_nodes"], 0.0
        if s["waves"] >= warm + waves and not fs[0].in_wave():
            break
        recs, tb = [], []
        done = False
        for f in fs:
            t = time.perf_counter()
            rec, d = f.round_begin()
            tb.append(time.perf_counter() - t)
            recs.append(rec)
            done |= d
        if done:
            break
        allw = np.concatenate(recs)
        counts = np.array([len(r) for r in recs], np.int32)
        tc = []
        for f in fs:
            t = time.perf_counter()
            f.round_commit(allw, counts)
            tc.append(time.perf_counter() - t)
        est += max(tb) + max(tc)
    s = fs[0].stats()
    val = (s["n_nodes"] - n0) / est
    for f in fs:
        f.close()
    for c in ctxs:
        c.close()
    return val, s["n_nodes"], est


if __name__ == "__main__":
    strong = "--strong" in sys.argv
    worlds = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [1, 2, 4, 8]
    base = None
    for w in worlds:
        v, n, e = run(w, strong=strong)
        base = base or v
        print("world %d: est. %.0f nodes/s (x%.2f), %d nodes, %.3f s charged" % (w, v, v / base, n, e), flush=True)
