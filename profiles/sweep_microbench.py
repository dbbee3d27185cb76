"""Roofline of the linear neighbour sweep (k_sweep) on its own - SURVEY.md 8(d) micro-benchmark: N nodes uniform
in the dense_3D limits (angles uniform in [-pi, pi)), Q queries per pass, radius for ~32 neighbours in xyz; algorithmic bytes = 24 B x N
per pass (six fp32 columns), kernel time from the library's HIP events.  The planner's own fixed-radius query
goes through the grid instead; this is the engine behind sffgpu_radius / sffgpu_knn.
Usage: python profiles/sweep_microbench.py   (prints one JSON line per (N, Q))"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import space_filling_forest_star_amd as S  # noqa: E402

lim = np.array([-60.0, 2060.0, -60.0, 2110.0, 0.0, 1000.0])
rs = np.random.RandomState(1)
ctx = S.Context(0)
for N in (100000, 1000000, 2000000, 8000000):
    pos = np.empty((N, 6))
    for a in range(3):
        pos[:, a] = rs.uniform(lim[2 * a], lim[2 * a + 1], N)
    pos[:, 3:] = rs.uniform(-np.pi, np.pi, (N, 3))
    ctx.nodes_reset(N + 64)
    ctx.nodes_append(pos, np.zeros(N, np.int32))
    vol = (lim[1] - lim[0]) * (lim[3] - lim[2]) * (lim[5] - lim[4])
    r = (32.0 * vol / N / 4.19) ** (1.0 / 3.0)           # ~32 nodes of the xyz ball (k of SFF* / RRT* at 1e6 nodes)
    for Q in (1, 4, 16, 64, 4096):
        q = pos[rs.randint(0, N, Q)] + rs.normal(0, 5.0, (Q, 6))
        ctx.radius(q, r, cap=64)                          # warm-up
        ms0, n0 = ctx.kernel_times()
        reps = 20 if Q <= 64 else 5
        for _ in range(reps):
            ctx.radius(q, r, cap=64)
        ms1, n1 = ctx.kernel_times()
        t = (ms1[0] - ms0[0]) / reps * 1e-3
        print(json.dumps({"kernel": "sffk::k_sweep", "N": N, "Q": Q, "radius": round(r, 1), "us_per_pass": round(t * 1e6, 2),
                          "achieved_GBps": round(24.0 * N / t / 1e9, 1), "frac_of_8TBps": round(24.0 * N / t / 8e12, 4),
                          "pairs_per_s": round(N * Q / t / 1e12, 3)}), flush=True)
ctx.close()
