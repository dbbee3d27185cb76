"""SURVEY.md 8(d) k-NN micro-benchmark: N nodes uniform in the dense_3D limits (angles uniform in [-pi, pi)), Q = 4096
queries per launch, k in {1, 32}, for the two exact k-nearest kernels behind sffgpu_knn:
  k_knn_linear  one wavefront per query sweeps the whole store (coalesced fp32 columns, fp64 re-test of what beats the
                current k-th distance): algorithmic bytes = 24 B x N PER QUERY - the formula of 8(d) prices the store
                once per launch, so both figures are printed
  k_knn_grid    with the index of sffgpu_nodes_index: shells of cells around the query until the k-th distance is covered
and the linear radius sweep k_sweep at N = 16 M / 32 M nodes (384 / 768 MB of fp32 columns: beyond the 256 MB Infinity
Cache, so its GB/s is an HBM figure), one query per pass.
Kernel time from the library's HIP events on its launch stream.  Usage: python profiles/knn_microbench.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import space_filling_forest_star_amd as S  # noqa: E402

lim = np.array([-60.0, 2060.0, -60.0, 2110.0, 0.0, 1000.0])
vol = (lim[1] - lim[0]) * (lim[3] - lim[2]) * (lim[5] - lim[4])
rs = np.random.RandomState(1)
ctx = S.Context(0)


def cloud(N):
    pos = np.empty((N, 6))
    for a in range(3):
        pos[:, a] = rs.uniform(lim[2 * a], lim[2 * a + 1], N)
    pos[:, 3:] = rs.uniform(-np.pi, np.pi, (N, 3))
    return pos


def timed(fn, reps):
    fn()
    ms0, _ = ctx.kernel_times()
    for _ in range(reps):
        fn()
    ms1, _ = ctx.kernel_times()
    return (ms1[0] - ms0[0]) / reps * 1e-3


Q = 4096
for N in (10000, 100000, 1000000, 2000000):
    pos = cloud(N)
    q = pos[rs.randint(0, N, Q)] + rs.normal(0, 5.0, (Q, 6))
    for kernel in ("k_knn_linear", "k_knn_grid"):
        ctx.nodes_reset(N + 64)
        ctx.nodes_append(pos, np.zeros(N, np.int32))
        if kernel == "k_knn_grid":
            # cell edge ~ the radius of a ball that holds 32 nodes (what the planner's grid has at the same density)
            ctx.nodes_index(lim, max(18.0, (32.0 * vol / N / 4.19) ** (1.0 / 3.0)))
        for k in (1, 32):
            reps = 3 if (kernel == "k_knn_linear" and N >= 1000000) else 10
            t = timed(lambda: ctx.knn(q, k), reps)
            print(json.dumps({"kernel": "sffk::" + kernel, "N": N, "Q": Q, "k": k, "us_per_launch": round(t * 1e6, 1),
                              "queries_per_s": round(Q / t), "GBps_store_once_per_launch": round(24.0 * N / t / 1e9, 2),
                              "frac_of_8TBps_store_once": round(24.0 * N / t / 8e12, 5),
                              "GBps_store_per_query": round(24.0 * N * Q / t / 1e9, 1) if kernel == "k_knn_linear" else None}),
                  flush=True)
for N in (16000000, 32000000):
    pos = cloud(N)
    ctx.nodes_reset(N + 64)
    for a in range(0, N, 4000000):
        ctx.nodes_append(pos[a:a + 4000000], np.zeros(len(pos[a:a + 4000000]), np.int32))
    r = (32.0 * vol / N / 4.19) ** (1.0 / 3.0)
    for Qs in (1, 4):
        q = pos[rs.randint(0, N, Qs)] + rs.normal(0, 2.0, (Qs, 6))
        t = timed(lambda: ctx.radius(q, r, cap=64), 20)
        print(json.dumps({"kernel": "sffk::k_sweep", "N": N, "Q": Qs, "store_MB": round(24.0 * N / 1e6), "radius": round(r, 2),
                          "us_per_pass": round(t * 1e6, 1), "achieved_GBps": round(24.0 * N / t / 1e9, 1),
                          "frac_of_8TBps": round(24.0 * N / t / 8e12, 4)}), flush=True)
ctx.close()
