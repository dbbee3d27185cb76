"""The stand-alone linear sweep of bench.py's `sweep_kernel_roofline` leg on its own (16 M uniform nodes in the dense_3D
limits = 384 MB of fp32 columns, ONE query per pass, radius for ~32 neighbours, 30 passes), for rocprofv3:
profiles/collect_sweep.sh traces it and collects FETCH_SIZE / WRITE_SIZE of sffk::k_sweep."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import space_filling_forest_star_amd as S  # noqa: E402

Nn = int(sys.argv[1]) if len(sys.argv) > 1 else 16000000
lim = np.array([-60.0, 2060.0, -60.0, 2110.0, 0.0, 1000.0])
rs = np.random.RandomState(1)
pts = np.empty((Nn, 6))
for a in range(3):
    pts[:, a] = rs.uniform(lim[2 * a], lim[2 * a + 1], Nn)
pts[:, 3:] = rs.uniform(-np.pi, np.pi, (Nn, 3))
cs = S.Context(0)
cs.nodes_reset(Nn + 64)
for a0 in range(0, Nn, 4000000):
    cs.nodes_append(pts[a0:a0 + 4000000], np.zeros(len(pts[a0:a0 + 4000000]), np.int32))
vol = (lim[1] - lim[0]) * (lim[3] - lim[2]) * (lim[5] - lim[4])
rad = (32.0 * vol / Nn / 4.19) ** (1.0 / 3.0)
qq = pts[rs.randint(0, Nn, 1)] + rs.normal(0, 5.0, (1, 6))
cs.radius(qq, rad, cap=64)
ms0, _ = cs.kernel_times()
reps = 30
for _ in range(reps):
    cs.radius(qq, rad, cap=64)
ms1, _ = cs.kernel_times()
tt = (ms1[0] - ms0[0]) / reps * 1e-3
print(json.dumps({"kernel": "sffk::k_sweep", "nodes": Nn, "queries_per_pass": 1, "us_per_pass_by_hip_events": tt * 1e6,
                  "achieved_GBps": 24.0 * Nn / tt / 1e9, "frac_of_8TBps": 24.0 * Nn / tt / 8e12}))
cs.close()
