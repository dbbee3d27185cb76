#!/usr/bin/env python3
"""Regenerate tests/golden/libm_runs.json: CPU-oracle runs in the REFERENCE-PINNED sampling mode (TRIG_LIBM: glibc
cos / sin / acos, the arithmetic tests/golden/ref_primitives.json pins to the reference's own randGen.h), and how
often the portable-trig mode the kernels use gives a different forest.

 * "runs": BASELINE configs[0] (test_2D geometry, 3 XML points, wave 1 = the reference's sequential loop), configs[1]
   (triang.obj, 5 XML points, 100 k-node budget, wave 64) and a 100 k-node slice of configs[2] (dense_3D, 10 seeded
   roots, wave 8192).  tests/test_gpu_parity.py::test_libm_sampling_mode_equals_the_libm_oracle replays them on the
   GPU with sffgpu_forest_cfg::libm_sampling = 1.
 * "divergence": 24 seeds x 2 maps, 20 k iterations at wave 1, PORTABLE vs LIBM sampling (collision arithmetic
   identical): fingerprint equal or not, and the iteration of the first node that differs.
Collision uses the portable rotation in both (the rotation matrix of a pose differs by <= 1 ulp between the two trig
implementations; tests/test_oracle_cpu.py checks that the collision boolean does not move on 60 k near-surface poses)."""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import common  # noqa: E402
import oracle_lib as O  # noqa: E402
from make_config_runs import summary  # noqa: E402

RUNS = {
    # name: (scenario, roots, budget, max_iterations, wave)
    "configs[0] dense2d 3 xml points wave 1": ("dense2d", 3, 10000, 200000, 1),
    "configs[1] triang 5 xml points 100k wave 64": ("triang", 5, 100000, 2**31 - 1, 64),
    "configs[2] dense3d 10 roots 100k slice wave 8192": ("dense3d", 10, 100000, 2**31 - 1, 8192),
}


def roots_of(sc, w, n):
    if sc["xml_points"] is not None:
        return sc["xml_points"][:n]
    return common.free_roots(w.collide, sc["limits"], n, seed=1, dim=sc["dim"])


if __name__ == "__main__":
    out = {"runs": {}, "divergence": []}
    for key, (name, nroots, budget, iters, wave) in RUNS.items():
        sc = common.scenario(name)
        w = O.World(sc["env"], sc["robot"], O.TRIG_PORTABLE)
        f = O.Forest(w, roots_of(sc, w, nroots), sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"],
                     dim=sc["dim"], max_iterations=iters, node_budget=budget, wave=wave, seed=1, trig=O.TRIG_LIBM)
        t0 = time.time()
        f.run()
        out["runs"][key] = summary(f)
        out["runs"][key]["oracle_seconds"] = round(time.time() - t0, 1)
        print(key, out["runs"][key], flush=True)
    for name in ("dense3d", "triang"):
        sc = common.scenario(name)
        w = O.World(sc["env"], sc["robot"], O.TRIG_PORTABLE)
        for seed in range(1, 25):
            res = {}
            for mode, trig in (("portable", O.TRIG_PORTABLE), ("libm", O.TRIG_LIBM)):
                f = O.Forest(w, roots_of(sc, w, 5), sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"],
                             dim=sc["dim"], max_iterations=20000, wave=1, seed=seed, trig=trig)
                f.run()
                res[mode] = (f.fingerprint(), f.nodes())
            same = res["portable"][0] == res["libm"][0]
            a, b = res["portable"][1], res["libm"][1]
            m = min(len(a["parent"]), len(b["parent"]))
            topo = (len(a["parent"]) == len(b["parent"]) and np.array_equal(a["parent"], b["parent"])
                    and np.array_equal(a["iter"], b["iter"]))
            diff = np.where((a["pos"][:m] != b["pos"][:m]).any(axis=1))[0]
            out["divergence"].append({"map": name, "seed": seed, "nodes": int(len(a["parent"])), "bit_identical": bool(same),
                                      "same_topology": bool(topo),
                                      "first_differing_node": int(diff[0]) if len(diff) else None,
                                      "max_abs_position_difference": float(np.abs(a["pos"][:m] - b["pos"][:m]).max()) if topo else None})
            print(out["divergence"][-1], flush=True)
    json.dump(out, open(os.path.join(HERE, "libm_runs.json"), "w"), indent=1)
