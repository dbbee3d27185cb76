"""wave = 1 (the reference's loop order) on the device: k_spec_waves (speculated over many wavefronts) against k_seq_waves
(one wavefront), plain SFF and SFF*, dense_3D, 10 roots; fingerprints must agree.  Environment: SFFGPU_SPEC_DEPTH / _SETS."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, common
import space_filling_forest_star_amd as S
sc = common.scenario(os.environ.get("PROBE_MAP", "dense3d"))
ctx = S.Context(0); ctx.upload_env(sc["env"]); ctx.upload_robot(sc["robot"])
roots = common.free_roots(lambda p: int(ctx.collide_poses(p[None, :])[0]), sc["limits"], 10, seed=1)
iters_list = [int(x) for x in os.environ.get("PROBE_ITERS", "8000,100000").split(",")]
for opt in (False, True):
    for iters in iters_list:
        fps = {}
        for spec in ("1", "0"):
            os.environ["SFFGPU_SPEC"] = spec
            f = S.Forest(ctx, roots, sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=6, max_iterations=iters, wave=1, seed=1, optimize=opt)
            t = time.perf_counter(); f.run(); dt = time.perf_counter() - t
            st = f.stats(); fps[spec] = f.fingerprint(); f.close()
            print("SFF*" if opt else "SFF ", "spec" if spec == "1" else "seq ", iters, "nodes/s %.0f it/s %.0f us/it %.2f" % ((st["n_nodes"] - 10) / dt, st["iterations"] / dt, 1e6 * dt / st["iterations"]),
                  "| steps %d evaluated %d committed %d ratio %.2f it/step %.2f fallback %d" % (st["spec_steps"], st["spec_evaluated"], st["spec_committed"],
                  st["spec_evaluated"] / max(1, st["spec_committed"]), st["spec_committed"] / max(1, st["spec_steps"]), st["host_fallback_waves"]), flush=True)
        print("   fingerprints equal:", fps["1"] == fps["0"], flush=True)
