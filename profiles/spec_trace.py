import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, common
import space_filling_forest_star_amd as S
sc = common.scenario("dense3d")
ctx = S.Context(0); ctx.upload_env(sc["env"]); ctx.upload_robot(sc["robot"])
roots = common.free_roots(lambda p: int(ctx.collide_poses(p[None, :])[0]), sc["limits"], 10, seed=1)
iters = int(os.environ.get("DBG_ITERS", "50000"))
tr = {}
for spec in ("0", "1"):
    os.environ["SFFGPU_SPEC"] = spec
    path = "/tmp/trace_%s.bin" % spec
    if os.path.exists(path): os.remove(path)
    os.environ["SFFGPU_SEQ_TRACE"] = path
    f = S.Forest(ctx, roots, sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=6, max_iterations=iters, wave=1, seed=1, optimize=True)
    f.run(); st = f.stats(); f.close()
    tr[spec] = np.fromfile(path, dtype=np.int32).reshape(-1, 8)
    print(spec, "waves", st["waves"], "trace", tr[spec].shape, "steps", st["spec_steps"], "fallback", st["host_fallback_waves"])
a, b = tr["0"], tr["1"]
m = min(len(a), len(b))
d = np.nonzero(np.any(a[:m] != b[:m], axis=1))[0]
print("differing waves", len(d), d[:5])
if len(d):
    i = d[0]
    for j in range(max(0, i - 4), min(m, i + 3)):
        print(j, "seq", a[j], "spec", b[j])
