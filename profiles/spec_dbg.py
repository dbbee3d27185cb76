import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, common
import space_filling_forest_star_amd as S
sc = common.scenario("dense3d")
ctx = S.Context(0); ctx.upload_env(sc["env"]); ctx.upload_robot(sc["robot"])
roots = common.free_roots(lambda p: int(ctx.collide_poses(p[None, :])[0]), sc["limits"], 10, seed=1)
OPT = os.environ.get("DBG_OPT", "1") == "1"
def run(spec, iters, opt=OPT):
    os.environ["SFFGPU_SPEC"] = spec
    f = S.Forest(ctx, roots, sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=6, max_iterations=iters, wave=1, seed=1, optimize=opt)
    f.run()
    st = f.stats(); n = f.nodes(); b = f.borders(); fp = f.fingerprint(); f.close()
    return st, n, b, fp
def wrap(d):
    d = np.where(d > np.pi, d - 2 * np.pi, d); d = np.where(d < -np.pi, d + 2 * np.pi, d); return d
def dist6(a, b):
    d = a - b; d[..., 3:] = wrap(d[..., 3:]); return np.sqrt((d * d).sum(-1))
iters = int(os.environ.get("DBG_ITERS", "50000"))
s0, n0, b0, f0 = run("0", iters)
for rep in range(int(os.environ.get("DBG_REPS", "3"))):
    s1, n1, b1, f1 = run("1", iters)
    print(iters, "rep", rep, "equal", f1 == f0, "nodes", s1["n_nodes"], s0["n_nodes"], "steps", s1["spec_steps"])
    if f1 != f0:
        m = min(len(n1["parent"]), len(n0["parent"]))
        d = np.nonzero(np.any(n1["pos"][:m] != n0["pos"][:m], axis=1))[0]
        i = d[0]
        print("  first node with another position", i, "iter spec", n1["iter"][i], "seq", n0["iter"][i], "parent spec", n1["parent"][i], "seq", n0["parent"][i], "tree", n1["tree"][i], n0["tree"][i])
        p = n1["pos"][i]; par = n1["parent"][i]
        pd = dist6(n0["pos"][par], p)
        dd = dist6(n0["pos"][:i].copy(), p)
        near = np.nonzero(dd < max(pd, sc["dist_tree"]))[0]
        print("  pdist", pd, "nodes (seq run, < i) inside the query ball:")
        for j in near:
            print("     id", j, "tree", n0["tree"][j], "d", dd[j], "iter", n0["iter"][j], "qualifies", (n0["tree"][j] == n1["tree"][i] and dd[j] < pd - 1e-9) or (n0["tree"][j] != n1["tree"][i] and dd[j] < sc["dist_tree"] - 1e-9))
