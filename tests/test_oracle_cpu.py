"""CPU-only checks of the oracle itself: its accelerated collision against its own brute force,
the reference's parser quirks, local-planner semantics, wave semantics, trig modes, and a
committed regression fixture of small solver runs (tests/golden/oracle_runs.json)."""
import json
import os

import numpy as np
import pytest

import common
import oracle_lib as O


@pytest.mark.parametrize("name,n", [("dense3d", 1500), ("triang", 1500), ("dense2d", 1500), ("building", 400)])
def test_collide_hierarchy_equals_brute_force(name, n):
    sc = common.scenario(name)
    w = O.World(sc["env"], sc["robot"], O.TRIG_PORTABLE)
    poses = np.vstack([common.random_poses(sc["limits"], n // 3, 1, sc["dim"]),
                       common.poses_near_surface(sc["env"], n - n // 3, 2, 0.3 * sc["scale"], sc["dim"])])
    fast = w.collide_many(poses)
    brute = np.array([w.collide_brute(p) for p in poses], np.uint8)
    assert np.array_equal(fast, brute)
    assert 0.02 < brute.mean() < 0.98


def test_tri_contact_cases():
    L = O.lib()
    P = np.array([0, 0, 0, 1, 0, 0, 0, 1, 0], float)

    def c(Q):
        return L.sffo_tri_contact(O.dp(P), O.dp(np.array(Q, float)))

    assert c([0.2, 0.2, -1, 0.2, 0.2, 1, 0.8, 0.8, 1]) == 1          # pierces
    assert c([0.2, 0.2, 0.5, 0.2, 0.8, 0.5, 0.8, 0.2, 0.5]) == 0      # parallel above
    assert c([1, 0, 0, 2, 0, 0, 1, 1, 0]) == 1                        # coplanar, shares a vertex: touching counts
    assert c([1.001, 0, 0, 2, 0, 0, 1.001, 1, 0]) == 0                # coplanar, separated (needs the in-plane axes)
    assert c([0.25, 0.25, 0, 0.5, 0.25, 0, 0.25, 0.5, 0]) == 1        # coplanar, contained
    assert c([5, 5, 5, 5, 5, 5, 5, 5, 5]) == 0                        # degenerate far away: rejected by the box test


def test_parser_quirks(tmp_path):
    # 'vn' lines become vertices, 'f a//b' keeps a, the position is added BEFORE scaling, o-groups
    # do not reset indices (reference src/environment.h:125-223)
    p = tmp_path / "m.obj"
    p.write_text("# c\no A_x\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3//1\no B_y\nv 0 0 1\nf 1 2 5\nf 2 3 4\n")
    t = O.parse_obj(str(p), pos=(1, 2, 3), scale=10.0)
    assert t.shape == (3, 9)
    assert np.array_equal(t[0], [10, 20, 30, 20, 20, 30, 10, 30, 30])
    assert np.array_equal(t[1][6:], [10, 20, 40])          # vertex 5 is the one after the vn "vertex"
    assert np.array_equal(t[2][6:], [10, 20, 40 - 0])      # vertex 4 = the vn line (0 0 1) + pos, scaled
    q = tmp_path / "m.tri"
    q.write_text("0 0 1 0 0 1\n\n  2 2 3 2 2 3  \n")
    t2 = O.parse_tri2d(str(q), pos=(1, 1, 0), scale=2.0)
    assert t2.shape == (2, 9)
    assert np.array_equal(t2[0], [2, 2, 0, 4, 2, 0, 2, 4, 0])


def test_path_free_follows_reference_sampling():
    # samples index = 1 .. < dist/0.1, zero rotation, end points excluded, first hit stops the edge
    sc = common.scenario("dense3d")
    w = O.World(sc["env"], sc["robot"], O.TRIG_PORTABLE)
    L = O.lib()
    a = common.poses_near_surface(sc["env"], 200, 3, 2.0)
    rs = np.random.RandomState(4)
    b = a.copy()
    b[:, :3] += rs.normal(0, 6, (200, 3))
    n_hit = 0
    for i in range(200):
        free, fh, ns = w.path_free(a[i], b[i])
        parts = L.sffo_distance(O.dp(a[i]), O.dp(b[i])) / 0.1
        idx = [k for k in range(1, 100000) if k < parts]
        assert ns == len(idx)
        first = -1
        for k in idx:
            p = np.zeros(6)
            p[:3] = a[i, :3] + k * (b[i, :3] - a[i, :3]) / parts
            if w.collide(p):
                first = k
                break
        assert (free, fh) == (int(first < 0), first)
        n_hit += first > 0
    assert n_hit > 10


def small_run(name, wave, optimize, trig, iters=1500, seed=11, rrt=False):
    sc = common.scenario(name)
    w = O.World(sc["env"], sc["robot"], trig)
    roots = sc["xml_points"][:4] if sc["xml_points"] is not None else common.free_roots(w.collide, sc["limits"], 4, seed=seed,
                                                                                        dim=sc["dim"])
    if rrt:
        r = O.Rrt(w, roots[:1] if optimize else roots, sc["limits"], sc["dist_tree"], sc["sampling_dist"], dim=sc["dim"],
                  optimize=optimize, max_iterations=iters, seed=seed, trig=trig)
        r.run()
        return r
    f = O.Forest(w, roots, sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=sc["dim"],
                 optimize=optimize, max_iterations=iters, wave=wave, seed=seed, trig=trig)
    f.run()
    return f


def test_wave_one_is_deterministic_and_waves_differ_only_by_definition():
    a = small_run("triang", 1, False, O.TRIG_PORTABLE)
    b = small_run("triang", 1, False, O.TRIG_PORTABLE)
    assert a.fingerprint() == b.fingerprint()
    c = small_run("triang", 64, False, O.TRIG_PORTABLE)
    assert c.stats()["iterations"] == a.stats()["iterations"] == 1500
    assert c.stats()["waves"] < a.stats()["waves"]


def test_portable_and_libm_trig_give_the_same_topology():
    # the two trig modes differ by at most 1 ulp per sample coordinate; on these runs no decision flips
    for name, opt in (("triang", False), ("dense3d", True)):
        a = small_run(name, 1, opt, O.TRIG_LIBM)
        b = small_run(name, 1, opt, O.TRIG_PORTABLE)
        na, nb = a.nodes(), b.nodes()
        assert np.array_equal(na["parent"], nb["parent"]) and np.array_equal(na["tree"], nb["tree"])
        assert np.array_equal(na["iter"], nb["iter"])
        assert np.max(np.abs(na["pos"] - nb["pos"])) < 1e-9
        assert np.allclose(na["cost"], nb["cost"], rtol=1e-12)


RUNS = [("triang", 1, False, False), ("triang", 32, True, False), ("dense3d", 16, False, False),
        ("dense2d", 8, False, False), ("building", 64, True, False), ("triang", 1, False, True), ("building", 1, True, True)]


def test_collision_boolean_does_not_depend_on_the_trig_implementation():
    """The rotation matrix of a pose differs by <= 1 ulp between glibc's and the portable cos / sin; the triangle
    contact boolean does not move: 60 k poses scattered around the obstacle surfaces (both outcomes, many grazing)."""
    for name, n in (("dense3d", 30000), ("triang", 30000)):
        sc = common.scenario(name)
        wl = O.World(sc["env"], sc["robot"], O.TRIG_LIBM)
        wp = O.World(sc["env"], sc["robot"], O.TRIG_PORTABLE)
        poses = np.vstack([common.poses_near_surface(sc["env"], n * 2 // 3, 31, 0.4 * sc["scale"]),
                           common.poses_near_surface(sc["env"], n // 3, 32, 0.02 * sc["scale"])])
        a, b = wl.collide_many(poses), wp.collide_many(poses)
        assert np.array_equal(a, b)
        assert 0.1 < a.mean() < 0.9


def test_libm_divergence_record_is_reproducible(golden_dir):
    """tests/golden/libm_runs.json records, for 24 seeds x 2 maps, whether PORTABLE and LIBM sampling give the same
    forest; three of the entries are recomputed here."""
    import json
    import os
    path = os.path.join(golden_dir, "libm_runs.json")
    if not os.path.exists(path):
        pytest.skip("tests/golden/libm_runs.json not generated")
    rec = json.load(open(path))["divergence"]
    assert len(rec) >= 40
    for e in rec[:2] + rec[24:25]:
        sc = common.scenario(e["map"])
        w = O.World(sc["env"], sc["robot"], O.TRIG_PORTABLE)
        roots = sc["xml_points"][:5] if sc["xml_points"] is not None else common.free_roots(w.collide, sc["limits"], 5, seed=1)
        fp = []
        for trig in (O.TRIG_PORTABLE, O.TRIG_LIBM):
            f = O.Forest(w, roots, sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=sc["dim"],
                         max_iterations=20000, wave=1, seed=e["seed"], trig=trig)
            f.run()
            fp.append((f.fingerprint(), f.stats()["n_nodes"]))
        assert (fp[0][0] == fp[1][0]) == e["bit_identical"]
        assert fp[0][1] == e["nodes"]


def test_oracle_regression_fixture(golden_dir):
    """Pins today's oracle behaviour (tests/golden/make_oracle_runs.py regenerates the file)."""
    with open(os.path.join(golden_dir, "oracle_runs.json")) as f:
        gold = json.load(f)
    for name, wave, opt, rrt in RUNS:
        key = "%s/w%d/%s/%s" % (name, wave, "star" if opt else "plain", "rrt" if rrt else "sff")
        r = small_run(name, wave, opt, O.TRIG_PORTABLE, iters=1200, seed=13, rrt=rrt)
        s = r.stats()
        n = r.nodes()
        got = {"n_nodes": int(s["n_nodes"]), "iterations": int(s["iterations"]), "collide_calls": int(s["collide_calls"]),
               "parent_sum": int(n["parent"].astype(np.int64).sum()), "cost_sum": float(n["cost"].sum()).hex()}
        assert got == gold[key], key


def test_rrt_smoothing_only_shortens_link_plans():
    """RapidExpTree::smoothPaths (src/rrt.h:354-379) works on the central tree's link plans: every smoothed plan is
    a subsequence of the raw one with the same end points, its new edges are collision-free, and the matrix
    (copies made in getPaths, src/rrt.h:350) is untouched."""
    sc = common.scenario("triang")
    w = O.World(sc["env"], sc["robot"], O.TRIG_LIBM)
    pts = sc["xml_points"]
    r = O.Rrt(w, pts[:3], sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=6,
              max_iterations=2500, seed=5)
    r.run()
    d0, c0 = r.paths(3)
    raw = [r.plan(a, b) for a in range(3) for b in range(a + 1, 3)]
    nodes = r.nodes()
    plans = r.smooth()
    assert len(plans) >= 1
    shortened = 0
    for p in plans:
        cands = [q for q in raw if len(q) and q[0] == p[0] and q[-1] == p[-1]]
        assert cands, "a link plan keeps its end points"
        q = list(cands[0])
        it = iter(q)
        assert all(any(x == y for y in it) for x in p), "subsequence of the raw plan"
        shortened += len(p) < len(q)
        for a, b in zip(p[:-1], p[1:]):
            if abs(q.index(a) - q.index(b)) > 1:
                assert w.path_free(nodes["pos"][a], nodes["pos"][b])[0]
    d1, c1 = r.paths(3)
    assert np.array_equal(d0, d1) and c0 == c1
    assert shortened >= 1


def test_lazy_edge_rrt_oracle_properties():
    """LazyTSP::runRRT restatement (src/lazy.h:160-284): the plan runs root -> ... -> the node that came within
    treeDistance of the goal, the edge's distance is that node's cost plus its distance to the goal (:262), an
    unsolved edge reports DBL_MAX (:280), and rng_skip continues one engine stream across sessions."""
    sc = common.scenario("dense2d")
    w = O.World(sc["env"], sc["robot"])
    pts = sc["xml_points"]
    kw = dict(dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=2, seed=9, lazy_edge=True)
    r = O.Rrt(w, pts[0:1], sc["limits"], goal=pts[1], max_iterations=4000, **kw)
    r.run()
    st, n, plan = r.stats(), r.nodes(), r.lazy_plan()
    assert st["solved"] == 1 and st["n_live_trees"] == 1 and st["merges"] == 0
    assert plan[0] == 0 and all(n["parent"][plan[k + 1]] == plan[k] for k in range(len(plan) - 1))
    last = plan[-1]
    gd = np.sqrt(((n["pos"][last] - pts[1]) ** 2).sum())
    assert gd < sc["dist_tree"] and st["lazy_distance"] == gd + n["cost"][last]
    assert last == st["n_nodes"] - 1                      # the search stops with the node that reached the goal
    # too few iterations: unsolved, DBL_MAX, no plan
    r2 = O.Rrt(w, pts[0:1], sc["limits"], goal=pts[1], max_iterations=50, **kw)
    r2.run()
    s2 = r2.stats()
    assert s2["solved"] == 0 and s2["iterations"] == 50 and s2["lazy_distance"] > 1e300 and len(r2.lazy_plan()) == 0
    # the stream: a session started with rng_skip = the words the first one consumed draws what a fresh engine
    # draws after discarding that many words (the RRT* variant consumes the same words per iteration)
    r3 = O.Rrt(w, pts[2:3], sc["limits"], goal=pts[1], max_iterations=300, rng_skip=st["rng_draws"], **kw)
    r3.run()
    assert r3.stats()["rng_draws"] == st["rng_draws"] + 2 * 300 or r3.stats()["solved"] == 1
