"""GPU parity: every hot-path entry point of libsffgpu.so (through the C ABI) against the CPU
oracle on the same seeded inputs.  Integer / boolean / index results must be identical and the
fp64 values bit-equal (both sides evaluate the same IEEE expressions, -ffp-contract=off)."""
import os
import numpy as np
import pytest

import common
import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    import space_filling_forest_star_amd as S
    return S


@pytest.fixture(scope="module")
def ctx(S):
    c = S.Context(0)
    yield c
    c.close()


def load_world(ctx, name):
    sc = common.scenario(name)
    ctx.upload_env(sc["env"])
    ctx.upload_robot(sc["robot"])
    w = O.World(sc["env"], sc["robot"], O.TRIG_PORTABLE)
    return sc, w


@pytest.mark.parametrize("dim", [6, 2])
def test_sample_steer_bit_exact(ctx, dim):
    lim = [-60, 2060, -60, 2110, 0, 1000]
    rs = np.random.RandomState(3)
    n = 20000
    words = rs.randint(0, 2**63, size=(n, 6), dtype=np.uint64) * np.uint64(2) + rs.randint(0, 2, (n, 6)).astype(np.uint64)
    words[0] = 0
    words[1] = np.uint64(2**64 - 1)
    cen = common.random_poses(lim, n, 4, dim)
    cen[:, 3:] *= 2.2  # angles outside [-pi, pi) too (they drift un-normalised in the reference)
    if dim == 2:
        cen[:, 3:] = 0
    out, ok = ctx.sample_steer(words, cen, 14.0, dim, lim)
    L = O.lib()
    ref = np.zeros(6)
    limv = O.f64(lim)
    bad = 0
    for i in range(n):
        r = L.sffo_sample_from_words(words[i].ctypes.data_as(O.c_u64p), O.dp(cen[i]), 14.0, dim, O.dp(limv),
                                     O.TRIG_PORTABLE, O.dp(ref))
        if r != ok[i] or not np.array_equal(ref, out[i]):
            bad += 1
    assert bad == 0


@pytest.mark.parametrize("name", ["dense3d", "triang", "dense2d"])
def test_collide_poses_match_oracle(ctx, name):
    sc, w = load_world(ctx, name)
    dim = sc["dim"]
    spread = 0.4 * sc["scale"]
    poses = np.vstack([common.random_poses(sc["limits"], 4000, 5, dim),
                       common.poses_near_surface(sc["env"], 12000, 6, spread, dim),
                       common.poses_near_surface(sc["env"], 4000, 7, 0.02 * sc["scale"], dim)])
    got = ctx.collide_poses(poses)
    want = w.collide_many(poses)
    assert np.array_equal(got, want)
    assert 0.05 < want.mean() < 0.95  # the set really exercises both outcomes
    # oracle's hierarchy against its own brute force on a subset
    sub = poses[::40]
    assert np.array_equal(np.array([w.collide_brute(p) for p in sub], np.uint8), want[::40])


@pytest.mark.parametrize("name", ["dense3d", "triang", "dense2d"])
def test_collide_segments_match_oracle(ctx, name):
    sc, w = load_world(ctx, name)
    dim = sc["dim"]
    n = 3000
    a = common.poses_near_surface(sc["env"], n, 8, (3.0 if dim == 6 else 30.0) * sc["scale"], dim)
    rs = np.random.RandomState(9)
    d = rs.normal(0, 1, (n, 3))
    if dim == 2:
        d[:, 2] = 0
    d /= np.linalg.norm(d, axis=1)[:, None]
    b = a.copy()
    b[:, :3] += d * sc["sampling_dist"] * rs.uniform(0.01, 1.5, (n, 1))
    if dim == 6:
        b[:, 3:] = a[:, 3:] + rs.uniform(-0.5, 0.5, (n, 3))
    # degenerate edges: zero length, shorter than one step
    b[0] = a[0]
    b[1, :3] = a[1, :3] + 1e-3
    free, fh, ns = ctx.collide_segments(a, b)
    for i in range(n):
        r, first, cnt = w.path_free(a[i], b[i])
        assert (free[i], fh[i], ns[i]) == (r, first, cnt), i
    assert 0.02 < free.mean() < 0.98


def test_radius_and_knn_exact(ctx):
    rs = np.random.RandomState(11)
    n = 30000
    pts = common.random_poses([0, 400, 0, 400, 0, 400], n, 12)
    pts[:, 3:] *= 1.5
    tree = rs.randint(0, 7, n).astype(np.int32)
    ctx.nodes_reset(n)
    ctx.nodes_append(pts[:20000], tree[:20000])
    ctx.nodes_append(pts[20000:], tree[20000:])
    assert ctx.nodes_count() == n
    q = common.random_poses([0, 400, 0, 400, 0, 400], 64, 13)
    L = O.lib()
    idx, dist, cnt = ctx.radius(q, 30.0, cap=512)
    for i in range(len(q)):
        ri = np.zeros(4096, np.int32)
        rd = np.zeros(4096)
        k = L.sffo_radius(O.dp(pts), n, O.dp(q[i]), 30.0, O.ip(ri), O.dp(rd), 4096)
        assert cnt[i] == k
        assert np.array_equal(idx[i, :k], ri[:k])
        assert np.array_equal(dist[i, :k], rd[:k])
    # per-tree + max_id filters, k nearest
    for t, mx in ((3, n), (5, 15000)):
        idx, dist, cnt = ctx.knn(q, 32, tree=t, max_id=mx)
        sel = np.where((tree == t) & (np.arange(n) < mx))[0]
        sub = np.ascontiguousarray(pts[sel])
        for i in range(len(q)):
            ri = np.zeros(32, np.int32)
            rd = np.zeros(32)
            k = L.sffo_knn(O.dp(sub), len(sub), O.dp(q[i]), 32, O.ip(ri), O.dp(rd))
            assert cnt[i] == k
            assert np.array_equal(idx[i, :k], sel[ri[:k]])
            assert np.array_equal(dist[i, :k], rd[:k])
    # fewer eligible nodes than k
    idx, dist, cnt = ctx.knn(q[:4], 32, tree=2, max_id=40)
    want = int(np.sum(tree[:40] == 2))
    assert np.all(cnt == want)


def test_indexed_knn_equals_the_linear_sweep(ctx):
    """sffgpu_nodes_index (the Index::buildIndex counterpart): with the grid over the store, sffgpu_knn answers from the
    cells around each query (k_knn_grid) - the same exact lists as the per-query sweep (k_knn_linear) and the oracle,
    also for nodes appended after the index was built, queries outside the limits and k larger than a sparse corner holds."""
    rs = np.random.RandomState(11)
    lim = np.array([-60.0, 2060.0, -60.0, 2110.0, 0.0, 1000.0])
    n = 60000
    pts = np.empty((n, 6))
    for a in range(3):
        pts[:, a] = rs.uniform(lim[2 * a], lim[2 * a + 1], n)
    pts[:4000, :3] = rs.normal(500.0, 6.0, (4000, 3))      # a dense clump: cells far over their bucket size
    pts[:, 3:] = rs.uniform(-np.pi, np.pi, (n, 3))
    tree = rs.randint(0, 8, n).astype(np.int32)
    q = np.vstack([pts[rs.randint(0, n, 200)] + rs.normal(0, 4.0, (200, 6)),
                   np.array([[-500.0, -500.0, -200.0, 0.1, 0.2, 0.3], [3000.0, 1000.0, 500.0, 0.0, 0.0, 0.0]])])
    ctx.nodes_reset(n + 64)
    ctx.nodes_append(pts[:40000], tree[:40000])
    lin = {k: ctx.knn(q, k) for k in (1, 32, 64)}
    big = ctx.knn(np.tile(q, (13, 1))[:2600], 32)           # (> 2048 queries: k_knn_linear itself)
    assert np.array_equal(big[0][:len(q)], lin[32][0]) and np.array_equal(big[1][:len(q)], lin[32][1])
    ctx.nodes_index(lim, 18.2)
    for k in (1, 32, 64):
        gi, gd, gc = ctx.knn(q, k)
        assert np.array_equal(gc, lin[k][2]) and np.array_equal(gi, lin[k][0]) and np.array_equal(gd, lin[k][1])
    # launches of more than 2048 queries take the one-wavefront-per-query kernels (k_knn_grid with the same shell
    # enumeration and sweep fallback), smaller ones a workgroup per query (k_knn_grid_wg): the same lists
    reps = 2600 // len(q) + 1
    qb = np.tile(q, (reps, 1))[:2600]
    for k in (1, 32):
        bi, bd, bc = ctx.knn(qb, k)
        want = [np.tile(a, (reps,) + (1,) * (a.ndim - 1))[:2600] for a in lin[k]]
        assert np.array_equal(bc, want[2]) and np.array_equal(bi, want[0]) and np.array_equal(bd, want[1])
    ctx.nodes_append(pts[40000:], tree[40000:])             # enters the index in the same launch
    gi, gd, gc = ctx.knn(q, 32)
    L = O.lib()
    for i in range(0, len(q), 7):
        ri = np.zeros(32, np.int32)
        rd = np.zeros(32)
        k = L.sffo_knn(O.dp(np.ascontiguousarray(pts)), n, O.dp(q[i]), 32, O.ip(ri), O.dp(rd))
        assert gc[i] == k and np.array_equal(gi[i, :k], ri[:k]) and np.array_equal(gd[i, :k], rd[:k])
    ctx.nodes_reset(1024)


def run_pair(S, ctx, name, wave, iters, seed, n_roots=5, budget=0, optimize=False, goal_idx=None, priority_bias=0.0):
    sc, w = load_world(ctx, name)
    if sc["xml_points"] is not None:
        roots = sc["xml_points"][:n_roots]
    else:
        roots = common.free_roots(w.collide, sc["limits"], n_roots, seed=seed, dim=sc["dim"])
    kw = dict(dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=sc["dim"], max_iterations=iters,
              node_budget=budget, wave=wave, seed=seed, optimize=optimize, priority_bias=priority_bias)
    if goal_idx is not None:   # a goal a few steps away from the first root (free in triang / building)
        g = roots[0].copy()
        g[:3] += np.array(goal_idx, dtype=np.float64)
        kw["goal"] = g
    fo = O.Forest(w, roots, sc["limits"], **kw)
    fo.run()
    fg = S.Forest(ctx, roots, sc["limits"], **kw)
    fg.run()
    return fo, fg


def assert_same_forest(fo, fg):
    so, sg = fo.stats(), fg.stats()
    for k in ("iterations", "solved", "n_nodes", "n_trees", "frontier_size", "closed_size", "n_connected", "n_borders",
              "collide_calls", "path_free_calls", "nn_queries", "waves"):
        assert so[k] == sg[k], (k, so[k], sg[k])
    no, ng = fo.nodes(), fg.nodes()
    for k in ("parent", "tree", "iter"):
        assert np.array_equal(no[k], ng[k]), k
    for k in ("pos", "cost", "dpar"):
        assert np.array_equal(no[k], ng[k]), k  # bit-exact fp64
    bo, bg = fo.borders(), fg.borders()
    for k in bo:
        assert np.array_equal(bo[k], bg[k]), k
    assert fo.fingerprint() == fg.fingerprint()


@pytest.mark.parametrize("name,wave,iters", [
    ("dense3d", 1, 1500), ("dense3d", 16, 4000), ("dense3d", 256, 12000),
    ("dense3d_coarse", 1, 1500), ("dense3d_coarse", 64, 6000),
    ("triang", 1, 1500), ("triang", 128, 8000),
    ("dense2d", 1, 1500), ("dense2d", 32, 4000),
])
def test_forest_topology_identical(S, ctx, name, wave, iters):
    fo, fg = run_pair(S, ctx, name, wave, iters, seed=2)
    assert fo.stats()["n_nodes"] > 50
    assert_same_forest(fo, fg)


@pytest.mark.parametrize("name,wave,iters", [
    ("dense3d", 1, 1200), ("dense3d", 64, 5000), ("triang", 1, 1500), ("triang", 256, 10000),
    ("dense3d_coarse", 32, 3000), ("building", 128, 6000), ("dense2d", 16, 2500),
])
def test_sff_star_rewire_identical(S, ctx, name, wave, iters):
    """SFF* (optimize=true): choose-parent + rewire (src/forest.h:307-351) — parents, costs and the
    reference-equivalent call counters must equal the oracle's sequential replay."""
    fo, fg = run_pair(S, ctx, name, wave, iters, seed=5, optimize=True)
    assert fo.stats()["n_nodes"] > 50
    assert_same_forest(fo, fg)
    # rewiring really happened: some node's parent is younger than the node itself
    no = fo.nodes()
    assert np.any(no["parent"] > np.arange(len(no["parent"])))


@pytest.mark.parametrize("name", ["building"])
def test_building_map_collision(ctx, name):
    sc, w = load_world(ctx, name)
    poses = np.vstack([common.random_poses(sc["limits"], 2000, 15), common.poses_near_surface(sc["env"], 6000, 16, 2.0)])
    assert np.array_equal(ctx.collide_poses(poses), w.collide_many(poses))
    fo, fg = run_pair(__import__("space_filling_forest_star_amd"), ctx, name, 64, 3000, seed=6)
    assert_same_forest(fo, fg)


@pytest.mark.parametrize("name,wave,n_roots,optimize", [("triang", 1, 1, False), ("triang", 64, 2, False),
                                                         ("triang", 32, 1, True), ("building", 128, 1, False)])
def test_forest_single_goal_mode(S, ctx, name, wave, n_roots, optimize):
    """Problem::hasGoal (src/forest.h:91-109, :286-287, :369-372): the goal is a one-node tree that is
    searched but never expanded; reaching it ends the run."""
    fo, fg = run_pair(S, ctx, name, wave, 20000, seed=8, n_roots=n_roots, optimize=optimize, goal_idx=[12, 8, 5])
    assert_same_forest(fo, fg)
    assert fo.stats()["solved"] == 1 and fo.stats()["n_borders"] >= 1


@pytest.mark.parametrize("optimize", [False, True])
def test_path_costs_and_plans_match(S, ctx, optimize):
    """getPaths + getAllPaths: the root-to-root cost matrix (what params.csv / the TSP file print) and the
    node plans.  BASELINE north_star allows 1e-5 relative on path costs; both sides run the same fp64
    expressions, so they are compared for equality."""
    fo, fg = run_pair(S, ctx, "dense3d_coarse", 64, 6000, seed=2, optimize=optimize)
    assert_same_forest(fo, fg)
    do = fo.paths()
    dg, conn = fg.paths()
    assert len(conn) >= 3
    finite = do < 1e300
    assert finite.sum() > len(do)
    assert np.array_equal(finite, dg < 1e300)
    assert np.array_equal(do[finite], dg[finite])
    assert np.allclose(do[finite], dg[finite], rtol=1e-5)
    for i in range(len(do)):
        for j in range(i + 1, len(do)):
            assert np.array_equal(fo.plan(i, j), fg.plan(i, j))
    # smoothPaths (src/forest.h:464-511): shortcutting with batched edge checks gives the same plans and costs
    so, sg = fo.smooth(), fg.smooth()
    assert np.array_equal(so[finite], sg[finite])
    assert np.all(so[finite] <= do[finite] + 1e-9) and np.any(so[finite] < do[finite] - 1e-6)
    for i in range(len(do)):
        for j in range(i + 1, len(do)):
            assert np.array_equal(fo.plan(i, j), fg.plan(i, j))
    assert fo.stats()["collide_calls"] == fg.stats()["collide_calls"]


@pytest.mark.parametrize("name,wave,n_roots,optimize,goal", [
    ("dense3d_coarse", 1, 5, False, None), ("dense3d_coarse", 64, 5, False, None), ("triang", 128, 4, True, None),
    ("triang", 1, 1, False, [12, 8, 5]), ("triang", 32, 2, False, [12, 8, 5]), ("dense2d", 16, 3, False, None),
])
def test_priority_frontier_mode(S, ctx, name, wave, n_roots, optimize, goal, monkeypatch):
    """Problem::priorityBias = 0.95 (what the reference's example XMLs set): frontier nodes come from the
    per-tree priority heaps of src/heap.h (best node w.p. bias, random heap position otherwise)."""
    fo, fg = run_pair(S, ctx, name, wave, 5000, seed=14, n_roots=n_roots, optimize=optimize, goal_idx=goal,
                      priority_bias=0.95)
    assert fo.stats()["n_nodes"] > 5
    assert_same_forest(fo, fg)
    # the heaps live on the device (devprio.hip) unless there is a goal or the wave is the reference's one-slot loop
    assert bool(fg.device_engine()) == (goal is None and wave >= 2), (fg.device_engine(), goal, wave)
    # and it is not the plain mode in disguise
    fp, _ = run_pair(S, ctx, name, wave, 5000, seed=14, n_roots=n_roots, optimize=optimize, goal_idx=goal)
    assert fp.fingerprint() != fo.fingerprint()


@pytest.mark.parametrize("name,wave,n_roots,optimize,iters", [
    ("dense3d", 1024, 10, False, 40000), ("dense3d", 4096, 10, False, 45000), ("dense3d", 512, 6, True, 20000),
    ("triang", 2048, 5, False, 30000), ("dense3d_coarse", 256, 8, False, 20000),
])
def test_priority_frontier_mode_on_the_device_engine(S, ctx, name, wave, n_roots, optimize, iters, monkeypatch):
    """The priority-frontier mode at the wave sizes it is run at: thousands of pops per wave (minimum / random heap entry,
    src/heap.h:175-238), pushes of the accepted nodes onto every heap of their tree, exhausted nodes leaving the tree's
    other heaps (src/forest.h:160-181), heaps running empty (dense3d_coarse saturates: closed-list waves and back).  The
    heap ARRAY order is part of the result (random entries are taken by index): the forest must equal the oracle's."""
    monkeypatch.setenv("SFFGPU_PRIO_DEVICE", "1")
    fo, fg = run_pair(S, ctx, name, wave, iters, seed=21, n_roots=n_roots, optimize=optimize, priority_bias=0.95)
    assert fg.device_engine()
    assert fo.stats()["n_nodes"] > 500
    assert_same_forest(fo, fg)
    # (the heaps crowd the samples together: at these small waves a bounded device list runs over now and then and the
    # wave is finished by the host engine - heaps, slots and the wave's pending pushes travel with it)


def test_priority_frontier_mode_default_engine_choice(S, ctx):
    """Every wave size but the reference's one-slot loop runs on the device engine; SFFGPU_PRIO_DEVICE=0 keeps the mode on
    the host-replay engine."""
    sc, w = load_world(ctx, "dense3d")
    roots = common.free_roots(w.collide, sc["limits"], 4, seed=2)
    kw = dict(dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=6, max_iterations=100, seed=2, priority_bias=0.9)
    one = S.Forest(ctx, roots, sc["limits"], wave=1, **kw)
    assert not one.device_engine()
    one.close()
    big = S.Forest(ctx, roots, sc["limits"], wave=4096, **kw)
    assert big.device_engine()
    big.close()
    os.environ["SFFGPU_PRIO_DEVICE"] = "0"
    try:
        off = S.Forest(ctx, roots, sc["limits"], wave=4096, **kw)
        assert not off.device_engine()
        off.close()
    finally:
        os.environ.pop("SFFGPU_PRIO_DEVICE")


def test_priority_frontier_mode_sequential_picks(S, ctx, monkeypatch):
    """The picks one slot after the other (k_prio_begin: what runs when the parallel plan is not valid - a heap asked for
    more nodes than it holds, a draw in the rejection zone) give the same forest."""
    monkeypatch.setenv("SFFGPU_PRIO_SEQ", "1")
    monkeypatch.setenv("SFFGPU_PRIO_DEVICE", "1")
    fo, fg = run_pair(S, ctx, "dense3d", 512, 15000, seed=22, n_roots=6, priority_bias=0.95)
    assert fg.device_engine()
    assert_same_forest(fo, fg)


def test_priority_frontier_mode_staged_runs(S, ctx, monkeypatch):
    """run() in pieces (the mirror is read between the pieces) and a hand-over to the host engine and back: the heaps
    travel with the state."""
    monkeypatch.setenv("SFFGPU_PRIO_DEVICE", "1")
    sc, w = load_world(ctx, "dense3d")
    roots = common.free_roots(w.collide, sc["limits"], 6, seed=9)
    kw = dict(dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=6, wave=512, seed=9, priority_bias=0.95)
    fo = O.Forest(w, roots, sc["limits"], max_iterations=30000, **kw)
    fo.run()
    fg = S.Forest(ctx, roots, sc["limits"], max_iterations=30000, **kw)
    assert fg.device_engine()
    for k in range(6):
        fg.run(3)
        assert fg.stats()["n_nodes"] > 0      # (reads the mirror: heaps and slots come down and go up again)
    fg.run()
    assert_same_forest(fo, fg)


@pytest.mark.parametrize("name,optimize,n_roots,iters", [
    ("dense3d", False, 1, 3000), ("dense3d", True, 1, 1500), ("dense3d", False, 10, 6000),
])
def test_rrt_waves_with_repaired_slots_identical(S, ctx, monkeypatch, name, optimize, n_roots, iters):
    """A speculative RRT wave (csrc/rrt.cpp run_wave) used to end at the first slot whose nearest node would be a new point
    of the same wave.  Such a slot is now evaluated a second time FROM that point (k_rrt_mates names it, Ctx::rrt_chain_alt
    evaluates the repaired row) and the replay takes that row when the earlier slot became a node as speculated: the
    committed sequence stays the reference's (src/rrt.h:128-322) - nodes, parents, costs, links, counters, stream position
    equal the oracle's at every wave size - and the job takes fewer waves than with the repair switched off."""
    sc, w = load_world(ctx, name)
    pts = sc["xml_points"] if sc["xml_points"] is not None else common.free_roots(w.collide, sc["limits"], 10, seed=5,
                                                                                 dim=sc["dim"])
    roots = pts[:n_roots]
    kw = dict(dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=sc["dim"], optimize=optimize,
              max_iterations=iters, seed=7)
    ro = O.Rrt(w, roots, sc["limits"], **kw)
    ro.run()
    so, no, lo = ro.stats(), ro.nodes(), ro.links()
    assert so["n_nodes"] > 300
    waves = {}
    for repair in ("1", "0"):
        monkeypatch.setenv("SFFGPU_RRT_REPAIR", repair)
        for wave in ((0, 256, 4096) if repair == "1" else (0,)):
            rg = S.Rrt(ctx, roots, sc["limits"], wave=wave, **kw)
            rg.run()
            sg = rg.stats()
            for k in so:
                assert so[k] == sg[k], (repair, wave, k, so[k], sg[k])
            ng, lg = rg.nodes(), rg.links()
            for k in no:
                assert np.array_equal(no[k], ng[k]), (repair, wave, k)
            for k in lo:
                assert np.array_equal(lo[k], lg[k]), (repair, wave, k)
            waves[(repair, wave)] = sg["waves"]
            rg.close()
    assert waves[("1", 0)] < waves[("0", 0)], waves
    assert 2 * waves[("1", 0)] < waves[("0", 0)] or n_roots > 1 or iters < 2000, waves


def test_rrt_wave_engine_variants_agree_on_random_configurations():
    """profiles/rrt_stress.py, one configuration per map: the RRT session with its default wave engine (one chain, repaired slots, the
    replay's walk done ahead, edges in two batches), without the repair, without the dry walk, without the chain and - small
    budgets - one iteration per round trip must commit the same nodes, parents, costs, links, counters and stream position
    on random maps / root counts / goal biases / wave sizes (no oracle in the loop: the oracle pins the engines elsewhere)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, STRESS_SEEDS="2")
    out = subprocess.run([sys.executable, os.path.join(root, "profiles", "rrt_stress.py")], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    last = [ln for ln in out.stdout.splitlines() if ln.startswith("cases ")]
    assert last and "mismatches 0," in last[-1] and not last[-1].startswith("cases 0,"), out.stdout[-2000:]


def test_gpu_matches_committed_golden_runs(S, ctx, golden_dir):
    """The GPU path against the COMMITTED fixture tests/golden/oracle_runs.json (not only against the oracle
    built on this box): node counts, reference-equivalent collision calls, parent and cost checksums."""
    import json
    import os
    from test_oracle_cpu import RUNS
    gold = json.load(open(os.path.join(golden_dir, "oracle_runs.json")))
    for name, wave, opt, rrt in RUNS:
        key = "%s/w%d/%s/%s" % (name, wave, "star" if opt else "plain", "rrt" if rrt else "sff")
        sc, w = load_world(ctx, name)
        roots = sc["xml_points"][:4] if sc["xml_points"] is not None else common.free_roots(w.collide, sc["limits"], 4, seed=13,
                                                                                            dim=sc["dim"])
        if rrt:
            r = S.Rrt(ctx, roots[:1] if opt else roots, sc["limits"], sc["dist_tree"], sc["sampling_dist"], dim=sc["dim"],
                      optimize=opt, max_iterations=1200, seed=13)
        else:
            r = S.Forest(ctx, roots, sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=sc["dim"],
                         optimize=opt, max_iterations=1200, wave=wave, seed=13)
        r.run()
        s, n = r.stats(), r.nodes()
        got = {"n_nodes": int(s["n_nodes"]), "iterations": int(s["iterations"]), "collide_calls": int(s["collide_calls"]),
               "parent_sum": int(n["parent"].astype(np.int64).sum()), "cost_sum": float(n["cost"].sum()).hex()}
        assert got == gold[key], key
        r.close()


@pytest.mark.parametrize("fixture", ["full_size_run_w16384.json", "full_size_run_w65536.json"])
def test_full_size_run_at_the_bench_wave_equals_the_oracle(S, ctx, golden_dir, fixture):
    """The headline job exactly as bench.py runs it (waves of 16 384 slots, to the 1 M-node budget) against the CPU
    oracle's run of the same configuration (tests/golden/full_size_run_w16384.json, FULL_SIZE_WAVE=16384
    FULL_SIZE_WAVES=0 tests/golden/make_full_size.py): fingerprint over every node, counters, checksums.  The same at
    waves of 65 536 slots - the wave that grows with the rank count (8 ranks x 8 192 slots; bench.py --force-dist
    --scaled-wave 65536), the largest the device engine takes."""
    import json
    import os
    path = os.path.join(golden_dir, fixture)
    if not os.path.exists(path):
        pytest.skip("tests/golden/%s not generated" % fixture)
    g = json.load(open(path))
    sc, w = load_world(ctx, "dense3d")
    roots = common.free_roots(lambda p: int(ctx.collide_poses(p[None, :])[0]), sc["limits"], 10, seed=1)
    f = S.Forest(ctx, roots, sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=6,
                 max_iterations=2**31 - 1, node_budget=1000000, wave=g["wave"], seed=1)
    f.run()
    s, n = f.stats(), f.nodes()
    got = {"waves": int(s["waves"]), "fingerprint": "%016x" % f.fingerprint(), "n_nodes": int(s["n_nodes"]),
           "iterations": int(s["iterations"]), "collide_calls": int(s["collide_calls"]),
           "path_free_calls": int(s["path_free_calls"]), "nn_queries": int(s["nn_queries"]), "n_borders": int(s["n_borders"]),
           "parent_sum": int(n["parent"].astype(np.int64).sum()), "cost_sum": float(n["cost"].sum()).hex()}
    for k in got:
        assert got[k] == g[k], (k, got[k], g[k])
    assert got["n_nodes"] >= 1000000
    f.close()


def test_full_size_bench_job_in_the_reference_pinned_arithmetic(S, ctx, golden_dir):
    """The headline job (bench.py: waves of 16 384 slots to the 1 M-node budget) with sffgpu_forest_cfg::libm_sampling on
    the DEVICE-RESIDENT engine: the transcendental functions of RandGen::randomPointInDistance are evaluated by the host's
    C library (the arithmetic tests/golden/ref_primitives.json pins to the reference's own randGen.h) and travel beside
    the engine words; everything else runs as in the bench.  Must reproduce the CPU oracle's TRIG_LIBM run of the same job
    (tests/golden/full_size_run_w16384_libm.json; FULL_SIZE_TRIG=libm FULL_SIZE_WAVE=16384 FULL_SIZE_WAVES=0
    tests/golden/make_full_size.py) fingerprint for fingerprint."""
    import json
    import os
    path = os.path.join(golden_dir, "full_size_run_w16384_libm.json")
    if not os.path.exists(path):
        pytest.skip("tests/golden/full_size_run_w16384_libm.json not generated")
    g = json.load(open(path))
    assert g["sampling_trig"] == "libm"
    sc, w = load_world(ctx, "dense3d")
    roots = common.free_roots(lambda p: int(ctx.collide_poses(p[None, :])[0]), sc["limits"], 10, seed=1)
    f = S.Forest(ctx, roots, sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=6,
                 max_iterations=2**31 - 1, node_budget=1000000, wave=g["wave"], seed=1, libm_sampling=True)
    assert f.device_engine()
    f.run()
    s, n = f.stats(), f.nodes()
    got = {"waves": int(s["waves"]), "fingerprint": "%016x" % f.fingerprint(), "n_nodes": int(s["n_nodes"]),
           "iterations": int(s["iterations"]), "collide_calls": int(s["collide_calls"]),
           "path_free_calls": int(s["path_free_calls"]), "nn_queries": int(s["nn_queries"]), "n_borders": int(s["n_borders"]),
           "parent_sum": int(n["parent"].astype(np.int64).sum()), "cost_sum": float(n["cost"].sum()).hex()}
    for k in got:
        assert got[k] == g[k], (k, got[k], g[k])
    f.close()
    # and what the kernels' own (portable, <= 1 ulp from glibc) trig changes at this size: position bits, not the
    # forest's shape - the same nodes with the same parents, trees and iterations of creation
    fp = S.Forest(ctx, roots, sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=6,
                  max_iterations=2**31 - 1, node_budget=1000000, wave=g["wave"], seed=1)
    fp.run()
    npn = fp.nodes()
    assert "%016x" % fp.fingerprint() != g["fingerprint"]
    same_topology = (len(npn["parent"]) == len(n["parent"]) and np.array_equal(npn["parent"], n["parent"])
                     and np.array_equal(npn["tree"], n["tree"]) and np.array_equal(npn["iter"], n["iter"]))
    print("portable vs libm sampling at 1 M nodes: same topology = %s, max |dpos| = %.3g"
          % (same_topology, float(np.abs(npn["pos"] - n["pos"]).max()) if same_topology else float("nan")))
    assert same_topology
    fp.close()


def test_full_size_headline_run_equals_the_oracle(S, ctx, golden_dir):
    """BASELINE's headline configuration at full size (dense_3D, 10 roots, 1 M-node budget, waves of 8192 slots,
    3 + 255 waves = what bench.py times): the GPU run must reproduce the committed summary of the CPU oracle's run
    (tests/golden/full_size_run.json: fingerprint over every node's parent / tree / iteration, counters, cost
    checksum) - also when the waves are split over several calls - and satisfy the size-independent properties
    of a forest: nodes inside the limits, parents older and in the same tree, costs accumulating along the
    parent chain, and - on a sample checked by the oracle - collision-free poses and parent edges."""
    import json
    import os
    path = os.path.join(golden_dir, "full_size_run.json")
    if not os.path.exists(path):
        pytest.skip("tests/golden/full_size_run.json not generated (tests/golden/make_full_size.py)")
    g = json.load(open(path))
    sc, w = load_world(ctx, "dense3d")
    roots = common.free_roots(lambda p: int(ctx.collide_poses(p[None, :])[0]), sc["limits"], 10, seed=1)
    f = S.Forest(ctx, roots, sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=6,
                 max_iterations=2**31 - 1, node_budget=1000000, wave=8192, seed=1)
    f.run(100)
    f.run(g["waves"] - 100)
    s, n = f.stats(), f.nodes()
    got = {"fingerprint": "%016x" % f.fingerprint(), "n_nodes": int(s["n_nodes"]), "iterations": int(s["iterations"]),
           "collide_calls": int(s["collide_calls"]), "path_free_calls": int(s["path_free_calls"]),
           "nn_queries": int(s["nn_queries"]), "n_borders": int(s["n_borders"]),
           "parent_sum": int(n["parent"].astype(np.int64).sum()), "cost_sum": float(n["cost"].sum()).hex()}
    for k in got:
        assert got[k] == g[k], (k, got[k], g[k])
    # size-independent properties
    N = s["n_nodes"]
    pos, par, tree, cost, dpar = n["pos"], n["parent"], n["tree"], n["cost"], n["dpar"]
    lim = np.asarray(sc["limits"], dtype=np.float64)
    for a in range(3):
        assert (pos[:, a] >= lim[2 * a]).all() and (pos[:, a] <= lim[2 * a + 1]).all()
    kids = np.nonzero(par >= 0)[0]
    assert len(kids) == N - 10
    assert (par[kids] < kids).all() and (tree[par[kids]] == tree[kids]).all()
    assert np.array_equal(cost[kids], cost[par[kids]] + dpar[kids])        # src/forest.h:353, same additions
    d3 = np.linalg.norm(pos[kids, :3] - pos[par[kids], :3], axis=1)
    assert (d3 <= dpar[kids] * (1 + 1e-12)).all() and (dpar[kids] <= sc["sampling_dist"] * (1 + 1e-9)).all()
    rs = np.random.RandomState(4)
    for i in rs.choice(kids, 1500, replace=False):
        assert not w.collide(pos[i])
        assert w.path_free(pos[par[i]], pos[i])[0], i
    f.close()


def test_c5_building_sff_star_full_run_properties(S, ctx):
    """BASELINE configs[4] on one GPU at full size: building.obj (26 908 triangles), 20 seeded roots, SFF* (optimize =
    true: choose-parent + rewire), 2 M-node budget, waves of 8192 slots.  With the example's distances (dtree 0.5,
    circum 0.4, scale 10) the forest SATURATES before the budget: every tree connected, frontier empty -> "solved" at
    ~2.1e5 nodes, so the run is the whole job.  Size-independent properties on every node (limits, tree of the
    parent, edge length = stored parent distance, costs along the parent chain: equal where nothing was rewired
    above, never smaller otherwise - the reference leaves descendants' costs untouched),
    oracle-checked poses and parent edges on a sample; the first 150 k nodes' worth of the same run is pinned
    bit for bit by test_baseline_configs_equal_the_oracle."""
    sc, w = load_world(ctx, "building")
    roots = common.free_roots(lambda p: int(ctx.collide_poses(p[None, :])[0]), sc["limits"], 20, seed=1)
    f = S.Forest(ctx, roots, sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=6, optimize=True,
                 max_iterations=2**31 - 1, node_budget=2000000, wave=8192, seed=1)
    f.run()
    s, n = f.stats(), f.nodes()
    N = s["n_nodes"]
    assert s["solved"] == 1 and s["frontier_size"] == 0 and 150000 < N < 2000000 and s["n_connected"] == 20
    pos, par, tree, cost, dpar = n["pos"], n["parent"], n["tree"], n["cost"], n["dpar"]
    lim = np.asarray(sc["limits"], dtype=np.float64)
    for a in range(3):
        assert (pos[:, a] >= lim[2 * a]).all() and (pos[:, a] <= lim[2 * a + 1]).all()
    kids = np.nonzero(par >= 0)[0]
    assert len(kids) == N - 20 and (cost[par < 0] == 0).all()
    assert (tree[par[kids]] == tree[kids]).all()
    rewired_up = (par[kids] > kids).sum()
    assert rewired_up > 1000                                  # rewiring happened: younger parents
    # costs: cost = parent's cost + edge at the moment the edge was made; a later rewire of an ancestor only lowers
    # the ancestor's cost
    slack = cost[kids] - (cost[par[kids]] + dpar[kids])
    assert (slack >= -1e-9).all() and (slack == 0).mean() > 0.5
    # stored parent distance = the 6-D metric of the edge, bit for bit (the oracle's Point::distance)
    L = O.lib()
    rs = np.random.RandomState(9)
    for i in rs.choice(kids, 4000, replace=False):
        assert L.sffo_distance(O.dp(np.ascontiguousarray(pos[i])), O.dp(np.ascontiguousarray(pos[par[i]]))) == dpar[i]
    for i in rs.choice(kids, 600, replace=False):
        assert not w.collide(pos[i])
        # choose-parent checks isPathFree(new, neighbour), rewire isPathFree(neighbour, new), the default parent edge
        # isPathFree(expanded, new) (src/forest.h:246,323,336): one direction of the stored edge was verified
        assert w.path_free(pos[par[i]], pos[i])[0] or w.path_free(pos[i], pos[par[i]])[0], i
    f.close()


def test_sff_star_at_two_million_nodes(S, ctx):
    """The regime BASELINE configs[4] names but its own map never reaches (building.obj saturates at 2e5 nodes): SFF* with
    a 2 M-node store and k = floor(2e log10 N) = 34 neighbours per accepted sample.  dense_3D with a step of 11 (dtree 14)
    has the room: 10 roots, waves of 16 384 slots, optimize = true, 2 M-node budget on the device engine.  The start of the job is pinned bit for bit
    against the oracle (35 k-node budget, same waves); the whole 2 M-node forest is checked through the size-independent
    properties (limits, trees, edge lengths = stored parent distances bit for bit, costs never below parent cost + edge,
    equal where nothing above was rewired, oracle-checked poses and edges on a sample)."""
    sc, w = load_world(ctx, "dense3d")
    roots = common.free_roots(w.collide, sc["limits"], 10, seed=1)
    # (with the bench's step - circum 14 / dtree 18 - dense_3D saturates at 1.05 M nodes: a step of 11 / 14 has room for 2 M)
    kw = dict(dist_tree=14.0, sampling_dist=11.0, dim=6, optimize=True, max_iterations=2**31 - 1, wave=16384, seed=1)
    fo = O.Forest(w, roots, sc["limits"], node_budget=35000, **kw)
    fo.run()
    fg = S.Forest(ctx, roots, sc["limits"], node_budget=35000, **kw)
    assert fg.device_engine()
    fg.run()
    assert_same_forest(fo, fg)
    fg.close()
    f = S.Forest(ctx, roots, sc["limits"], node_budget=2000000, **kw)
    f.run()
    s, n = f.stats(), f.nodes()
    N = s["n_nodes"]
    assert N >= 2000000 and s["host_fallback_waves"] == 0 and s["star_rounds"] > 100
    # the k-nearest sets really were 33-34 members large at the end (k(1.9e6) = 34)
    assert s["star_members"] / (N - 10) > 28
    pos, par, tree, cost, dpar = n["pos"], n["parent"], n["tree"], n["cost"], n["dpar"]
    lim = np.asarray(sc["limits"], dtype=np.float64)
    for a in range(3):
        assert (pos[:, a] >= lim[2 * a]).all() and (pos[:, a] <= lim[2 * a + 1]).all()
    kids = np.nonzero(par >= 0)[0]
    assert len(kids) == N - 10 and (cost[par < 0] == 0).all()
    assert (tree[par[kids]] == tree[kids]).all()
    assert (par[kids] > kids).sum() > 1000                    # rewiring happened: younger parents
    slack = cost[kids] - (cost[par[kids]] + dpar[kids])
    assert (slack >= -1e-9).all() and (slack == 0).mean() > 0.5
    L = O.lib()
    rs = np.random.RandomState(9)
    for i in rs.choice(kids, 4000, replace=False):
        assert L.sffo_distance(O.dp(np.ascontiguousarray(pos[i])), O.dp(np.ascontiguousarray(pos[par[i]]))) == dpar[i]
    for i in rs.choice(kids, 300, replace=False):
        assert not w.collide(pos[i])
        assert w.path_free(pos[par[i]], pos[i])[0] or w.path_free(pos[i], pos[par[i]])[0], i
    print("SFF* dense_3D 2 M nodes: %.2f s, %.2f M nodes/s, %.2f passes per round, %d rewires"
          % (s["total_ms"] / 1e3, (N - 10) / s["total_ms"] / 1e3, s["star_passes"] / s["star_rounds"], s["star_rewires"]))
    f.close()


@pytest.mark.parametrize("fixture", ["c5_full_run.json", "c5_full_run_w8192.json", "c5_full_run_w16384.json",
                                     "c5_full_run_w32768.json"])
def test_c5_building_sff_star_whole_job_equals_the_oracle(S, ctx, golden_dir, fixture):
    """BASELINE configs[4] run to its END (building.obj, 20 seeded roots, SFF* with rewire, 2 M-node budget): the forest
    saturates - frontier empty, every tree connected, "solved" - at about 2e5 nodes, so the WHOLE job is pinned against
    the CPU oracle (tests/golden/make_c5_full.py; waves of 4096, 8192, 16 384 and 32 768 slots - the open list never holds more
    than 16 384 nodes, so the last two are the same job): fingerprint over every node,
    counters, checksums.  Runs on the device-resident engine (rewire fixed point on the GPU, csrc/devstar.hip)."""
    import json
    import os
    path = os.path.join(golden_dir, fixture)
    if not os.path.exists(path):
        pytest.skip("tests/golden/%s not generated" % fixture)
    g = json.load(open(path))
    sc, w = load_world(ctx, "building")
    roots = common.free_roots(w.collide, sc["limits"], 20, seed=1, dim=sc["dim"])
    f = S.Forest(ctx, roots, sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=6, optimize=True,
                 max_iterations=2**31 - 1, node_budget=2000000, wave=g["wave"], seed=1)
    assert f.device_engine()
    f.run()
    s, n = f.stats(), f.nodes()
    got = {"waves": int(s["waves"]), "fingerprint": "%016x" % f.fingerprint(), "n_nodes": int(s["n_nodes"]),
           "iterations": int(s["iterations"]), "collide_calls": int(s["collide_calls"]), "solved": int(s["solved"]),
           "path_free_calls": int(s["path_free_calls"]), "nn_queries": int(s["nn_queries"]), "n_borders": int(s["n_borders"]),
           "frontier_size": int(s["frontier_size"]),
           "parent_sum": int(n["parent"].astype(np.int64).sum()), "cost_sum": float(n["cost"].sum()).hex()}
    for k in got:
        assert got[k] == g[k], (k, got[k], g[k])
    assert got["solved"] == 1 and got["frontier_size"] == 0
    # (waves of 16 384 slots on this map: a few rounds hold a sample with more qualifying neighbours than its record - SFF*
    # reads whole records - and are finished by the host engine; the forest is the oracle's either way)
    if g["wave"] <= 8192:
        assert s["host_fallback_waves"] == 0
    f.close()


def test_baseline_configs_equal_the_oracle(S, ctx, golden_dir):
    """BASELINE.json configs[0] (2-D, 3 roots, 10 k nodes) and configs[1] (triang, 5 roots, 100 k nodes) in full and configs[4] (building, 20 roots, SFF* with
    rewire) at a 150 k-node budget: the GPU runs reproduce the committed oracle summaries
    (tests/golden/config_runs.json, tests/golden/make_config_runs.py)."""
    import json
    import os
    sys_path = os.path.join(golden_dir, "config_runs.json")
    if not os.path.exists(sys_path):
        pytest.skip("tests/golden/config_runs.json not generated")
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_config_runs", os.path.join(golden_dir, "make_config_runs.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    gold = json.load(open(sys_path))
    for key, (name, nroots, opt, budget, wave, waves) in mk.CONFIGS.items():
        sc, w = load_world(ctx, name)
        roots = common.free_roots(lambda p: int(ctx.collide_poses(p[None, :])[0]), sc["limits"], nroots, seed=1,
                                  dim=sc["dim"])
        f = S.Forest(ctx, roots, sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=sc["dim"],
                     optimize=opt, max_iterations=2**31 - 1, node_budget=budget, wave=wave, seed=1)
        f.run(waves)
        got = mk.summary(f)
        for k in got:
            assert got[k] == gold[key][k], (key, k, got[k], gold[key][k])
        f.close()


def test_libm_sampling_mode_equals_the_libm_oracle(S, ctx, golden_dir):
    """Parity mode sffgpu_forest_cfg::libm_sampling: the samples are computed on the host with glibc's cos / sin /
    acos - the arithmetic tests/golden/ref_primitives.json pins to the reference's own randGen.h - and the GPU forest
    must reproduce the oracle's TRIG_LIBM runs (tests/golden/libm_runs.json): BASELINE configs[0] at wave 1 (the
    reference's sequential loop), configs[1] in full, a 100 k-node slice of configs[2]."""
    import importlib.util
    import json
    import os
    path = os.path.join(golden_dir, "libm_runs.json")
    if not os.path.exists(path):
        pytest.skip("tests/golden/libm_runs.json not generated")
    spec = importlib.util.spec_from_file_location("make_config_runs", os.path.join(golden_dir, "make_config_runs.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    RUNS = {   # (same table as tests/golden/make_libm_runs.py)
        "configs[0] dense2d 3 xml points wave 1": ("dense2d", 3, 10000, 200000, 1),
        "configs[1] triang 5 xml points 100k wave 64": ("triang", 5, 100000, 2**31 - 1, 64),
        "configs[2] dense3d 10 roots 100k slice wave 8192": ("dense3d", 10, 100000, 2**31 - 1, 8192),
    }
    gold = json.load(open(path))["runs"]
    for key, (name, nroots, budget, iters, wave) in RUNS.items():
        sc, w = load_world(ctx, name)
        roots = sc["xml_points"][:nroots] if sc["xml_points"] is not None else \
            common.free_roots(w.collide, sc["limits"], nroots, seed=1, dim=sc["dim"])
        kw = dict(dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=sc["dim"], max_iterations=iters,
                  node_budget=budget, wave=wave, seed=1)
        f = S.Forest(ctx, roots, sc["limits"], libm_sampling=True, **kw)
        # (the device-resident engine at every wave size: the host's libm values travel beside the engine words)
        assert f.device_engine()
        f.run()
        got = mk.summary(f)
        for k in got:
            assert got[k] == gold[key][k], (key, k, got[k], gold[key][k])
        f.close()
    # the mode is not a no-op: the default (portable trig on the GPU) run of configs[0] differs in the position bits
    sc, w = load_world(ctx, "dense2d")
    f = S.Forest(ctx, sc["xml_points"][:3], sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"],
                 dim=2, max_iterations=200000, node_budget=10000, wave=1, seed=1)
    f.run()
    assert "%016x" % f.fingerprint() != gold["configs[0] dense2d 3 xml points wave 1"]["fingerprint"]
    f.close()


def test_more_seeds_equal_the_oracle(S, ctx, golden_dir):
    """More seeds, root counts and wave sizes of mid-size runs (SFF and SFF*, all four maps) against the committed
    oracle summaries tests/golden/soak_runs.json (tests/golden/make_soak_runs.py)."""
    import json
    import os
    path = os.path.join(golden_dir, "soak_runs.json")
    if not os.path.exists(path):
        pytest.skip("tests/golden/soak_runs.json not generated")
    import importlib.util
    import sys
    sys.path.insert(0, golden_dir)
    spec = importlib.util.spec_from_file_location("make_soak_runs", os.path.join(golden_dir, "make_soak_runs.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    gold = json.load(open(path))
    for name, nroots, opt, budget, wave, seed in mk.RUNS:
        key = "%s/%d roots/%s/budget %d/wave %d/seed %d" % (name, nroots, "star" if opt else "plain", budget, wave, seed)
        sc, w = load_world(ctx, name)
        roots = common.free_roots(lambda p: int(ctx.collide_poses(p[None, :])[0]), sc["limits"], nroots, seed=seed,
                                  dim=sc["dim"])
        f = S.Forest(ctx, roots, sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=sc["dim"],
                     optimize=opt, max_iterations=2**31 - 1, node_budget=budget, wave=wave, seed=seed)
        f.run(0)
        got = mk.summary(f)
        for k in got:
            assert got[k] == gold[key][k], (key, k, got[k], gold[key][k])
        f.close()


def test_forest_node_budget_and_seeds(S, ctx):
    for seed in (1, 3):
        fo, fg = run_pair(S, ctx, "dense3d", 512, 10**6, seed=seed, n_roots=10, budget=6000)
        assert fo.stats()["n_nodes"] >= 6000
        assert_same_forest(fo, fg)


def test_device_list_overflow_takes_host_path(S, ctx, monkeypatch):
    """Shrink the device-side hit / neighbour lists so that they overflow constantly: the host
    path (generic radius query + host classification) must give the same forest."""
    monkeypatch.setenv("SFFGPU_TEST_HITCAP", "3")
    monkeypatch.setenv("SFFGPU_TEST_NBCAP", "1")
    fo, fg = run_pair(S, ctx, "dense3d_coarse", 64, 5000, seed=7)
    assert_same_forest(fo, fg)
    assert fg.stats()["slow_path_samples"] > 20
    fo, fg = run_pair(S, ctx, "dense3d", 128, 4000, seed=7, optimize=True)
    assert_same_forest(fo, fg)


def test_edge_work_list_overflow_scans_slot_table(S, ctx, monkeypatch):
    """A work list too small for the round's (edge, chunk) items: the edge kernel falls back to scanning the
    slot table; edges and forests must not change."""
    monkeypatch.setenv("SFFGPU_SEG_LISTCAP", "5")
    sc, w = load_world(ctx, "dense3d")
    n = 400
    a = common.poses_near_surface(sc["env"], n, 8, 3.0 * sc["scale"], 6)
    b = a.copy()
    b[:, :3] += np.random.RandomState(2).normal(0, 1, (n, 3)) * sc["sampling_dist"]
    free, fh, ns = ctx.collide_segments(a, b)
    for i in range(n):
        assert (free[i], fh[i], ns[i]) == w.path_free(a[i], b[i]), i
    fo, fg = run_pair(S, ctx, "dense3d", 128, 3000, seed=3, optimize=True)
    assert_same_forest(fo, fg)


def test_grid_recells_under_overflow_pressure(S, ctx, monkeypatch):
    """One-item buckets and a 64-entry overflow list: the grid has to re-cell itself repeatedly; the
    neighbour sets (hence the forest) must not change."""
    monkeypatch.setenv("SFFGPU_TEST_GRID_BK", "1")
    monkeypatch.setenv("SFFGPU_TEST_GRID_OVF", "64")
    fo, fg = run_pair(S, ctx, "triang", 128, 12000, seed=12)
    assert_same_forest(fo, fg)
    assert fg.stats()["grid_rebuilds"] >= 1


def test_grid_list_growth_settles_when_cells_and_buckets_are_exhausted(S, ctx, monkeypatch):
    """Many nodes per xyz cell (a 6-DoF forest whose step is small against the angular range, in a box of a few cells)
    with buckets that may not deepen: cells shrink once, then only the overflow list can grow.  The re-cell trigger has
    to scale with the list from there on (a fixed trigger re-inserted every node and quadrupled the list after every
    wave); the neighbour sets - hence the forest - must not change."""
    monkeypatch.setenv("SFFGPU_TEST_GRID_BK", "1")
    monkeypatch.setenv("SFFGPU_TEST_GRID_BKMAX", "1")
    monkeypatch.setenv("SFFGPU_TEST_GRID_OVF", "64")
    sc, w = load_world(ctx, "triang")
    lim = [-100, -86, -100, -86, 60, 74]
    roots = common.free_roots(w.collide, lim, 2, seed=5)
    kw = dict(dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=6, max_iterations=9000, wave=128, seed=5)
    fo = O.Forest(w, roots, lim, **kw)
    fo.run()
    fg = S.Forest(ctx, roots, lim, **kw)
    fg.run()
    assert_same_forest(fo, fg)
    st = fg.stats()
    assert st["n_nodes"] > 80, st["n_nodes"]
    assert 2 <= st["grid_rebuilds"] <= 12, st["grid_rebuilds"]   # (cells, buckets, then a few list growths - not one per wave)


def test_forest_errors(S, ctx):
    sc, w = load_world(ctx, "dense3d")
    roots = common.free_roots(w.collide, sc["limits"], 3)
    with pytest.raises(S.SffGpuError):
        S.Forest(ctx, roots, sc["limits"], 18.0, 14.0, dim=3)
    with pytest.raises(S.SffGpuError):
        S.Forest(ctx, roots, sc["limits"], 18.0, 14.0, rank=2, world=2)


@pytest.mark.parametrize("world,wave,name,optimize", [(2, 64, "dense3d", False), (3, 256, "dense3d", False),
                                                      (4, 128, "triang", False), (2, 128, "triang", True)])
def test_sharded_rounds_equal_single_gpu(S, name, world, wave, optimize):
    """Multi-GPU wave protocol on ONE device: `world` contexts play the ranks, the record streams
    are exchanged in-process.  Every rank must end in the state of the world=1 run == the oracle."""
    sc = common.scenario(name)
    w = O.World(sc["env"], sc["robot"], O.TRIG_PORTABLE)
    roots = sc["xml_points"][:5] if sc["xml_points"] is not None else common.free_roots(w.collide, sc["limits"], 5, seed=4)
    kw = dict(dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=sc["dim"], max_iterations=6000,
              wave=wave, seed=4, optimize=optimize)
    fo = O.Forest(w, roots, sc["limits"], **kw)
    fo.run()
    ctxs, forests = [], []
    for r in range(world):
        c = S.Context(0)
        c.upload_env(sc["env"])
        c.upload_robot(sc["robot"])
        ctxs.append(c)
        forests.append(S.Forest(c, roots, sc["limits"], rank=r, world=world, **kw))
    rounds = 0
    while True:
        outs = [f.round_begin() for f in forests]
        dones = [d for _, d in outs]
        assert len(set(dones)) == 1
        if dones[0]:
            break
        allw = np.concatenate([r for r, _ in outs]).astype(np.int32)
        counts = np.array([len(r) for r, _ in outs], np.int32)
        for f in forests:
            f.round_commit(allw, counts)
        rounds += 1
        assert rounds < 100000
    for f in forests:
        assert_same_forest(fo, f)
    # the work really was sharded: each rank executed roughly 1/world of the edge checks
    ex = [f.stats()["segments_executed"] for f in forests]
    assert max(ex) < 1.0 * sum(ex) / world * 1.5 + 50
    for f in forests:
        f.close()
    for c in ctxs:
        c.close()


@pytest.mark.parametrize("name,optimize,iters,edges,expect_solved", [
    ("dense2d", False, 3000, ((0, 1), (2, 1)), 2), ("dense2d", True, 3000, ((0, 1), (2, 1)), 2),
    ("dense3d_coarse", True, 8000, ((2, 1), (0, 1)), 1),      # the second start is walled in: DBL_MAX edge (src/lazy.h:280)
    ("triang", False, 1500, ((0, 1), (2, 1)), 0),             # 6-D goal never reached within the budget
])
def test_lazy_edge_rrt_identical(S, ctx, name, optimize, iters, edges, expect_solved):
    """LazyTSP::runRRT (src/lazy.h:160-284): one tree from one root towards a goal, solved when a node comes within
    treeDistance of it.  Two consecutive edges drawn from ONE engine stream (rng_skip), every wave size: nodes,
    parents, costs, the edge's distance and plan, the counters and the stream position equal the oracle's."""
    sc, w = load_world(ctx, name)
    pts = sc["xml_points"] if sc["xml_points"] is not None else common.free_roots(w.collide, sc["limits"], 6, seed=3,
                                                                                 dim=sc["dim"])
    kw = dict(dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=sc["dim"], optimize=optimize,
              max_iterations=iters, seed=11, lazy_edge=True)
    skip_o = 0
    n_solved = 0
    for a, b in edges:
        ro = O.Rrt(w, pts[a:a + 1], sc["limits"], goal=pts[b], rng_skip=skip_o, **kw)
        ro.run()
        so, no, po = ro.stats(), ro.nodes(), ro.lazy_plan()
        assert so["iterations"] > 20
        n_solved += int(so["solved"])
        assert (len(po) > 0) == bool(so["solved"])
        assert (so["lazy_distance"] < 1e300) == bool(so["solved"])
        # (wave = 1 - one iteration per GPU round trip - only where the budget is small: 8 000 RRT* iterations take half a minute)
        for wave in ((1, 0, 64) if iters <= 3000 else (0, 64, 1000)):
            rg = S.Rrt(ctx, pts[a:a + 1], sc["limits"], goal=pts[b], rng_skip=skip_o, wave=wave, **kw)
            rg.run()
            sg = rg.stats()
            for k in so:
                assert so[k] == sg[k], (a, b, wave, k, so[k], sg[k])
            ng = rg.nodes()
            for k in no:
                assert np.array_equal(no[k], ng[k]), (a, b, wave, k)
            assert np.array_equal(po, rg.lazy_plan())
            rg.close()
        skip_o = so["rng_draws"]          # the next edge continues the stream (one RandGen per solver, src/lazy.h:181)
    assert n_solved == expect_solved


@pytest.mark.parametrize("name,optimize,n_roots,goal,bias,iters", [
    ("triang", False, 4, False, 0.0, 1500),        # Multi-T-RRT: trees merge until one is left
    ("triang", True, 1, True, 0.1, 600),           # RRT* towards a goal with goal bias
    ("dense2d", False, 3, False, 0.0, 800),        # 2-D
    ("building", True, 1, False, 0.0, 500),        # RRT* rewiring on the big map
    ("dense3d_coarse", False, 5, False, 0.0, 700),
])
def test_rrt_identical(S, ctx, name, optimize, n_roots, goal, bias, iters):
    """RapidExpTree (src/rrt.h): nodes, parents, merged tree membership, links and the
    reference-equivalent counters equal the oracle's."""
    sc, w = load_world(ctx, name)
    pts = sc["xml_points"] if sc["xml_points"] is not None else common.free_roots(w.collide, sc["limits"], 6, seed=3,
                                                                                 dim=sc["dim"])
    roots = pts[:n_roots]
    g = pts[4] if goal else None
    kw = dict(dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=sc["dim"], optimize=optimize, goal=g,
              priority_bias=bias, max_iterations=iters, seed=3)
    ro = O.Rrt(w, roots, sc["limits"], **kw)
    ro.run()
    so = ro.stats()
    no, lo = ro.nodes(), ro.links()
    assert so["n_nodes"] > 30
    # wave = 1: one iteration per GPU round trip; 0: adaptive speculative waves; 64 / 1000: fixed wave sizes.
    # Every wave size must commit exactly the reference's sequence (conflicts cut the wave, the RNG rewinds).
    for wave in (1, 0, 64, 1000):
        rg = S.Rrt(ctx, roots, sc["limits"], wave=wave, **kw)
        rg.run()
        sg = rg.stats()
        for k in so:
            assert so[k] == sg[k], (wave, k, so[k], sg[k])
        ng = rg.nodes()
        for k in no:
            assert np.array_equal(no[k], ng[k]), (wave, k)
        lg = rg.links()
        for k in lo:
            assert np.array_equal(lo[k], lg[k]), (wave, k)
        nt = n_roots + (1 if goal else 0)
        do, co = ro.paths(nt)
        dg, cg = rg.paths(nt)
        assert co == len(cg) and np.array_equal(do, dg)
        for a in range(nt):
            for b in range(a + 1, nt):
                assert np.array_equal(ro.plan(a, b), rg.plan(a, b))
        if wave == 0:
            # RapidExpTree::smoothPaths (src/rrt.h:354-379): the link plans shrink identically, and - as in the
            # reference, whose matrix holds earlier copies - the cost matrix and matrix plans stay what they were
            po = ro.smooth() if not hasattr(ro, "_smoothed") else ro._smoothed
            ro._smoothed = po
            pg = rg.smooth()
            assert len(po) == len(pg)
            for x, y in zip(po, pg):
                assert np.array_equal(x, y)
            dg2, _ = rg.paths(nt)
            assert np.array_equal(dg, dg2)
        if wave != 1:
            assert sg["waves"] < so["iterations"] or so["iterations"] < 8
        rg.close()
