"""The device-resident SFF engine (csrc/devforest.hip + forest_dev.cpp: in-order commit, frontier picks and wave
bookkeeping on the GPU, one host sync per wave) against the CPU oracle at the same wave size - and against the
host-replay engine it replaces.  Also its fault paths: bounded device lists that overflow (the round is redone on
the host path), arrays and the border table that have to grow, staged runs with getters in between."""
import os

import numpy as np
import pytest

import common
import oracle_lib as O
from test_gpu_parity import assert_same_forest, load_world

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


class engine:
    """SFFGPU_ENGINE / test knobs are read when a forest is created"""

    def __init__(self, **env):
        self.env = {k: str(v) for k, v in env.items()}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.env}
        os.environ.update(self.env)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def make(S, ctx, name, wave, iters, seed, n_roots=5, budget=0, which="device", optimize=False, goal_offset=None, **env):
    sc, w = load_world(ctx, name)
    roots = sc["xml_points"][:n_roots] if sc["xml_points"] is not None else \
        common.free_roots(w.collide, sc["limits"], n_roots, seed=seed, dim=sc["dim"])
    kw = dict(dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=sc["dim"], max_iterations=iters,
              node_budget=budget, wave=wave, seed=seed, optimize=optimize)
    if goal_offset is not None:
        g = roots[0].copy()
        g[:3] += np.array(goal_offset, dtype=np.float64)
        kw["goal"] = g
    fo = O.Forest(w, roots, sc["limits"], **kw)
    with engine(SFFGPU_ENGINE=which, **env):
        fg = S.Forest(ctx, roots, sc["limits"], **kw)
    return fo, fg


@pytest.mark.parametrize("name,wave,iters", [
    ("dense3d", 1, 600), ("dense3d", 7, 2500), ("dense3d", 64, 8000), ("dense3d", 512, 40000), ("dense3d", 4096, 60000),
    ("dense3d_coarse", 1, 800), ("dense3d_coarse", 64, 8000), ("dense3d_coarse", 1024, 20000),
    ("triang", 1, 600), ("triang", 128, 10000), ("triang", 2048, 60000),
    ("dense2d", 3, 1500), ("dense2d", 256, 8000), ("building", 256, 8000),
])
def test_device_engine_equals_the_oracle(S, ctx, name, wave, iters):
    fo, fg = make(S, ctx, name, wave, iters, seed=3)
    fo.run()
    fg.run()
    assert fo.stats()["n_nodes"] > 40
    assert_same_forest(fo, fg)


@pytest.mark.parametrize("name,wave,n_roots,optimize,offset", [
    ("triang", 64, 2, False, [12, 8, 5]), ("triang", 512, 3, False, [30, 20, 10]), ("triang", 256, 1, True, [12, 8, 5]),
    ("building", 128, 1, False, [12, 8, 5]), ("dense3d", 1024, 4, False, [30, 25, 8]),
])
def test_single_goal_mode_on_the_device_engine(S, ctx, name, wave, n_roots, optimize, offset):
    """Problem::hasGoal (src/forest.h:91-109, :283-299, :369-372) on the device engine: a neighbour of another tree
    rejects the sample without an edge check unless it is the goal; the round in which a sample reaches the goal is
    rolled back on the device and replayed by the host engine, which stops in the middle of it like the reference."""
    fo, fg = make(S, ctx, name, wave, 60000, seed=8, n_roots=n_roots, optimize=optimize, goal_offset=offset)
    fo.run()
    fg.run()
    assert_same_forest(fo, fg)
    st = fg.stats()
    assert fg.device_engine() and st["solved"] == 1 and st["n_borders"] >= 1
    assert st["waves"] > st["host_fallback_waves"] >= 1   # (only the solving wave went to the host)


@pytest.mark.parametrize("name,wave,iters", [
    ("dense3d", 1, 500), ("dense3d", 7, 2500), ("dense3d", 64, 8000), ("dense3d", 512, 40000), ("dense3d", 4096, 60000),
    ("dense3d_coarse", 64, 8000), ("dense3d_coarse", 1024, 20000),
    ("triang", 1, 600), ("triang", 128, 10000), ("triang", 2048, 60000),
    ("dense2d", 3, 1500), ("dense2d", 256, 8000), ("building", 256, 8000), ("building", 2048, 80000),
])
def test_sff_star_on_the_device_engine_equals_the_oracle(S, ctx, name, wave, iters):
    """SFF* (optimize = true) with the rounds committed on the GPU: the k-nearest sets, choose-parent and the rewires
    (src/forest.h:307-351) of the samples a round accepts are the fixed point of "sample i sees the rewires of every
    accepted sample before it" (csrc/devstar.hip) - parents, costs, borders and the reference-equivalent call counters
    must equal the oracle's sequential loop."""
    fo, fg = make(S, ctx, name, wave, iters, seed=5, optimize=True)
    assert fg.device_engine()
    fo.run()
    fg.run()
    assert fo.stats()["n_nodes"] > 40
    assert_same_forest(fo, fg)
    sg = fg.stats()
    assert sg["star_rounds"] > 0 and sg["star_passes"] >= sg["star_rounds"] and sg["star_members"] > 0
    assert sg["host_fallback_waves"] == 0
    no = fo.nodes()
    if len(no["parent"]) > 300:
        # rewiring really happened: some node's parent is younger than the node itself
        assert np.any(no["parent"] > np.arange(len(no["parent"]))) and sg["star_rewires"] > 0


def test_sff_star_device_engine_staged_runs_and_the_host_engine(S, ctx):
    """SFF* on the device engine in stages with getters in between (the host mirror has to pick up rewired parents and
    costs of nodes it already holds), against the oracle and against the host-replay engine."""
    fo, fg = make(S, ctx, "dense3d", 512, 10 ** 7, seed=8, budget=15000, optimize=True)
    fo.run()
    fps = []
    while True:
        w0 = fg.stats()["waves"]
        fg.run(4)
        if fg.stats()["waves"] == w0:
            break
        fps.append(fg.fingerprint())
        assert len(fg.nodes()["parent"]) == fg.stats()["n_nodes"]
    assert len(fps) > 3
    assert_same_forest(fo, fg)
    do = fo.paths()
    dg, _ = fg.paths()
    assert np.array_equal(do, dg)
    fp, sd = fg.fingerprint(), fg.stats()
    fg.close()
    _, fh = make(S, ctx, "dense3d", 512, 10 ** 7, seed=8, budget=15000, optimize=True, which="host")
    assert not fh.device_engine()
    fh.run()
    assert fh.fingerprint() == fp
    sh = fh.stats()
    for k in ("iterations", "n_nodes", "n_borders", "collide_calls", "path_free_calls", "nn_queries", "waves"):
        assert sd[k] == sh[k], k


@pytest.mark.parametrize("env", [dict(SFFGPU_TEST_HITCAP=3), dict(SFFGPU_TEST_NBCAP=1), dict(SFFGPU_TEST_STAR_PASSES=1)])
def test_sff_star_device_faults_finish_the_wave_on_the_host_path(S, ctx, env):
    """a bounded device list that overflows - or a rewire fixed point that does not settle within the launches of a round
    (forced: one pass only) - hands the wave to the host-replay engine; the forest must not change"""
    fo, fg = make(S, ctx, "dense3d_coarse", 256, 12000, seed=4, optimize=True, **env)
    fo.run()
    fg.run()
    assert fg.stats()["host_fallback_waves"] > 0
    assert_same_forest(fo, fg)


@pytest.mark.parametrize("name,wave,iters,env,faults", [
    ("dense3d", 2048, 80000, dict(SFFGPU_STAR_TAIL=0), False),          # one launch per pass (the fixed chain)
    ("dense3d", 2048, 80000, dict(SFFGPU_STAR_TAIL_WGS=3), False),      # three workgroups loop over all accepted samples / items
    ("building", 1024, 40000, dict(SFFGPU_STAR_TAIL_WGS=16), False),
    ("building", 4096, 120000, dict(SFFGPU_STAR_TAIL_WGS=2), False),    # 8 wavefronts < the 64 ticket groups of the exact items: one ticket word
    ("building", 1024, 40000, dict(SFFGPU_TEST_STAR_PASSES=3), True),   # the tail gives up after its second pass: host path
    ("building", 1024, 40000, dict(SFFGPU_TEST_STAR_STALL=7), True),    # every 7th round a workgroup stays away: the barrier times out
])
def test_sff_star_passes_as_one_launch_and_as_a_chain_build_the_same_forest(S, ctx, name, wave, iters, env, faults):
    """the rewire fixed point's passes after the first run as ONE launch (k_star_tail: the first workgroups alternate pass
    and exact phases with a barrier between them, the words they exchange written through / read from memory) - the
    forest must be the oracle's whatever the number of workgroups, equal to what one launch per pass builds, and a round
    that does not settle within the allowed passes must go to the host path"""
    fo, fg = make(S, ctx, name, wave, iters, seed=6, optimize=True, **env)
    fo.run()
    fg.run()
    assert fo.stats()["n_nodes"] > 1000
    assert_same_forest(fo, fg)
    sg = fg.stats()
    assert sg["star_rounds"] > 0 and sg["star_passes"] > sg["star_rounds"]      # some round needed more than its first pass
    assert (sg["host_fallback_waves"] > 0) == faults


@pytest.mark.parametrize("which", ["device", "host"])
def test_sff_star_parent_history_gives_the_forest_after_any_iteration(S, ctx, which):
    """record_parents: the (node, parent, iteration) history of node creations and applied rewires - what an exact
    per-iteration tree dump of SFF* needs (saveIterCheck, src/forest.h:570-578).  The forest "after iteration k" rebuilt
    from it must equal the oracle's run that stops at iteration k (same waves, same seed: a prefix of the longer run)."""
    sc, w = load_world(ctx, "triang")
    roots = sc["xml_points"][:4]
    kw = dict(dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=6, wave=96, seed=9, optimize=True)
    with engine(SFFGPU_ENGINE=which):
        fg = S.Forest(ctx, roots, sc["limits"], max_iterations=9000, record_parents=True, **kw)
    assert fg.device_engine() == (which == "device")
    fg.run()
    h = fg.parent_history()
    ng = fg.nodes()
    assert np.all(np.diff(h["iter"]) >= 0) and len(h["node"]) > len(ng["parent"])     # creations + rewires
    # the last entry of every node is its final parent
    last = np.full(len(ng["parent"]), -2, np.int32)
    last[h["node"]] = h["parent"]
    assert np.array_equal(last, ng["parent"])
    for k in (137, 1000, 4321, 9000):
        fo = O.Forest(w, roots, sc["limits"], max_iterations=k, **kw)
        fo.run()
        no = fo.nodes()
        n_k = int(np.sum(ng["iter"] <= k))
        assert n_k == len(no["parent"])
        par = np.full(n_k, -2, np.int32)
        m = h["iter"] <= k
        par[h["node"][m]] = h["parent"][m]          # (entries are sorted by iteration: the last one wins)
        assert np.array_equal(par, no["parent"]), k
    fg.close()


def test_device_engine_is_the_default_for_gpu_sized_waves_and_equals_the_host_engine(S, ctx):
    # (a forest owns its context's node store while it lives: one at a time)
    fo, fd = make(S, ctx, "dense3d", 1024, 60000, seed=11, which="")       # default choice
    fd.run()
    fo.run()
    assert_same_forest(fo, fd)
    fp_d, sd = fd.fingerprint(), fd.stats()
    fd.close()
    _, fh = make(S, ctx, "dense3d", 1024, 60000, seed=11, which="host")
    fh.run()
    assert fp_d == fh.fingerprint()
    sh = fh.stats()
    for k in ("iterations", "n_nodes", "n_borders", "collide_calls", "path_free_calls", "nn_queries", "waves", "sweeps",
              "sweep_queries", "sweep_nodes", "poses_executed", "segments_executed", "samples_executed"):
        assert sd[k] == sh[k], k
    # the device engine really ran: far less host time per wave than the replaying engine
    assert sd["host_ms"] < 0.6 * sh["host_ms"]


def test_spatial_order_of_the_slots_changes_nothing(S, ctx):
    """round 5: the query kernel takes a round's samples from per-sub-range lists of the wave's spatial order
    (sffk::OrderView) instead of by sample index; SFFGPU_NO_ORDER=1 is the walk by index.  Same forest either way,
    plain SFF and SFF*, also when a wave holds fewer slots than the launch is sized for."""
    for name, wave, iters, optimize in (("dense3d", 2048, 20000, False), ("dense3d", 1000, 12000, True),
                                        ("building", 4096, 20000, False), ("dense3d", 8192, 50000, False)):
        # (the order is on by default from waves of 4 096 slots; SFFGPU_ORDER_MIN_WAVE lowers that for the smaller cases)
        fo, fg = make(S, ctx, name, wave, iters, seed=5, optimize=optimize, SFFGPU_ORDER_MIN_WAVE=2)
        fo.run()
        fg.run()
        assert fg.device_engine()
        assert_same_forest(fo, fg)
        fp = fg.fingerprint()
        fg.close()
        _, fn = make(S, ctx, name, wave, iters, seed=5, optimize=optimize, SFFGPU_NO_ORDER=1)
        fn.run()
        assert fn.fingerprint() == fp
        fn.close()


def test_many_candidate_items_shared_by_the_workgroup_change_nothing(S, ctx):
    """round 5: an exact-kernel item (edge chunk or pose) with many candidate triangles is worked off in blocks by the
    idle wavefronts of its workgroup (share_help in csrc/kernels.hip; environments above 4 096 triangles by default).
    SFFGPU_SHARE=0 / 1 forces the choice at every launch: same forest - and the oracle's - either way, plain SFF and SFF*."""
    for name, wave, iters, optimize in (("building", 2048, 30000, False), ("building", 1024, 12000, True),
                                        ("dense3d", 1024, 15000, False)):
        fps = []
        for share in ("0", "1"):
            with engine(SFFGPU_SHARE=share):
                fo, fg = make(S, ctx, name, wave, iters, seed=9, optimize=optimize)
                fg.run()
                if share == "0":
                    fo.run()
                    assert_same_forest(fo, fg)
                fps.append((fg.fingerprint(), fg.stats()["collide_calls"], fg.stats()["path_free_calls"]))
                fg.close()
        assert fps[0] == fps[1]


def test_saturating_forest_terminates_solved_with_closed_list_picks(S, ctx):
    """coarse steps: the frontier runs empty, the engine keeps expanding from the closed list (src/forest.h:
    136-141) until every tree is connected -> solved"""
    fo, fg = make(S, ctx, "dense3d_coarse", 512, 10 ** 7, seed=2)
    fo.run()
    fg.run()
    so = fo.stats()
    assert so["solved"] == 1 and so["frontier_size"] == 0 and so["closed_size"] > 100
    assert_same_forest(fo, fg)


def test_waves_of_one_slot_run_as_one_persistent_wavefront(S, ctx):
    """wave = 1 - the reference's own loop order - on ONE wavefront (SFFGPU_SPEC=0: k_seq_waves runs whole outer iterations back
    to back inside one launch; the speculative kernel of round 6 is tested below and falls back to this one).  Staged runs with
    getters in between, the node budget, a forced hit-list overflow (that wave is finished on the host engine) - against the
    oracle's sequential loop - and the round engine at wave 1 gives the same forest.  (The saturating forest - closed-list
    picks, frontier erases, termination by maxConnected - runs through this kernel in
    test_waves_of_one_slot_speculated_over_many_wavefronts' second run.)"""
    fo, fg = make(S, ctx, "dense3d_coarse", 1, 6000, seed=2, SFFGPU_SPEC=0)
    fo.run()
    fg.run()
    assert_same_forest(fo, fg)
    assert fg.stats()["sweeps"] == fo.stats()["iterations"] and fg.stats()["spec_steps"] == 0     # (one round per iteration)
    fp = fg.fingerprint()
    fg.close()
    with engine(SFFGPU_NO_SEQ=1):                            # the round engine (33 launches per wave) on the same job
        _, fr = make(S, ctx, "dense3d_coarse", 1, 6000, seed=2)
        fr.run()
    assert fr.fingerprint() == fp
    fr.close()
    # staged, with a node budget
    fo, fg = make(S, ctx, "triang", 1, 10 ** 7, seed=5, budget=3000, SFFGPU_SPEC=0)
    fo.run()
    while True:
        w0 = fg.stats()["waves"]
        fg.run(700)
        if fg.stats()["waves"] == w0:
            break
        assert len(fg.nodes()["parent"]) == fg.stats()["n_nodes"] and len(fg.frontier()) == fg.stats()["frontier_size"]
    assert_same_forest(fo, fg)
    fg.close()
    # a hit list of three entries: overflows hand single waves to the host engine
    fo, fg = make(S, ctx, "dense3d_coarse", 1, 6000, seed=4, SFFGPU_TEST_HITCAP=3, SFFGPU_SPEC=0)
    fo.run()
    fg.run()
    assert fg.stats()["host_fallback_waves"] > 0
    assert_same_forest(fo, fg)
    fg.close()


@pytest.mark.parametrize("name,iters,seed,optimize,env", [
    ("dense3d", 6000, 2, False, dict()),                                   # the default tree: 3 waves deep, one set of workers
    ("dense3d", 6000, 2, True, dict()),                                    # SFF*: the chain of all-fail scenarios
    ("dense3d", 3000, 7, False, dict(SFFGPU_SPEC_DEPTH=1)),
    ("dense3d", 3000, 7, False, dict(SFFGPU_SPEC_DEPTH=2, SFFGPU_SPEC_SETS=2)),
    ("dense3d", 3000, 7, False, dict(SFFGPU_SPEC_DEPTH=4, SFFGPU_SPEC_SETS=1)),
    ("dense3d", 3000, 7, True, dict(SFFGPU_SPEC_DEPTH=4, SFFGPU_SPEC_SETS=3)),
    ("triang", 4000, 5, False, dict()),
    ("triang", 4000, 5, True, dict()),
    ("dense2d", 3000, 4, False, dict()),                                   # one engine word per attempt
    ("dense3d_coarse", 10 ** 7, 2, False, dict()),                         # runs into saturation: closed-list picks, maxConnected
])
def test_waves_of_one_slot_speculated_over_many_wavefronts(S, ctx, name, iters, seed, optimize, env):
    """wave = 1 through k_spec_waves (round 6): a step evaluates the tree of scenarios (which attempt of each of the next
    waves is accepted, if any) side by side, one workgroup per (scenario, attempt); the leader commits the path that really
    happened in the reference's order.  The forest, the reference-equivalent counters and the border list equal the oracle's
    sequential loop (src/forest.h:122-202) and the single-wavefront kernel's; every iteration went through the kernel."""
    fo, fg = make(S, ctx, name, 1, iters, seed=seed, optimize=optimize, **env)
    fo.run()
    fg.run()
    so, sg = fo.stats(), fg.stats()
    assert so["n_nodes"] > 60
    assert_same_forest(fo, fg)
    assert sg["spec_committed"] == so["iterations"] and sg["host_fallback_waves"] == 0
    assert sg["spec_evaluated"] >= sg["spec_committed"] and sg["spec_steps"] > 0
    print("\n[k_spec_waves %s%s %s] %d iterations in %d steps (%.2f per step), %d attempts evaluated: speculation ratio %.2f"
          % (name, " SFF*" if optimize else "", env, so["iterations"], sg["spec_steps"], so["iterations"] / sg["spec_steps"],
             sg["spec_evaluated"], sg["spec_evaluated"] / sg["spec_committed"]))
    fp = fg.fingerprint()
    fg.close()
    _, f1 = make(S, ctx, name, 1, iters, seed=seed, optimize=optimize, SFFGPU_SPEC=0)     # one wavefront (k_seq_waves)
    f1.run()
    assert f1.fingerprint() == fp and f1.stats()["spec_steps"] == 0
    f1.close()


def test_speculated_waves_staged_budget_faults_and_parity_mode(S, ctx):
    """k_spec_waves where the launch has to stop in the middle of a step: staged runs (run(n) ends after exactly n waves), the
    node budget, a hit list of three entries (the attempt that overflows is rolled back and its wave finished on the host
    engine), arrays that grow, the libm parity mode (the host's trig table), and a worker that never answers (tests only: the
    leader times out, the wave is finished on the host path and the single-wavefront kernel takes over)."""
    fo, fg = make(S, ctx, "triang", 1, 10 ** 7, seed=5, budget=3000)
    fo.run()
    while True:
        w0 = fg.stats()["waves"]
        fg.run(333)
        w1 = fg.stats()["waves"]
        if w1 == w0:
            break
        assert w1 - w0 == 333 or fg.stats()["n_nodes"] >= 3000
        assert len(fg.nodes()["parent"]) == fg.stats()["n_nodes"] and len(fg.frontier()) == fg.stats()["frontier_size"]
    assert_same_forest(fo, fg)
    assert fg.stats()["spec_steps"] > 0
    fg.close()
    fo, fg = make(S, ctx, "dense3d_coarse", 1, 6000, seed=4, SFFGPU_TEST_HITCAP=3)
    fo.run()
    fg.run()
    assert fg.stats()["host_fallback_waves"] > 0 and fg.stats()["spec_steps"] > 0
    assert_same_forest(fo, fg)
    fg.close()
    for opt in (False, True):
        sc, w = load_world(ctx, "dense3d")
        roots = common.free_roots(w.collide, sc["limits"], 5, seed=9, dim=6)
        kw = dict(dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=6, max_iterations=2500, wave=1, seed=9, optimize=opt)
        wl = O.World(sc["env"], sc["robot"], O.TRIG_LIBM)
        fo = O.Forest(wl, roots, sc["limits"], **kw)
        fg = S.Forest(ctx, roots, sc["limits"], libm_sampling=True, **kw)
        fo.run()
        fg.run()
        assert_same_forest(fo, fg)
        assert fg.stats()["spec_committed"] == fo.stats()["iterations"]
        fg.close()
    mid_wave = 0
    for knob in (8 * 40 + 0, 8 * 40 + 1, 8 * 41 + 1, 8 * 42 + 1, 8 * 43 + 2, 8 * 44 + 1):
        # slot 0 = the step's first attempt (always waited for: the launch ends in front of the wave); the others are only
        # waited for when the attempts before them were rejected (then the wave is finished on the host engine)
        fo, fg = make(S, ctx, "dense3d", 1, 3000, seed=6, SFFGPU_TEST_SPEC_STALL=knob)
        fo.run()
        fg.run()
        sg = fg.stats()
        assert_same_forest(fo, fg)
        if knob & 7 == 0:
            assert 0 < sg["spec_steps"] <= 41 and sg["spec_committed"] < fo.stats()["iterations"]      # the rest: k_seq_waves
        elif sg["spec_committed"] < fo.stats()["iterations"]:
            mid_wave += 1
            assert sg["host_fallback_waves"] == 1
        fg.close()
    assert mid_wave > 0


def test_speculated_waves_other_attempt_counts_iteration_caps_and_the_parent_history(S, ctx):
    """k_spec_waves off the beaten path: ThresholdMisses other than 5 (the tree has ThresholdMisses + 1 outcomes per wave), an
    iteration cap that ends the run in the middle of a wave (the attempts beyond it are never evaluated: src/forest.h:155-159),
    and SFF*'s parent history (record_parents: creations and rewires in the reference's order, written by the leader)."""
    sc, w = load_world(ctx, "dense3d")
    roots = common.free_roots(w.collide, sc["limits"], 6, seed=11, dim=6)
    base = dict(dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=6, wave=1, seed=11)
    for tm, optimize in ((1, False), (2, True), (3, False), (8, False), (7, True)):
        kw = dict(base, threshold_misses=tm, max_iterations=2200, optimize=optimize)
        fo = O.Forest(w, roots, sc["limits"], **kw)
        fg = S.Forest(ctx, roots, sc["limits"], **kw)
        fo.run()
        fg.run()
        assert_same_forest(fo, fg)
        assert fg.stats()["spec_committed"] == fo.stats()["iterations"], tm
        fg.close()
    for iters in (1501, 1502, 1503, 1504, 1505, 1506):
        for optimize in (False, True):
            kw = dict(base, max_iterations=iters, optimize=optimize)
            fo = O.Forest(w, roots, sc["limits"], **kw)
            fg = S.Forest(ctx, roots, sc["limits"], **kw)
            fo.run()
            fg.run()
            assert fo.stats()["iterations"] == iters
            assert_same_forest(fo, fg)
            fg.close()
    hist = {}
    for spec in ("1", "0"):
        with engine(SFFGPU_SPEC=spec):
            fg = S.Forest(ctx, roots, sc["limits"], max_iterations=5000, optimize=True, record_parents=True, **base)
        fg.run()
        assert (fg.stats()["spec_steps"] > 0) == (spec == "1")
        hist[spec] = fg.parent_history()
        fg.close()
    for k in ("node", "parent", "iter"):
        assert np.array_equal(hist["1"][k], hist["0"][k]), k
    assert len(hist["1"]["node"]) > 1000


def test_staged_runs_with_getters_in_between(S, ctx):
    fo, fg = make(S, ctx, "dense3d", 512, 10 ** 7, seed=5, budget=20000)
    fo.run()
    fps = []
    while True:
        w0 = fg.stats()["waves"]
        fg.run(3)
        if fg.stats()["waves"] == w0:
            break
        fps.append(fg.fingerprint())          # forces a host-mirror refresh in the middle of the run
        assert len(fg.nodes()["parent"]) == fg.stats()["n_nodes"]
        assert len(fg.frontier()) == fg.stats()["frontier_size"]
    assert len(fps) > 3 and len(set(fps)) == len(fps)
    assert_same_forest(fo, fg)
    # path extraction works on the refreshed mirror
    do = fo.paths()
    dg, _ = fg.paths()
    assert np.array_equal(do, dg)


@pytest.mark.parametrize("env", [dict(SFFGPU_TEST_HITCAP=3), dict(SFFGPU_TEST_NBCAP=1)])
def test_list_overflow_faults_finish_the_wave_on_the_host_path(S, ctx, env):
    fo, fg = make(S, ctx, "dense3d_coarse", 256, 12000, seed=4, **env)
    fo.run()
    fg.run()
    assert fg.stats()["slow_path_samples"] > 0      # the fault path ran
    assert_same_forest(fo, fg)


def test_survivor_list_overflow_scans_the_slot_table(S, ctx):
    """the fused cull's survivor list holds 5 items only: the exact kernel has to test every live pose and every chunk
    of every live edge itself; the forest must not change"""
    fo, fg = make(S, ctx, "dense3d", 512, 30000, seed=6, SFFGPU_SEG_LISTCAP=5)
    with engine(SFFGPU_SEG_LISTCAP=5):
        fo.run()
        fg.run()
    assert_same_forest(fo, fg)


def test_library_driven_rccl_exchange_on_one_rank(S):
    """the library's own RCCL communicator (sffgpu_ctx_rccl_init): with the self-exchange knob a one-rank forest packs
    its answer records, all-gathers them with ncclAllGather on the library's stream and unpacks them in every round,
    whole waves enqueued by the library - the forest must not change"""
    import space_filling_forest_star_amd._lib as L
    c2 = S.Context(0)
    try:
        c2.rccl_init(L.rccl_unique_id(), 0, 1)
        fo, fg = make(S, c2, "dense3d", 1024, 60000, seed=8, SFFGPU_TEST_EXCHANGE_SELF=1)
        with engine(SFFGPU_TEST_EXCHANGE_SELF=1):
            fo.run()
            assert fg.run() is None
        assert fo.stats()["n_nodes"] > 2000
        assert_same_forest(fo, fg)
        fg.close()
    finally:
        c2.close()


def test_arrays_and_border_table_grow_on_demand(S, ctx):
    """no node budget: the store starts at 4096 nodes and has to grow; many borders: the border list and its hash
    table start small (test knob) and have to grow too"""
    fo, fg = make(S, ctx, "dense3d_coarse", 256, 26000, seed=6, SFFGPU_TEST_BORDER_CAP=64)
    fo.run()
    fg.run()
    assert fo.stats()["n_nodes"] > 3000 and fo.stats()["n_borders"] > 300
    assert_same_forest(fo, fg)


def test_iteration_cap_inside_a_round(S, ctx):
    for iters in (1000, 1003, 2500):
        fo, fg = make(S, ctx, "triang", 300, iters, seed=8)
        fo.run()
        fg.run()
        assert fo.stats()["iterations"] == iters
        assert_same_forest(fo, fg)
