"""Edge cases of the C ABI on the GPU: empty and ragged inputs, argument errors, missing map, big batches,
deep box hierarchies, tiny robots."""
import ctypes as C

import numpy as np
import pytest

import common
import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    import space_filling_forest_star_amd as S
    return S


def test_calls_before_upload_and_bad_arguments(S):
    ctx = S.Context(0)
    L = S.lib()
    one = np.zeros((1, 6))
    with pytest.raises(S.SffGpuError):
        ctx.collide_poses(one)                      # no meshes yet
    with pytest.raises(S.SffGpuError):
        ctx.collide_segments(one, one)
    assert L.sffgpu_collide_poses(ctx.h, None, 3, None) == -1          # SFFGPU_ERR_ARG
    assert L.sffgpu_mesh_upload(ctx.h, 7, one.ctypes.data_as(C.POINTER(C.c_double)), 1) < 0
    assert b"role" in L.sffgpu_last_error(ctx.h)
    with pytest.raises(S.SffGpuError):
        ctx.upload_robot(np.zeros((0, 9)))          # a robot needs triangles
    h = C.c_void_p()
    assert L.sffgpu_create(99, C.byref(h)) < 0 and b"range" in L.sffgpu_last_error(None)
    ctx.close()


def test_empty_batches_and_missing_map(S):
    sc = common.scenario("triang")
    ctx = S.Context(0)
    ctx.upload_robot(sc["robot"])
    ctx.upload_env(np.zeros((0, 9)))                # Environment::HasMap == false (src/environment.h:307-309)
    poses = common.random_poses(sc["limits"], 100, 1)
    assert not ctx.collide_poses(poses).any()
    a, b = poses[:50], poses[50:]
    free, fh, ns = ctx.collide_segments(a, b)
    assert free.all() and (fh == -1).all()
    L = O.lib()
    for i in range(50):                              # the sample counts still follow dist6 / 0.1
        parts = L.sffo_distance(O.dp(a[i]), O.dp(b[i])) / 0.1
        assert ns[i] == (int(np.ceil(parts)) - 1 if parts > 1 else 0)
    assert len(ctx.collide_poses(np.zeros((0, 6)))) == 0
    f, h, n = ctx.collide_segments(np.zeros((0, 6)), np.zeros((0, 6)))
    assert len(f) == 0
    ctx.nodes_reset(0)
    idx, dist, cnt = ctx.radius(np.zeros((3, 6)), 5.0)   # empty store
    assert (cnt == 0).all()
    idx, dist, cnt = ctx.knn(np.zeros((3, 6)), 4)
    assert (cnt == 0).all()
    # a forest without a map just fills the limits
    f = S.Forest(ctx, np.zeros((2, 6)) + [[1, 1, 1, 0, 0, 0], [50, 50, 50, 0, 0, 0]], sc["limits"], 5.0, 4.0,
                 max_iterations=800, wave=16, seed=3)
    f.run()
    assert f.stats()["n_nodes"] > 100 and f.stats()["collide_calls"] > 0
    f.close()
    ctx.close()


def test_big_batches_match_oracle_on_samples(S):
    sc = common.scenario("dense3d")
    ctx = S.Context(0)
    ctx.upload_env(sc["env"])
    ctx.upload_robot(sc["robot"])
    w = O.World(sc["env"], sc["robot"], O.TRIG_PORTABLE)
    n = 300000
    poses = np.vstack([common.random_poses(sc["limits"], n // 2, 21), common.poses_near_surface(sc["env"], n // 2, 22, 0.5)])
    hit = ctx.collide_poses(poses)
    pick = np.random.RandomState(1).choice(n, 3000, replace=False)
    assert np.array_equal(hit[pick], w.collide_many(poses[pick]))
    m = 120000
    a = common.poses_near_surface(sc["env"], m, 23, 4.0)
    b = a.copy()
    b[:, :3] += np.random.RandomState(2).normal(0, 8, (m, 3))
    free, fh, ns = ctx.collide_segments(a, b)
    for i in np.random.RandomState(3).choice(m, 600, replace=False):
        assert (free[i], fh[i], ns[i]) == w.path_free(a[i], b[i])
    ctx.close()


@pytest.mark.parametrize("n_side,levels", [(200, 2), (400, 3)])
def test_deep_hierarchies(S, n_side, levels):
    """n_side x n_side x 2 triangles: 80 k -> 2 hierarchy levels, 320 k -> 3 (64-ary)."""
    xs = np.arange(n_side, dtype=np.float64)
    gx, gy = np.meshgrid(xs, xs, indexing="ij")
    z = 3.0 * np.sin(gx * 0.05) * np.cos(gy * 0.07)

    def P(i, j):
        ii = np.clip(i, 0, n_side - 1)
        jj = np.clip(j, 0, n_side - 1)
        return np.stack([gx[ii, jj], gy[ii, jj], z[ii, jj]], -1)
    i, j = np.meshgrid(np.arange(n_side - 1), np.arange(n_side - 1), indexing="ij")
    t1 = np.concatenate([P(i, j), P(i + 1, j), P(i, j + 1)], -1).reshape(-1, 9)
    t2 = np.concatenate([P(i + 1, j), P(i + 1, j + 1), P(i, j + 1)], -1).reshape(-1, 9)
    env = np.ascontiguousarray(np.vstack([t1, t2]))
    assert (len(env) > 64 * 64) and ((len(env) > 64 ** 3) == (levels == 3))
    rob = common.meshes()["robot_cylinder_small"] * 3.0
    ctx = S.Context(0)
    ctx.upload_env(env)
    ctx.upload_robot(rob)
    w = O.World(env, rob, O.TRIG_PORTABLE)
    rs = np.random.RandomState(5)
    poses = np.zeros((4000, 6))
    poses[:, 0] = rs.uniform(0, n_side - 1, 4000)
    poses[:, 1] = rs.uniform(0, n_side - 1, 4000)
    poses[:, 2] = rs.uniform(-5, 5, 4000)
    poses[:, 3:] = rs.uniform(-np.pi, np.pi, (4000, 3))
    got = ctx.collide_poses(poses)
    assert np.array_equal(got, w.collide_many(poses))
    assert 0.1 < got.mean() < 0.9
    a = poses[:1500]
    b = a.copy()
    b[:, :3] += rs.normal(0, 3, (1500, 3))
    free, fh, ns = ctx.collide_segments(a, b)
    for k in range(0, 1500, 5):
        assert (free[k], fh[k], ns[k]) == w.path_free(a[k], b[k])
    ctx.close()


def test_one_triangle_robot_and_wave_larger_than_frontier(S):
    sc = common.scenario("triang")
    rob = sc["robot"][:1]
    ctx = S.Context(0)
    ctx.upload_env(sc["env"])
    ctx.upload_robot(rob)
    w = O.World(sc["env"], rob, O.TRIG_PORTABLE)
    poses = common.poses_near_surface(sc["env"], 3000, 31, 1.0)
    assert np.array_equal(ctx.collide_poses(poses), w.collide_many(poses))
    kw = dict(dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=6, max_iterations=1500, wave=100000, seed=9)
    roots = sc["xml_points"][:3]
    fo = O.Forest(w, roots, sc["limits"], **kw)
    fo.run()
    fg = S.Forest(ctx, roots, sc["limits"], **kw)
    fg.run()
    assert fo.fingerprint() == fg.fingerprint() and fo.stats()["collide_calls"] == fg.stats()["collide_calls"]
    fg.close()
    ctx.close()


@pytest.mark.parametrize("mode", ["off", "coarse", "fine"])
@pytest.mark.parametrize("name", ["dense3d", "building"])
def test_clearance_bits_never_change_an_answer(S, monkeypatch, name, mode):
    """The clearance grid only skips work: with it disabled, very coarse (a few thousand cells) or fine,
    poses and edges that graze the surfaces must get the oracle's answers."""
    if mode == "off":
        monkeypatch.setenv("SFFGPU_NO_CLEARANCE", "1")
    elif mode == "coarse":
        monkeypatch.setenv("SFFGPU_CLEAR_CELLS", "4096")
    sc = common.scenario(name)
    ctx = S.Context(0)
    ctx.upload_robot(sc["robot"])       # robot first: the grid is built when the second mesh arrives
    ctx.upload_env(sc["env"])
    w = O.World(sc["env"], sc["robot"], O.TRIG_PORTABLE)
    rs = np.random.RandomState(31)
    # bounding-sphere radius of the robot: poses from far inside the clear cells to touching
    r = np.asarray(sc["robot"]).reshape(-1, 3)
    rr = float(np.linalg.norm(r - (r.min(0) + r.max(0)) / 2, axis=1).max())
    n = 1500
    p = np.concatenate([common.poses_near_surface(sc["env"], n // 4, 5, k * rr, 6) for k in (0.6, 1.5, 6.0, 40.0)])
    got = ctx.collide_poses(p)
    want = np.array([w.collide(q) for q in p], dtype=np.uint8)
    assert np.array_equal(got, want)
    assert 0.02 < want.mean() < 0.98
    a = p[: n // 2].copy()
    b = a.copy()
    d = rs.normal(0, 1, (len(a), 3))
    b[:, :3] += d / np.linalg.norm(d, axis=1)[:, None] * sc["sampling_dist"] * rs.uniform(0.2, 2.0, (len(a), 1))
    free, fh, ns = ctx.collide_segments(a, b)
    for i in range(len(a)):
        assert (free[i], fh[i], ns[i]) == w.path_free(a[i], b[i]), i
    # far outside the environment's box: outside the grid
    far = p[:50].copy()
    far[:, :3] += 1e4
    assert not ctx.collide_poses(far).any()
    ctx.close()
