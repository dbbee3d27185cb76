"""ctypes binding of the CPU ORACLE (oracle/libsff_oracle.so) — test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None

TRIG_LIBM, TRIG_PORTABLE = 0, 1

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int32)
c_u64p = C.POINTER(C.c_uint64)


class ForestCfg(C.Structure):
    _fields_ = [("dim", C.c_int), ("optimize", C.c_int), ("has_goal", C.c_int), ("goal", C.c_double * 6),
                ("limits", C.c_double * 6), ("dist_tree", C.c_double), ("sampling_dist", C.c_double),
                ("threshold_misses", C.c_int), ("max_iterations", C.c_int), ("node_budget", C.c_int),
                ("wave", C.c_int), ("seed", C.c_uint64), ("trig", C.c_int), ("priority_bias", C.c_double)]


class ForestStats(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("solved", C.c_int32), ("n_nodes", C.c_int32), ("n_trees", C.c_int32),
                ("frontier_size", C.c_int32), ("closed_size", C.c_int32), ("n_connected", C.c_int32),
                ("n_borders", C.c_int32), ("collide_calls", C.c_uint64), ("path_free_calls", C.c_uint64),
                ("nn_queries", C.c_uint64), ("waves", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class RrtCfg(C.Structure):
    _fields_ = [("dim", C.c_int), ("optimize", C.c_int), ("has_goal", C.c_int), ("goal", C.c_double * 6),
                ("limits", C.c_double * 6), ("dist_tree", C.c_double), ("sampling_dist", C.c_double),
                ("priority_bias", C.c_double), ("max_iterations", C.c_int), ("seed", C.c_uint64), ("trig", C.c_int),
                ("lazy_edge", C.c_int), ("rng_skip", C.c_uint64)]


class RrtStats(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("solved", C.c_int32), ("n_nodes", C.c_int32), ("n_live_trees", C.c_int32),
                ("merges", C.c_int32), ("n_links", C.c_int32), ("collide_calls", C.c_uint64),
                ("path_free_calls", C.c_uint64), ("nn_queries", C.c_uint64), ("rng_draws", C.c_uint64),
                ("lazy_distance", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libsff_oracle.so"])


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(ROOT, "oracle", "libsff_oracle.so")
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    L.sffo_parse_obj.argtypes = [C.c_char_p, c_dp, C.c_double, c_dp, C.c_int]
    L.sffo_parse_tri2d.argtypes = [C.c_char_p, c_dp, C.c_double, c_dp, C.c_int]
    L.sffo_distance.restype = C.c_double
    L.sffo_distance.argtypes = [c_dp, c_dp]
    L.sffo_steer.argtypes = [c_dp, c_dp, C.c_double, c_dp]
    L.sffo_rotation.argtypes = [c_dp, C.c_int, c_dp]
    for f in (L.sffo_sin, L.sffo_cos, L.sffo_acos):
        f.restype = C.c_double
        f.argtypes = [C.c_double, C.c_int]
    L.sffo_rng_create.restype = C.c_void_p
    L.sffo_rng_create.argtypes = [C.c_uint64, c_dp, C.c_int]
    L.sffo_rng_destroy.argtypes = [C.c_void_p]
    L.sffo_rng_raw.restype = C.c_uint64
    L.sffo_rng_raw.argtypes = [C.c_void_p]
    L.sffo_rng_int.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.sffo_rng_prob.restype = C.c_double
    L.sffo_rng_prob.argtypes = [C.c_void_p]
    L.sffo_rng_point_in_distance.argtypes = [C.c_void_p, c_dp, C.c_double, C.c_int, c_dp]
    L.sffo_rng_point_in_space.argtypes = [C.c_void_p, C.c_int, c_dp]
    L.sffo_sample_from_words.argtypes = [c_u64p, c_dp, C.c_double, C.c_int, c_dp, C.c_int, c_dp]
    L.sffo_world_create.restype = C.c_void_p
    L.sffo_world_create.argtypes = [c_dp, C.c_int, c_dp, C.c_int, C.c_int]
    L.sffo_world_destroy.argtypes = [C.c_void_p]
    L.sffo_tri_contact.argtypes = [c_dp, c_dp]
    L.sffo_collide_pose_brute.argtypes = [C.c_void_p, c_dp]
    L.sffo_collide_pose.argtypes = [C.c_void_p, c_dp]
    L.sffo_path_free.argtypes = [C.c_void_p, c_dp, c_dp, c_ip, c_ip]
    L.sffo_world_collide_calls.restype = C.c_uint64
    L.sffo_world_collide_calls.argtypes = [C.c_void_p]
    L.sffo_radius.argtypes = [c_dp, C.c_int, c_dp, C.c_double, c_ip, c_dp, C.c_int]
    L.sffo_knn.argtypes = [c_dp, C.c_int, c_dp, C.c_int, c_ip, c_dp]
    L.sffo_heap_script.argtypes = [c_dp, C.c_int, C.c_int, c_dp, c_ip, C.c_int, c_ip, c_ip, c_ip, C.c_int]
    L.sffo_forest_create.restype = C.c_void_p
    L.sffo_forest_create.argtypes = [C.c_void_p, C.POINTER(ForestCfg), c_dp, C.c_int]
    L.sffo_forest_destroy.argtypes = [C.c_void_p]
    L.sffo_forest_run.argtypes = [C.c_void_p, C.c_int]
    L.sffo_forest_get_stats.argtypes = [C.c_void_p, C.POINTER(ForestStats)]
    L.sffo_forest_get_nodes.argtypes = [C.c_void_p, c_dp, c_ip, c_ip, c_ip, c_dp, c_dp]
    L.sffo_forest_get_borders.argtypes = [C.c_void_p, c_ip, c_ip, c_ip, c_ip, c_dp, C.c_int]
    L.sffo_forest_paths.argtypes = [C.c_void_p, c_dp]
    L.sffo_forest_smooth.argtypes = [C.c_void_p, c_dp]
    L.sffo_forest_path_plan.argtypes = [C.c_void_p, C.c_int, C.c_int, c_ip, C.c_int]
    L.sffo_forest_fingerprint.restype = C.c_uint64
    L.sffo_forest_fingerprint.argtypes = [C.c_void_p]
    L.sffo_rrt_create.restype = C.c_void_p
    L.sffo_rrt_create.argtypes = [C.c_void_p, C.POINTER(RrtCfg), c_dp, C.c_int]
    L.sffo_rrt_destroy.argtypes = [C.c_void_p]
    L.sffo_rrt_run.argtypes = [C.c_void_p, C.c_int]
    L.sffo_rrt_get_stats.argtypes = [C.c_void_p, C.POINTER(RrtStats)]
    L.sffo_rrt_get_nodes.argtypes = [C.c_void_p, c_dp, c_ip, c_ip, c_ip, c_ip, c_dp, c_dp]
    L.sffo_rrt_paths.argtypes = [C.c_void_p, c_dp]
    L.sffo_rrt_path_plan.argtypes = [C.c_void_p, C.c_int, C.c_int, c_ip, C.c_int]
    L.sffo_rrt_get_links.argtypes = [C.c_void_p, c_ip, c_ip, c_ip, c_dp, C.c_int]
    L.sffo_rrt_smooth.argtypes = [C.c_void_p]
    L.sffo_rrt_link_plan.argtypes = [C.c_void_p, C.c_int, c_ip, C.c_int]
    L.sffo_rrt_lazy_plan.argtypes = [C.c_void_p, c_ip, C.c_int]
    _LIB = L
    return L


def dp(a):
    return a.ctypes.data_as(c_dp)


def ip(a):
    return a.ctypes.data_as(c_ip)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def parse_obj(path, pos=(0, 0, 0), scale=1.0, cap=200000):
    out = np.zeros((cap, 9))
    n = lib().sffo_parse_obj(path.encode(), dp(f64(pos)), scale, dp(out), cap)
    if n < 0:
        raise RuntimeError("parse_obj failed: %d" % n)
    return out[:n].copy()


def parse_tri2d(path, pos=(0, 0, 0), scale=1.0, cap=200000):
    out = np.zeros((cap, 9))
    n = lib().sffo_parse_tri2d(path.encode(), dp(f64(pos)), scale, dp(out), cap)
    if n < 0:
        raise RuntimeError("parse_tri2d failed: %d" % n)
    return out[:n].copy()


class Rng:
    def __init__(self, seed, limits, trig=TRIG_LIBM):
        self.h = lib().sffo_rng_create(seed, dp(f64(limits)), trig)

    def __del__(self):
        if getattr(self, "h", None):
            lib().sffo_rng_destroy(self.h)
            self.h = None

    def raw(self):
        return lib().sffo_rng_raw(self.h)

    def randint(self, lo, hi):
        return lib().sffo_rng_int(self.h, lo, hi)

    def prob(self):
        return lib().sffo_rng_prob(self.h)

    def point_in_distance(self, center, dist, dim):
        out = np.zeros(6)
        ok = lib().sffo_rng_point_in_distance(self.h, dp(f64(center)), dist, dim, dp(out))
        return ok, out

    def point_in_space(self, dim):
        out = np.zeros(6)
        lib().sffo_rng_point_in_space(self.h, dim, dp(out))
        return out


class World:
    def __init__(self, env_tri9, robot_tri9, trig=TRIG_PORTABLE):
        self.env = f64(env_tri9).reshape(-1, 9)
        self.robot = f64(robot_tri9).reshape(-1, 9)
        self.trig = trig
        self.h = lib().sffo_world_create(dp(self.env), len(self.env), dp(self.robot), len(self.robot), trig)

    def __del__(self):
        if getattr(self, "h", None):
            lib().sffo_world_destroy(self.h)
            self.h = None

    def collide(self, p):
        return lib().sffo_collide_pose(self.h, dp(f64(p)))

    def collide_brute(self, p):
        return lib().sffo_collide_pose_brute(self.h, dp(f64(p)))

    def collide_many(self, poses):
        poses = f64(poses).reshape(-1, 6)
        return np.array([self.collide(p) for p in poses], dtype=np.uint8)

    def path_free(self, a, b):
        fh = np.zeros(1, np.int32)
        ns = np.zeros(1, np.int32)
        r = lib().sffo_path_free(self.h, dp(f64(a)), dp(f64(b)), ip(fh), ip(ns))
        return r, int(fh[0]), int(ns[0])


class Forest:
    def __init__(self, world, roots, limits, dist_tree, sampling_dist, dim=6, optimize=False, goal=None,
                 threshold_misses=5, max_iterations=100000, node_budget=0, wave=1, seed=1, trig=None,
                 priority_bias=0.0):
        self.world = world
        cfg = ForestCfg()
        cfg.priority_bias = priority_bias
        cfg.dim = dim
        cfg.optimize = int(optimize)
        cfg.has_goal = int(goal is not None)
        if goal is not None:
            cfg.goal = (C.c_double * 6)(*goal)
        cfg.limits = (C.c_double * 6)(*limits)
        cfg.dist_tree = dist_tree
        cfg.sampling_dist = sampling_dist
        cfg.threshold_misses = threshold_misses
        cfg.max_iterations = max_iterations
        cfg.node_budget = node_budget
        cfg.wave = wave
        cfg.seed = seed
        cfg.trig = world.trig if trig is None else trig
        self.cfg = cfg
        roots = f64(roots).reshape(-1, 6)
        self.h = lib().sffo_forest_create(world.h, C.byref(cfg), dp(roots), len(roots))

    def __del__(self):
        if getattr(self, "h", None):
            lib().sffo_forest_destroy(self.h)
            self.h = None

    def run(self, max_waves=0):
        lib().sffo_forest_run(self.h, max_waves)

    def stats(self):
        s = ForestStats()
        lib().sffo_forest_get_stats(self.h, C.byref(s))
        return s.as_dict()

    def nodes(self):
        n = self.stats()["n_nodes"]
        pos = np.zeros((n, 6))
        parent = np.zeros(n, np.int32)
        tree = np.zeros(n, np.int32)
        it = np.zeros(n, np.int32)
        cost = np.zeros(n)
        dpar = np.zeros(n)
        lib().sffo_forest_get_nodes(self.h, dp(pos), ip(parent), ip(tree), ip(it), dp(cost), dp(dpar))
        return dict(pos=pos, parent=parent, tree=tree, iter=it, cost=cost, dpar=dpar)

    def borders(self, cap=1 << 20):
        ta, tb, n1, n2 = (np.zeros(cap, np.int32) for _ in range(4))
        d = np.zeros(cap)
        k = lib().sffo_forest_get_borders(self.h, ip(ta), ip(tb), ip(n1), ip(n2), dp(d), cap)
        k = min(k, cap)
        return dict(ta=ta[:k].copy(), tb=tb[:k].copy(), n1=n1[:k].copy(), n2=n2[:k].copy(), dist=d[:k].copy())

    def fingerprint(self):
        return lib().sffo_forest_fingerprint(self.h)

    def paths(self):
        n = self.stats()["n_trees"]
        d = np.zeros((n, n))
        lib().sffo_forest_paths(self.h, dp(d))
        return d

    def smooth(self):
        n = self.stats()["n_trees"]
        d = np.zeros((n, n))
        assert lib().sffo_forest_smooth(self.h, dp(d)) == n
        return d

    def plan(self, i, j, cap=1 << 16):
        ids = np.zeros(cap, np.int32)
        k = lib().sffo_forest_path_plan(self.h, i, j, ip(ids), cap)
        return ids[:min(k, cap)].copy()


class Rrt:
    def __init__(self, world, roots, limits, dist_tree, sampling_dist, dim=6, optimize=False, goal=None,
                 priority_bias=0.0, max_iterations=10000, seed=1, trig=None, lazy_edge=False, rng_skip=0):
        self.world = world
        cfg = RrtCfg()
        cfg.dim = dim
        cfg.optimize = int(optimize)
        cfg.has_goal = int(goal is not None and not lazy_edge)
        cfg.lazy_edge = int(lazy_edge)
        cfg.rng_skip = rng_skip
        if goal is not None:
            cfg.goal = (C.c_double * 6)(*goal)
        cfg.limits = (C.c_double * 6)(*limits)
        cfg.dist_tree = dist_tree
        cfg.sampling_dist = sampling_dist
        cfg.priority_bias = priority_bias
        cfg.max_iterations = max_iterations
        cfg.seed = seed
        cfg.trig = world.trig if trig is None else trig
        roots = f64(roots).reshape(-1, 6)
        self.h = lib().sffo_rrt_create(world.h, C.byref(cfg), dp(roots), len(roots))

    def __del__(self):
        if getattr(self, "h", None):
            lib().sffo_rrt_destroy(self.h)
            self.h = None

    def run(self, max_iters=0):
        lib().sffo_rrt_run(self.h, max_iters)

    def stats(self):
        s = RrtStats()
        lib().sffo_rrt_get_stats(self.h, C.byref(s))
        return s.as_dict()

    def nodes(self):
        n = self.stats()["n_nodes"]
        pos = np.zeros((n, 6))
        parent, tree, root_tree, it = (np.zeros(n, np.int32) for _ in range(4))
        cost = np.zeros(n)
        dpar = np.zeros(n)
        lib().sffo_rrt_get_nodes(self.h, dp(pos), ip(parent), ip(tree), ip(root_tree), ip(it), dp(cost), dp(dpar))
        return dict(pos=pos, parent=parent, tree=tree, root_tree=root_tree, iter=it, cost=cost, dpar=dpar)

    def links(self, cap=1 << 16):
        t, n1, n2 = (np.zeros(cap, np.int32) for _ in range(3))
        d = np.zeros(cap)
        k = min(lib().sffo_rrt_get_links(self.h, ip(t), ip(n1), ip(n2), dp(d), cap), cap)
        return dict(tree=t[:k].copy(), n1=n1[:k].copy(), n2=n2[:k].copy(), dist=d[:k].copy())

    def paths(self, n_trees):
        d = np.zeros((n_trees, n_trees))
        k = lib().sffo_rrt_paths(self.h, dp(d))
        return d, k

    def lazy_plan(self, cap=1 << 16):
        ids = np.zeros(cap, np.int32)
        k = lib().sffo_rrt_lazy_plan(self.h, ip(ids), cap)
        return ids[:k].copy()

    def plan(self, i, j, cap=1 << 16):
        ids = np.zeros(cap, np.int32)
        k = lib().sffo_rrt_path_plan(self.h, i, j, ip(ids), cap)
        return ids[:min(k, cap)].copy()

    def smooth(self, cap=1 << 16):
        """smoothPaths on the central tree's link plans (after paths()); returns the list of plans"""
        n = lib().sffo_rrt_smooth(self.h)
        out = []
        for k in range(n):
            ids = np.zeros(cap, np.int32)
            m = lib().sffo_rrt_link_plan(self.h, k, ip(ids), cap)
            out.append(ids[:m].copy())
        return out
