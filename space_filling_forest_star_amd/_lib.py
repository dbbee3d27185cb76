"""ctypes binding of libsffgpu.so.  Names mirror the reference's interface for the hot path:
Context ~ Environment (+ the FLANN indices), Forest ~ SpaceForest (src/forest.h:31-54)."""
import ctypes as C
import os
import sys
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

EXPORTED_SYMBOLS = [
    "sffgpu_version", "sffgpu_device_count", "sffgpu_create", "sffgpu_destroy", "sffgpu_last_error",
    "sffgpu_mesh_upload", "sffgpu_collide_poses", "sffgpu_collide_segments", "sffgpu_sample_steer",
    "sffgpu_nodes_reset", "sffgpu_nodes_append", "sffgpu_nodes_count", "sffgpu_nodes_index", "sffgpu_radius", "sffgpu_knn",
    "sffgpu_forest_create", "sffgpu_forest_destroy", "sffgpu_forest_run", "sffgpu_forest_get_stats",
    "sffgpu_forest_get_nodes", "sffgpu_forest_get_borders", "sffgpu_forest_fingerprint", "sffgpu_forest_get_parent_history", "sffgpu_forest_paths",
    "sffgpu_forest_path_plan", "sffgpu_forest_smooth_paths",
    "sffgpu_rrt_create", "sffgpu_rrt_destroy", "sffgpu_rrt_run", "sffgpu_rrt_get_stats", "sffgpu_rrt_get_nodes",
    "sffgpu_rrt_get_links", "sffgpu_rrt_paths", "sffgpu_rrt_path_plan", "sffgpu_rrt_smooth_paths",
    "sffgpu_rrt_link_plan", "sffgpu_rrt_lazy_plan", "sffgpu_kernel_times", "sffgpu_forest_get_frontier",
    "sffgpu_collide_transforms", "sffgpu_ctx_set_stream", "sffgpu_rccl_unique_id", "sffgpu_ctx_rccl_init", "sffgpu_ctx_set_allgather", "sffgpu_forest_device_engine", "sffgpu_forest_exchange_bytes",
    "sffgpu_forest_rounds_per_wave", "sffgpu_forest_dev_wave_begin", "sffgpu_forest_dev_round_eval",
    "sffgpu_forest_dev_round_commit", "sffgpu_forest_dev_wave_end", "sffgpu_forest_in_wave", "sffgpu_forest_round_begin", "sffgpu_forest_round_records", "sffgpu_forest_round_commit",
]

# sffgpu_allgather_fn: int fn(void* user, const void* send_dev, void* recv_dev, size_t words_i32, void* hip_stream)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int32)
c_u8p = C.POINTER(C.c_uint8)
c_u64p = C.POINTER(C.c_uint64)


class SffGpuError(RuntimeError):
    pass


class ForestCfg(C.Structure):
    _fields_ = [("dim", C.c_int32), ("optimize", C.c_int32), ("has_goal", C.c_int32), ("goal", C.c_double * 6),
                ("limits", C.c_double * 6), ("dist_tree", C.c_double), ("sampling_dist", C.c_double),
                ("threshold_misses", C.c_int32), ("max_iterations", C.c_int32), ("node_budget", C.c_int32),
                ("wave", C.c_int32), ("seed", C.c_uint64), ("rank", C.c_int32), ("world", C.c_int32),
                ("priority_bias", C.c_double), ("libm_sampling", C.c_int32), ("record_parents", C.c_int32)]


class ForestStats(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("solved", C.c_int32), ("n_nodes", C.c_int32), ("n_trees", C.c_int32),
                ("frontier_size", C.c_int32), ("closed_size", C.c_int32), ("n_connected", C.c_int32),
                ("n_borders", C.c_int32), ("collide_calls", C.c_uint64), ("path_free_calls", C.c_uint64),
                ("nn_queries", C.c_uint64), ("waves", C.c_uint64), ("poses_executed", C.c_uint64),
                ("segments_executed", C.c_uint64), ("samples_executed", C.c_uint64), ("sweeps", C.c_uint64),
                ("sweep_nodes", C.c_uint64), ("sweep_queries", C.c_uint64), ("slow_path_samples", C.c_uint64), ("grid_rebuilds", C.c_uint64),
                ("sweep_ms", C.c_double),
                ("collide_ms", C.c_double), ("sample_ms", C.c_double), ("host_ms", C.c_double),
                ("total_ms", C.c_double), ("query_clock_ms", C.c_double), ("query_clock_launches", C.c_uint64),
                ("mate_overflow_requeries", C.c_uint64), ("star_rounds", C.c_uint64), ("star_passes", C.c_uint64),
                ("star_members", C.c_uint64), ("star_rewires", C.c_uint64), ("host_fallback_waves", C.c_uint64),
                ("commit_ms", C.c_double), ("exchange_ms", C.c_double), ("graph_launches", C.c_uint64),
                ("spec_steps", C.c_uint64), ("spec_evaluated", C.c_uint64), ("spec_committed", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class RrtCfg(C.Structure):
    _fields_ = [("dim", C.c_int32), ("optimize", C.c_int32), ("has_goal", C.c_int32), ("goal", C.c_double * 6),
                ("limits", C.c_double * 6), ("dist_tree", C.c_double), ("sampling_dist", C.c_double),
                ("priority_bias", C.c_double), ("max_iterations", C.c_int32), ("seed", C.c_uint64), ("wave", C.c_int32),
                ("lazy_edge", C.c_int32), ("rng_skip", C.c_uint64)]


class RrtStats(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("solved", C.c_int32), ("n_nodes", C.c_int32), ("n_live_trees", C.c_int32),
                ("merges", C.c_int32), ("n_links", C.c_int32), ("collide_calls", C.c_uint64),
                ("path_free_calls", C.c_uint64), ("nn_queries", C.c_uint64), ("total_ms", C.c_double),
                ("waves", C.c_uint64), ("speculated", C.c_uint64), ("committed", C.c_uint64), ("rng_draws", C.c_uint64),
                ("lazy_distance", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


NEED_HOST_EXCHANGE = 100   # sffgpu_forest_run: SFFGPU_NEED_HOST_EXCHANGE


def rccl_unique_id():
    """128-byte id for Context.rccl_init, made by rank 0 and shipped to the other ranks by the caller"""
    buf = (C.c_uint8 * 128)()
    if lib().sffgpu_rccl_unique_id(buf) != 0:
        raise SffGpuError("librccl could not be bound")
    return bytes(buf)


def lib_path():
    # SFFGPU_LIB: another build of the same library beside the shipped one (profiling builds with debug counters)
    return os.path.join(_HERE, os.environ.get("SFFGPU_LIB", "libsffgpu.so"))


def build_library(force=False):
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "csrc"), "clean"])
    subprocess.check_call(["make", "-s", "-j4", "-C", os.path.join(_HERE, "csrc")])
    return lib_path()


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise SffGpuError("libsffgpu.so is not built (%s): run __graft_entry__.build() or "
                          "`make -C space_filling_forest_star_amd/csrc`; there is no CPU fallback" % path)
    L = C.CDLL(path)
    L.sffgpu_version.restype = C.c_char_p
    L.sffgpu_last_error.restype = C.c_char_p
    L.sffgpu_last_error.argtypes = [C.c_void_p]
    L.sffgpu_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.sffgpu_destroy.argtypes = [C.c_void_p]
    L.sffgpu_mesh_upload.argtypes = [C.c_void_p, C.c_int, c_dp, C.c_int]
    L.sffgpu_collide_poses.argtypes = [C.c_void_p, c_dp, C.c_int, c_u8p]
    L.sffgpu_collide_transforms.argtypes = [C.c_void_p, c_dp, C.c_int, c_u8p]
    L.sffgpu_collide_segments.argtypes = [C.c_void_p, c_dp, c_dp, C.c_int, c_u8p, c_ip, c_ip]
    L.sffgpu_sample_steer.argtypes = [C.c_void_p, c_u64p, c_dp, C.c_int, C.c_double, C.c_int, c_dp, c_dp, c_u8p]
    L.sffgpu_nodes_reset.argtypes = [C.c_void_p, C.c_int]
    L.sffgpu_nodes_append.argtypes = [C.c_void_p, c_dp, c_ip, C.c_int]
    L.sffgpu_nodes_count.argtypes = [C.c_void_p]
    L.sffgpu_kernel_times.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.sffgpu_radius.argtypes = [C.c_void_p, c_dp, C.c_int, c_dp, c_ip, c_ip, c_ip, c_dp, c_ip, C.c_int]
    L.sffgpu_knn.argtypes = [C.c_void_p, c_dp, C.c_int, C.c_int, c_ip, c_ip, c_ip, c_dp, c_ip]
    L.sffgpu_nodes_index.argtypes = [C.c_void_p, c_dp, C.c_double]
    L.sffgpu_forest_create.argtypes = [C.c_void_p, C.POINTER(ForestCfg), c_dp, C.c_int, C.POINTER(C.c_void_p)]
    L.sffgpu_forest_destroy.argtypes = [C.c_void_p]
    L.sffgpu_forest_run.argtypes = [C.c_void_p, C.c_int]
    L.sffgpu_forest_get_stats.argtypes = [C.c_void_p, C.POINTER(ForestStats)]
    L.sffgpu_forest_get_nodes.argtypes = [C.c_void_p, c_dp, c_ip, c_ip, c_ip, c_dp, c_dp]
    L.sffgpu_forest_get_borders.argtypes = [C.c_void_p, c_ip, c_ip, c_ip, c_ip, c_dp, C.c_int]
    L.sffgpu_forest_paths.argtypes = [C.c_void_p, c_dp, c_ip, C.c_int]
    L.sffgpu_forest_smooth_paths.argtypes = [C.c_void_p, c_dp]
    L.sffgpu_forest_path_plan.argtypes = [C.c_void_p, C.c_int, C.c_int, c_ip, C.c_int]
    L.sffgpu_forest_fingerprint.restype = C.c_uint64
    L.sffgpu_forest_fingerprint.argtypes = [C.c_void_p]
    L.sffgpu_rrt_create.argtypes = [C.c_void_p, C.POINTER(RrtCfg), c_dp, C.c_int, C.POINTER(C.c_void_p)]
    L.sffgpu_rrt_destroy.argtypes = [C.c_void_p]
    L.sffgpu_rrt_run.argtypes = [C.c_void_p, C.c_int]
    L.sffgpu_rrt_get_stats.argtypes = [C.c_void_p, C.POINTER(RrtStats)]
    L.sffgpu_rrt_get_nodes.argtypes = [C.c_void_p, c_dp, c_ip, c_ip, c_ip, c_ip, c_dp, c_dp]
    L.sffgpu_rrt_get_links.argtypes = [C.c_void_p, c_ip, c_ip, c_ip, c_dp, C.c_int]
    L.sffgpu_rrt_paths.argtypes = [C.c_void_p, c_dp, c_ip, C.c_int]
    L.sffgpu_rrt_path_plan.argtypes = [C.c_void_p, C.c_int, C.c_int, c_ip, C.c_int]
    L.sffgpu_rrt_smooth_paths.argtypes = [C.c_void_p]
    L.sffgpu_rrt_link_plan.argtypes = [C.c_void_p, C.c_int, c_ip, C.c_int]
    L.sffgpu_rrt_lazy_plan.argtypes = [C.c_void_p, c_ip, C.c_int]
    L.sffgpu_forest_get_frontier.argtypes = [C.c_void_p, c_ip, C.c_int]
    L.sffgpu_forest_in_wave.argtypes = [C.c_void_p]
    L.sffgpu_ctx_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.sffgpu_rccl_unique_id.argtypes = [C.c_void_p]
    L.sffgpu_ctx_rccl_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.sffgpu_ctx_set_allgather.argtypes = [C.c_void_p, ALLGATHER_FN, C.c_void_p, C.c_int, C.c_int]
    L.sffgpu_forest_device_engine.argtypes = [C.c_void_p]
    L.sffgpu_forest_exchange_bytes.argtypes = [C.c_void_p]
    L.sffgpu_forest_exchange_bytes.restype = C.c_longlong
    L.sffgpu_forest_rounds_per_wave.argtypes = [C.c_void_p]
    L.sffgpu_forest_dev_wave_begin.argtypes = [C.c_void_p, c_ip]
    L.sffgpu_forest_dev_round_eval.argtypes = [C.c_void_p, C.c_void_p]
    L.sffgpu_forest_dev_round_commit.argtypes = [C.c_void_p, C.c_void_p]
    L.sffgpu_forest_dev_wave_end.argtypes = [C.c_void_p, c_ip]
    L.sffgpu_forest_round_begin.argtypes = [C.c_void_p, c_ip, c_ip]
    L.sffgpu_forest_round_records.argtypes = [C.c_void_p, c_ip, C.c_int]
    L.sffgpu_forest_round_commit.argtypes = [C.c_void_p, c_ip, C.c_int, c_ip, C.c_int]
    _LIB = L
    return L


def _f64(a, cols=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if cols is not None:
        a = a.reshape(-1, cols)
    return a


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _dp(a):
    return a.ctypes.data_as(c_dp)


def _ip(a):
    return None if a is None else a.ctypes.data_as(c_ip)


class Context:
    """One GPU: the collision world (robot + merged obstacles) and the node store."""

    def __init__(self, device=0):
        self._L = lib()
        h = C.c_void_p()
        rc = self._L.sffgpu_create(device, C.byref(h))
        if rc != 0:
            raise SffGpuError("sffgpu_create failed: %s" % self._L.sffgpu_last_error(None).decode())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            # (solver sessions hold a pointer to this context: they go first, whoever still references them)
            for child in list(getattr(self, "_children", ())):
                try:
                    child.close()
                except Exception:
                    pass
            self._L.sffgpu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _adopt(self, child):
        import weakref
        if not hasattr(self, "_children"):
            self._children = weakref.WeakSet()
        self._children.add(child)

    def _chk(self, rc):
        if rc < 0:
            raise SffGpuError("sffgpu error %d: %s" % (rc, self._L.sffgpu_last_error(self.h).decode()))
        return rc

    def upload_env(self, tri9):
        t = _f64(tri9, 9)
        self._chk(self._L.sffgpu_mesh_upload(self.h, 0, _dp(t), len(t)))

    def upload_robot(self, tri9):
        t = _f64(tri9, 9)
        self._chk(self._L.sffgpu_mesh_upload(self.h, 1, _dp(t), len(t)))

    def collide_poses(self, pos6):
        p = _f64(pos6, 6)
        out = np.zeros(len(p), np.uint8)
        self._chk(self._L.sffgpu_collide_poses(self.h, _dp(p), len(p), out.ctypes.data_as(c_u8p)))
        return out

    def collide_transforms(self, rt12):
        """robot placed by explicit transforms: rows of 9 rotation entries (row-major) + 3 translation entries"""
        p = _f64(rt12, 12)
        out = np.zeros(len(p), np.uint8)
        self._chk(self._L.sffgpu_collide_transforms(self.h, _dp(p), len(p), out.ctypes.data_as(c_u8p)))
        return out

    def collide_segments(self, a6, b6):
        a, b = _f64(a6, 6), _f64(b6, 6)
        n = len(a)
        free = np.zeros(n, np.uint8)
        fh = np.zeros(n, np.int32)
        ns = np.zeros(n, np.int32)
        self._chk(self._L.sffgpu_collide_segments(self.h, _dp(a), _dp(b), n, free.ctypes.data_as(c_u8p), _ip(fh), _ip(ns)))
        return free, fh, ns

    def sample_steer(self, words, center6, dist, dim, limits):
        w = np.ascontiguousarray(words, dtype=np.uint64).reshape(-1, 6)
        cen = _f64(center6, 6)
        lim = _f64(limits)
        out = np.zeros((len(w), 6))
        ok = np.zeros(len(w), np.uint8)
        self._chk(self._L.sffgpu_sample_steer(self.h, w.ctypes.data_as(c_u64p), _dp(cen), len(w), dist, dim, _dp(lim),
                                              _dp(out), ok.ctypes.data_as(c_u8p)))
        return out, ok

    def set_stream(self, hip_stream):
        """run the library's launches on the caller's HIP stream (e.g. torch.cuda.current_stream().cuda_stream)"""
        self._chk(self._L.sffgpu_ctx_set_stream(self.h, C.c_void_p(hip_stream)))

    def rccl_init(self, id128, rank, world):
        """give the context an RCCL communicator of its own (id128: the 128 bytes of rccl_unique_id() of rank 0)"""
        buf = (C.c_uint8 * 128)(*bytes(id128))
        self._chk(self._L.sffgpu_ctx_rccl_init(self.h, buf, rank, world))
        self.rccl = (rank, world)

    def set_allgather(self, fn, rank, world):
        """the library-driven exchange over the caller's collective (sffgpu_ctx_set_allgather): fn(send_dev, recv_dev,
        words, hip_stream) -> 0 on success, called wherever the library would enqueue ncclAllGather; None removes it"""
        if fn is None:
            self._chk(self._L.sffgpu_ctx_set_allgather(self.h, ALLGATHER_FN(0), None, 0, 1))
            self._xchg_cb, self.xchg = None, None
            return

        def tramp(_user, send, recv, words, stream):
            try:
                return int(fn(send, recv, int(words), stream) or 0)
            except Exception as e:   # (an exception must not unwind through the C frames)
                print("sffgpu: the caller's all-gather raised %r" % (e,), file=sys.stderr)
                return 1

        cb = ALLGATHER_FN(tramp)
        self._chk(self._L.sffgpu_ctx_set_allgather(self.h, cb, None, rank, world))
        self._xchg_cb, self.xchg = cb, (rank, world)   # (the trampoline lives as long as the context uses it)

    def nodes_reset(self, capacity=0):
        self._chk(self._L.sffgpu_nodes_reset(self.h, capacity))

    def nodes_append(self, pos6, tree_id):
        p = _f64(pos6, 6)
        t = _i32(tree_id)
        self._chk(self._L.sffgpu_nodes_append(self.h, _dp(p), _ip(t), len(p)))

    def nodes_index(self, limits, cell):
        """uniform grid over the store (Index::buildIndex): knn() without a tree filter then answers from the cells around
        each query instead of sweeping the store"""
        lim = _f64(limits)
        self._chk(self._L.sffgpu_nodes_index(self.h, _dp(lim), float(cell)))

    def nodes_count(self):
        return self._L.sffgpu_nodes_count(self.h)

    def kernel_times(self):
        """(ms[3], launches[3]) of the neighbour-query / collision / sampling kernels (HIP events)"""
        ms = (C.c_double * 3)()
        n = (C.c_uint64 * 3)()
        self._chk(self._L.sffgpu_kernel_times(self.h, ms, n))
        return list(ms), list(n)

    def radius(self, q6, r, tree=None, max_id=None, cap=256):
        q = _f64(q6, 6)
        nq = len(q)
        rr = _f64(np.broadcast_to(np.asarray(r, dtype=np.float64), (nq,)))
        tr = None if tree is None else _i32(np.broadcast_to(np.asarray(tree), (nq,)))
        mx = None if max_id is None else _i32(np.broadcast_to(np.asarray(max_id), (nq,)))
        idx = np.full((nq, cap), -1, np.int32)
        dist = np.zeros((nq, cap))
        cnt = np.zeros(nq, np.int32)
        self._chk(self._L.sffgpu_radius(self.h, _dp(q), nq, _dp(rr), _ip(tr), _ip(mx), _ip(idx), _dp(dist), _ip(cnt), cap))
        return idx, dist, cnt

    def knn(self, q6, k, tree=None, max_id=None):
        q = _f64(q6, 6)
        nq = len(q)
        tr = None if tree is None else _i32(np.broadcast_to(np.asarray(tree), (nq,)))
        mx = None if max_id is None else _i32(np.broadcast_to(np.asarray(max_id), (nq,)))
        idx = np.full((nq, k), -1, np.int32)
        dist = np.zeros((nq, k))
        cnt = np.zeros(nq, np.int32)
        self._chk(self._L.sffgpu_knn(self.h, _dp(q), nq, k, _ip(tr), _ip(mx), _ip(idx), _dp(dist), _ip(cnt)))
        return idx, dist, cnt


class Forest:
    """SpaceForest solver session (reference src/forest.h:31-54) on one Context."""

    def __init__(self, ctx, roots, limits, dist_tree, sampling_dist, dim=6, optimize=False, goal=None,
                 threshold_misses=5, max_iterations=100000, node_budget=0, wave=1, seed=1, rank=0, world=1,
                 priority_bias=0.0, libm_sampling=False, record_parents=False):
        self.ctx = ctx
        cfg = ForestCfg()
        cfg.priority_bias = priority_bias
        cfg.libm_sampling = int(libm_sampling)
        cfg.record_parents = int(record_parents)
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
        cfg.rank = rank
        cfg.world = world
        self.cfg = cfg
        r = _f64(roots, 6)
        h = C.c_void_p()
        ctx._chk(ctx._L.sffgpu_forest_create(ctx.h, C.byref(cfg), _dp(r), len(r), C.byref(h)))
        self.h = h
        ctx._adopt(self)

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None):     # (a context that is gone took its sessions with it)
                self.ctx._L.sffgpu_forest_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, max_waves=0):
        """returns True when a sharded wave has to be finished through round_begin / round_commit (native RCCL
        exchange, a bounded device list overflowed), else None"""
        rc = self.ctx._L.sffgpu_forest_run(self.h, max_waves)
        if rc == NEED_HOST_EXCHANGE:
            return True
        self.ctx._chk(rc)

    def stats(self):
        s = ForestStats()
        self.ctx._chk(self.ctx._L.sffgpu_forest_get_stats(self.h, C.byref(s)))
        return s.as_dict()

    def nodes(self):
        n = self.stats()["n_nodes"]
        pos = np.zeros((n, 6))
        parent = np.zeros(n, np.int32)
        tree = np.zeros(n, np.int32)
        it = np.zeros(n, np.int32)
        cost = np.zeros(n)
        dpar = np.zeros(n)
        self.ctx._chk(self.ctx._L.sffgpu_forest_get_nodes(self.h, _dp(pos), _ip(parent), _ip(tree), _ip(it), _dp(cost),
                                                          _dp(dpar)))
        return dict(pos=pos, parent=parent, tree=tree, iter=it, cost=cost, dpar=dpar)

    def borders(self, cap=1 << 20):
        ta, tb, n1, n2 = (np.zeros(cap, np.int32) for _ in range(4))
        d = np.zeros(cap)
        k = self.ctx._chk(self.ctx._L.sffgpu_forest_get_borders(self.h, _ip(ta), _ip(tb), _ip(n1), _ip(n2), _dp(d), cap))
        k = min(k, cap)
        return dict(ta=ta[:k].copy(), tb=tb[:k].copy(), n1=n1[:k].copy(), n2=n2[:k].copy(), dist=d[:k].copy())

    def fingerprint(self):
        return self.ctx._L.sffgpu_forest_fingerprint(self.h)

    def parent_history(self):
        """record_parents=True (SFF*): (node, parent, iteration) of every node creation and applied rewire, by iteration"""
        n = self.ctx._chk(self.ctx._L.sffgpu_forest_get_parent_history(self.h, None, None, None, 0))
        node, par, it = (np.zeros(n, np.int32) for _ in range(3))
        self.ctx._chk(self.ctx._L.sffgpu_forest_get_parent_history(self.h, _ip(node), _ip(par), _ip(it), n))
        return dict(node=node, parent=par, iter=it)

    def paths(self):
        """(cost matrix n_trees x n_trees, connected tree ids)"""
        n = self.stats()["n_trees"]
        d = np.zeros((n, n))
        conn = np.zeros(n, np.int32)
        k = self.ctx._chk(self.ctx._L.sffgpu_forest_paths(self.h, _dp(d), _ip(conn), n))
        return d, conn[:k].copy()

    def smooth(self):
        n = self.stats()["n_trees"]
        d = np.zeros((n, n))
        self.ctx._chk(self.ctx._L.sffgpu_forest_smooth_paths(self.h, _dp(d)))
        return d

    def plan(self, i, j, cap=1 << 16):
        ids = np.zeros(cap, np.int32)
        k = self.ctx._chk(self.ctx._L.sffgpu_forest_path_plan(self.h, i, j, _ip(ids), cap))
        return ids[:min(k, cap)].copy()

    def frontier(self, cap=1 << 22):
        ids = np.zeros(cap, np.int32)
        k = self.ctx._chk(self.ctx._L.sffgpu_forest_get_frontier(self.h, _ip(ids), cap))
        return ids[:min(k, cap)].copy()

    def in_wave(self):
        return bool(self.ctx._L.sffgpu_forest_in_wave(self.h))

    # ---- multi-GPU on the device-resident engine (include/sffgpu.h)
    def device_engine(self):
        return bool(self.ctx._L.sffgpu_forest_device_engine(self.h))

    def exchange_bytes(self):
        return int(self.ctx._L.sffgpu_forest_exchange_bytes(self.h))

    def rounds_per_wave(self):
        return int(self.ctx._L.sffgpu_forest_rounds_per_wave(self.h))

    def dev_wave_begin(self):
        done = C.c_int32(0)
        self.ctx._chk(self.ctx._L.sffgpu_forest_dev_wave_begin(self.h, C.byref(done)))
        return bool(done.value)

    def dev_round_eval(self, send_ptr):
        self.ctx._chk(self.ctx._L.sffgpu_forest_dev_round_eval(self.h, C.c_void_p(send_ptr)))

    def dev_round_commit(self, recv_ptr):
        self.ctx._chk(self.ctx._L.sffgpu_forest_dev_round_commit(self.h, C.c_void_p(recv_ptr)))

    def dev_wave_end(self):
        fault = C.c_int32(0)
        self.ctx._chk(self.ctx._L.sffgpu_forest_dev_wave_end(self.h, C.byref(fault)))
        return int(fault.value)

    # ---- multi-GPU round protocol (include/sffgpu.h "Multi-GPU wave protocol")
    def round_begin(self):
        """returns (records int32 array, done)"""
        n = C.c_int32(0)
        done = C.c_int32(0)
        self.ctx._chk(self.ctx._L.sffgpu_forest_round_begin(self.h, C.byref(n), C.byref(done)))
        rec = np.zeros(max(1, n.value), np.int32)
        if n.value:
            self.ctx._chk(self.ctx._L.sffgpu_forest_round_records(self.h, _ip(rec), n.value))
        return rec[:n.value], bool(done.value)

    def round_commit(self, all_words, words_per_rank):
        a = _i32(all_words)
        cnt = _i32(words_per_rank)
        self.ctx._chk(self.ctx._L.sffgpu_forest_round_commit(self.h, _ip(a), len(a), _ip(cnt), len(cnt)))


class Rrt:
    """RapidExpTree solver session (reference src/rrt.h:25-44) on one Context."""

    def __init__(self, ctx, roots, limits, dist_tree, sampling_dist, dim=6, optimize=False, goal=None,
                 priority_bias=0.0, max_iterations=10000, seed=1, wave=0, lazy_edge=False, rng_skip=0):
        self.ctx = ctx
        cfg = RrtCfg()
        cfg.wave = wave
        cfg.dim = dim
        cfg.optimize = int(optimize)
        cfg.has_goal = int(goal is not None and not lazy_edge)   # (lazy_edge: the goal is only a stopping test)
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
        r = _f64(roots, 6)
        h = C.c_void_p()
        ctx._chk(ctx._L.sffgpu_rrt_create(ctx.h, C.byref(cfg), _dp(r), len(r), C.byref(h)))
        self.h = h
        ctx._adopt(self)

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None):
                self.ctx._L.sffgpu_rrt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, max_iterations=0):
        self.ctx._chk(self.ctx._L.sffgpu_rrt_run(self.h, max_iterations))

    def stats(self):
        s = RrtStats()
        self.ctx._chk(self.ctx._L.sffgpu_rrt_get_stats(self.h, C.byref(s)))
        return s.as_dict()

    def nodes(self):
        n = self.stats()["n_nodes"]
        pos = np.zeros((n, 6))
        parent, tree, root_tree, it = (np.zeros(n, np.int32) for _ in range(4))
        cost = np.zeros(n)
        dpar = np.zeros(n)
        self.ctx._chk(self.ctx._L.sffgpu_rrt_get_nodes(self.h, _dp(pos), _ip(parent), _ip(tree), _ip(root_tree), _ip(it),
                                                       _dp(cost), _dp(dpar)))
        return dict(pos=pos, parent=parent, tree=tree, root_tree=root_tree, iter=it, cost=cost, dpar=dpar)

    def paths(self, n_trees):
        d = np.zeros((n_trees, n_trees))
        conn = np.zeros(n_trees, np.int32)
        k = self.ctx._chk(self.ctx._L.sffgpu_rrt_paths(self.h, _dp(d), _ip(conn), n_trees))
        return d, conn[:k].copy()

    def plan(self, i, j, cap=1 << 16):
        ids = np.zeros(cap, np.int32)
        k = self.ctx._chk(self.ctx._L.sffgpu_rrt_path_plan(self.h, i, j, _ip(ids), cap))
        return ids[:min(k, cap)].copy()

    def lazy_plan(self, cap=1 << 16):
        """lazy_edge: node ids root ... the node that reached the goal (empty while unsolved)"""
        ids = np.zeros(cap, np.int32)
        k = self.ctx._chk(self.ctx._L.sffgpu_rrt_lazy_plan(self.h, _ip(ids), cap))
        return ids[:min(k, cap)].copy()

    def smooth(self, cap=1 << 16):
        """RapidExpTree::smoothPaths on the central tree's link plans (after paths()); returns the plans"""
        n = self.ctx._chk(self.ctx._L.sffgpu_rrt_smooth_paths(self.h))
        out = []
        for k in range(n):
            ids = np.zeros(cap, np.int32)
            m = self.ctx._chk(self.ctx._L.sffgpu_rrt_link_plan(self.h, k, _ip(ids), cap))
            out.append(ids[:min(m, cap)].copy())
        return out

    def links(self, cap=1 << 16):
        t, n1, n2 = (np.zeros(cap, np.int32) for _ in range(3))
        d = np.zeros(cap)
        k = min(self.ctx._chk(self.ctx._L.sffgpu_rrt_get_links(self.h, _ip(t), _ip(n1), _ip(n2), _dp(d), cap)), cap)
        return dict(tree=t[:k].copy(), n1=n1[:k].copy(), n2=n2[:k].copy(), dist=d[:k].copy())


_XCHG = {"cap": 4096, "bufs": {}}


def exchange_records(local, group=None):
    """All-gather variable-length int32 record streams with torch.distributed (RCCL on GPUs, gloo on
    CPU): returns (concatenated stream in rank order, words per rank).

    One collective per round in the common case: every rank sends a fixed-capacity buffer
    [length, payload..., padding]; the capacity is a running bound shared by construction (every rank sees
    every length), and only when some stream does not fit is the gather repeated with a doubled capacity."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    local = np.ascontiguousarray(local, dtype=np.int32)
    n = len(local)
    while True:
        cap = _XCHG["cap"]
        key = (cap, world, str(dev))
        if key not in _XCHG["bufs"]:
            _XCHG["bufs"][key] = (torch.zeros(1 + cap, dtype=torch.int32).pin_memory() if dev.type == "cuda"
                                   else torch.zeros(1 + cap, dtype=torch.int32),
                                   torch.zeros(1 + cap, dtype=torch.int32, device=dev),
                                   torch.zeros(world * (1 + cap), dtype=torch.int32, device=dev),
                                   torch.zeros(world * (1 + cap), dtype=torch.int32).pin_memory() if dev.type == "cuda"
                                   else None)
        host, send, recv, hrecv = _XCHG["bufs"][key]
        hv = host.numpy()
        hv[0] = n
        m = min(n, cap)
        hv[1:1 + m] = local[:m]
        send.copy_(host, non_blocking=True)
        dist.all_gather_into_tensor(recv, send, group=group)
        if hrecv is not None:   # pinned landing buffer: one async copy + one stream sync (no pageable staging)
            hrecv.copy_(recv, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            allh = hrecv.numpy().reshape(world, 1 + cap)
        else:
            allh = recv.numpy().reshape(world, 1 + cap)
        counts = allh[:, 0].astype(np.int32)
        if int(counts.max()) <= cap:
            out = np.concatenate([allh[r, 1:1 + counts[r]] for r in range(world)]) if counts.sum() else np.zeros(0, np.int32)
            # keep the bound comfortable but tight (the whole buffer crosses PCIe twice per round): twice the
            # longest stream of this round, identically on every rank
            _XCHG["cap"] = max(1024, int(2 ** int(np.ceil(np.log2(2 * int(counts.max()) + 1)))))
            return out.astype(np.int32), counts
        _XCHG["cap"] = int(2 ** int(np.ceil(np.log2(int(counts.max()) + 1)))) * 2


def _run_distributed_device(forest, max_waves, group):
    """The device-resident engine over all ranks: per round one all-gather of fixed-size answer records between
    DEVICE buffers (RCCL on the stream the library launches on), one host synchronisation per wave."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    nccl = dist.get_backend(group) == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device())
    words = forest.exchange_bytes() // 4
    send = torch.zeros(words, dtype=torch.int32, device=dev)
    recv = torch.zeros(world * words, dtype=torch.int32, device=dev)
    # kernels and collectives in one order, no host hand-over: a stream of our own (torch's default stream is the
    # null stream, which the library cannot adopt) that is torch's current stream while the waves run
    stream = torch.cuda.Stream(device=dev)
    stream.wait_stream(torch.cuda.current_stream())
    forest.ctx.set_stream(stream.cuda_stream)
    rounds = forest.rounds_per_wave()
    w0 = forest.stats()["waves"]
    try:
        with torch.cuda.stream(stream):
            while True:
                if max_waves > 0:
                    s = forest.stats()
                    if s["waves"] - w0 >= max_waves and not forest.in_wave():
                        break
                if forest.dev_wave_begin():
                    break
                for _ in range(rounds):
                    forest.dev_round_eval(send.data_ptr())
                    if nccl:
                        dist.all_gather_into_tensor(recv, send, group=group)
                    else:   # (gloo in the CPU-side tests: staged through the host)
                        stream.synchronize()
                        parts = [torch.zeros(words, dtype=torch.int32) for _ in range(world)]
                        dist.all_gather(parts, send.cpu(), group=group)
                        recv.copy_(torch.cat(parts))
                    forest.dev_round_commit(recv.data_ptr())
                if forest.dev_wave_end():
                    # a bounded device list overflowed in this wave on every (identical) replica: finish it on the
                    # host protocol; the next dev_wave_begin moves the state back to the device
                    while forest.in_wave():
                        rec, done = forest.round_begin()
                        if done:
                            break
                        allw, counts = exchange_records(rec, group)
                        forest.round_commit(allw, counts)
    finally:
        stream.synchronize()
        forest.ctx.set_stream(None)
    return forest.stats()["waves"] - w0


def _native_rccl(forest, group):
    """True when the forest's context owns (or can be given) an RCCL communicator matching the process group: the
    library then drives the waves - kernels and ncclAllGather - by itself (no per-round Python)."""
    import torch
    import torch.distributed as dist
    # Opt-in (SFFGPU_NATIVE_RCCL=1): the library-driven exchange keeps a wave enqueued ahead and issues ncclAllGather by
    # itself; it has only ever run with ONE rank (this pool has single-GPU boxes), so the default stays the path the
    # multi-process tests cover (_run_distributed_device: torch.distributed's collective on the library's stream).
    if os.environ.get("SFFGPU_NATIVE_RCCL", "0") in ("", "0") or dist.get_backend(group) != "nccl":
        return False
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if getattr(forest.ctx, "rccl", None) == (rank, world):
        return True
    dev = torch.device("cuda", torch.cuda.current_device())

    def agreed(ok):   # every rank takes the same path
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        return bool(flag.item())

    # 1. can every rank bind librccl at all?  (decided BEFORE anybody enters the communicator's collective set-up: a
    #    rank that cannot would leave the others waiting in ncclCommInitRank)
    try:
        mine = rccl_unique_id()
        bound = True
    except Exception as e:   # (falls back to the torch.distributed exchange: say why)
        print("sffgpu: librccl cannot be bound on rank %d (%s); using the torch.distributed exchange" % (rank, e), file=sys.stderr)
        mine, bound = bytes(128), False
    if not agreed(bound):
        return False
    # 2. rank 0's id travels in a broadcast, every rank joins
    try:
        ident = torch.tensor(list(mine), dtype=torch.uint8, device=dev)
        dist.broadcast(ident, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        # (a rank that fails INSIDE ncclCommInitRank leaves its peers waiting there: that failure is fatal for the job,
        # which is why step 1 weeds out everything that can be known beforehand)
        forest.ctx.rccl_init(bytes(ident.cpu().tolist()), rank, world)
        ok = True
    except Exception as e:
        print("sffgpu: RCCL communicator set-up failed on rank %d (%s); using the torch.distributed exchange" % (rank, e), file=sys.stderr)
        ok = False
    return agreed(ok)


def host_staged_allgather(group):
    """A sffgpu_allgather_fn for process groups without device collectives (gloo): waits for the stream, stages this
    rank's words through the host, all-gathers them over the group and copies every rank's words back to the device -
    complete when it returns.  What the multi-process tests drive the library-driven exchange with on one GPU."""
    import torch
    import torch.distributed as dist
    hip = C.CDLL("libamdhip64.so")
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    world = dist.get_world_size(group)

    def fn(send, recv, words, stream):
        if hip.hipStreamSynchronize(C.c_void_p(stream)) != 0:
            return 1
        mine = torch.empty(words, dtype=torch.int32)
        if hip.hipMemcpy(C.c_void_p(mine.data_ptr()), C.c_void_p(send), words * 4, 2) != 0:   # hipMemcpyDeviceToHost
            return 1
        parts = [torch.empty(words, dtype=torch.int32) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        allw = torch.cat(parts).contiguous()
        return 0 if hip.hipMemcpy(C.c_void_p(recv), C.c_void_p(allw.data_ptr()), world * words * 4, 1) == 0 else 1   # HostToDevice

    return fn


def _run_distributed_native(forest, max_waves, group):
    w0 = forest.stats()["waves"]
    while True:
        left = 0
        if max_waves > 0:
            left = max_waves - (forest.stats()["waves"] - w0)
            if left <= 0 and not forest.in_wave():
                break
        if not forest.run(max(left, 1) if max_waves > 0 else 0):
            break
        # a bounded device list overflowed in this wave on every (identical) replica: finish it on the host protocol
        while forest.in_wave():
            rec, done = forest.round_begin()
            if done:
                break
            allw, counts = exchange_records(rec, group)
            forest.round_commit(allw, counts)
    return forest.stats()["waves"] - w0


def run_distributed(forest, max_waves=0, group=None):
    """Drive one shared forest over all ranks of the process group; returns waves done."""
    if forest.device_engine():
        import torch.distributed as dist
        # the library drives the waves itself when its context has a collective of this rank / world: the caller's
        # (Context.set_allgather) or an RCCL communicator of its own (_native_rccl)
        if getattr(forest.ctx, "xchg", None) == (dist.get_rank(group), dist.get_world_size(group)) or _native_rccl(forest, group):
            return _run_distributed_native(forest, max_waves, group)
        return _run_distributed_device(forest, max_waves, group)
    w0 = forest.stats()["waves"]
    while True:
        if max_waves > 0:
            s = forest.stats()
            # stop only at a wave boundary (frontier bookkeeping done), identically on every rank
            if s["waves"] - w0 >= max_waves and not forest.in_wave():
                break
        rec, done = forest.round_begin()
        if done:
            break
        allw, counts = exchange_records(rec, group)
        forest.round_commit(allw, counts)
    return forest.stats()["waves"] - w0
