// engine.cpp — device context and batched primitive operations of libsffgpu.
#include "engine.h"
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>

#include "sff_geom.h"

namespace sff {

void hip_check(hipError_t e, const char* what) {
  if (e != hipSuccess) throw HipError{std::string(what) + ": " + hipGetErrorString(e)};
}
#define HIPCHK(x) hip_check((x), #x)

void DevBuf::ensure(size_t bytes) {
  if (bytes <= cap) return;
  size_t want = std::max(bytes, cap * 2);
  want = (want + 255) & ~(size_t)255;
  void* np = nullptr;
  HIPCHK(hipMalloc(&np, want));
  if (p) {
    HIPCHK(hipMemcpy(np, p, cap, hipMemcpyDeviceToDevice));
    (void)hipFree(p);
  }
  p = np;
  cap = want;
}
void DevBuf::release() {
  if (p) (void)hipFree(p);
  p = nullptr;
  cap = 0;
}
void PinBuf::ensure(size_t bytes) {
  if (bytes <= cap) return;
  size_t want = std::max(bytes, cap * 2);
  void* np = nullptr;
  HIPCHK(hipHostMalloc(&np, want, hipHostMallocDefault));
  if (p) {
    memcpy(np, p, cap);
    (void)hipHostFree(p);
  }
  p = np;
  cap = want;
}
void PinBuf::release() {
  if (p) (void)hipHostFree(p);
  p = nullptr;
  cap = 0;
}

// ------------------------------------------------------------------ RNG
void Mt64::reseed(uint64_t seed) {
  mt[0] = seed;
  for (int i = 1; i < 312; ++i) mt[i] = 6364136223846793005ULL * (mt[i - 1] ^ (mt[i - 1] >> 62)) + (uint64_t)i;
  idx = 312;
  draws = 0;
  q = nullptr;
  qh = qn = 0;
}
static inline uint64_t mt_step(uint64_t a, uint64_t b, uint64_t far) {
  const uint64_t x = (a & 0xFFFFFFFF80000000ULL) | (b & 0x7FFFFFFFULL);
  return far ^ (x >> 1) ^ ((0ULL - (x & 1ULL)) & 0xB5026F5AA96619E9ULL);
}
static inline uint64_t mt_temper(uint64_t y) {
  y ^= (y >> 29) & 0x5555555555555555ULL;
  y ^= (y << 17) & 0x71D67FFFEDA60000ULL;
  y ^= (y << 37) & 0xFFF7EEE000000000ULL;
  y ^= (y >> 43);
  return y;
}
void Mt64::twist() {
  for (int i = 0; i < 156; ++i) mt[i] = mt_step(mt[i], mt[i + 1], mt[i + 156]);
  for (int i = 156; i < 311; ++i) mt[i] = mt_step(mt[i], mt[i + 1], mt[i - 156]);
  mt[311] = mt_step(mt[311], mt[0], mt[155]);
  idx = 0;
}
uint64_t Mt64::next() {
  ++draws;
  if (qh < qn) return q[qh++];
  if (idx >= 312) twist();
  return mt_temper(mt[idx++]);
}
void Mt64::fill(uint64_t* out, size_t n) {
  draws += n;
  size_t k = 0;
  if (qh < qn) {
    k = std::min(n, qn - qh);
    memcpy(out, q + qh, k * sizeof(uint64_t));
    qh += k;
  }
  while (k < n) {
    if (idx >= 312) twist();
    const size_t take = std::min<size_t>(312 - idx, n - k);
    for (size_t j = 0; j < take; ++j) out[k + j] = mt_temper(mt[idx + j]);
    idx += (int)take;
    k += take;
  }
}
void Mt64::prefetch(uint64_t* buf, size_t want) {
  if (q != buf || qh > 0) {
    const size_t left = qn - qh;
    if (left) memmove(buf, q + qh, left * sizeof(uint64_t));
    q = buf;
    qh = 0;
    qn = left;
  }
  while (qn < want) {
    if (idx >= 312) twist();
    size_t take = std::min<size_t>(312 - idx, want - qn);
    for (size_t k = 0; k < take; ++k) buf[qn + k] = mt_temper(mt[idx + k]);
    idx += (int)take;
    qn += take;
  }
}
int Mt64::uniform_int(int lo, int hi) {
  uint64_t range = (uint64_t)((int64_t)hi - (int64_t)lo) + 1ULL;
  unsigned __int128 prod = (unsigned __int128)next() * range;
  uint64_t low = (uint64_t)prod;
  if (low < range) {
    uint64_t thr = (0ULL - range) % range;
    while (low < thr) {
      prod = (unsigned __int128)next() * range;
      low = (uint64_t)prod;
    }
  }
  return lo + (int)(uint64_t)(prod >> 64);
}

// ------------------------------------------------------------------ context
Ctx::Ctx(int dev) : device(dev) {
  if (const char* e = getenv("SFFGPU_TIMER_STRIDE")) timer_stride = std::max(1, atoi(e));
  int n = 0;
  HIPCHK(hipGetDeviceCount(&n));
  if (n <= 0) throw HipError{"no HIP device visible: libsffgpu has no CPU fallback"};
  if (dev < 0 || dev >= n) throw HipError{"device index out of range"};
  HIPCHK(hipSetDevice(dev));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, dev));
  if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
    throw HipError{std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only"};
  {
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) == hipSuccess && khz > 0) wall_clock_khz = khz;
  }
  HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  own_stream = stream;
  HIPCHK(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
  HIPCHK(hipEventCreateWithFlags(&ev_mid, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&ev_early, hipEventDisableTiming));
}

// ---- RCCL, bound at run time (the library has no link-time dependency on it; a copy already mapped by the host
// process - PyTorch ships one - is reused)
namespace {
struct RcclApi {
  struct Id128 { char b[128]; };   // ncclUniqueId (passed by value)
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, Id128, int) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};
RcclApi& rccl() {
  static RcclApi api;
  static bool tried = false;
  if (tried) return api;
  tried = true;
  void* h = nullptr;
  for (const char* name : {"librccl.so.1", "librccl.so"}) {
    h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
    if (h) break;
  }
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    if (h) break;
    h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
  }
  if (!h) return api;
  api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
  api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(h, "ncclAllGather"));
  api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  api.ok = api.GetUniqueId && api.CommInitRank && api.AllGather && api.CommDestroy;
  return api;
}
void rccl_check(int rc, const char* what) {
  if (rc == 0) return;
  const char* msg = rccl().GetErrorString ? rccl().GetErrorString(rc) : "?";
  throw HipError{std::string("rccl: ") + what + " failed: " + msg};
}
}  // namespace

void Ctx::rccl_unique_id(uint8_t* id128) {
  if (!rccl().ok) throw HipError{"rccl: librccl could not be bound"};
  rccl_check(rccl().GetUniqueId(id128), "ncclGetUniqueId");
}
void Ctx::rccl_init(const uint8_t* id128, int rank, int world) {
  if (!rccl().ok) throw HipError{"rccl: librccl could not be bound"};
  if (world < 1 || rank < 0 || rank >= world) throw HipError{"rccl: bad rank / world"};
  HIPCHK(hipSetDevice(device));
  if (rccl_comm) { (void)rccl().CommDestroy(rccl_comm); rccl_comm = nullptr; }
  RcclApi::Id128 id;
  memcpy(id.b, id128, 128);
  rccl_check(rccl().CommInitRank(&rccl_comm, world, id, rank), "ncclCommInitRank");
  rccl_rank = rank;
  rccl_world = world;
}
void Ctx::rccl_all_gather_i32(const void* send, void* recv, size_t words) {
  if (xchg_fn) {   // the caller's collective (sffgpu_ctx_set_allgather)
    if (xchg_fn(xchg_user, send, recv, words, stream) != 0) throw HipError{"exchange: the caller's all-gather failed"};
    return;
  }
  rccl_check(rccl().AllGather(send, recv, words, /* ncclInt32 */ 2, rccl_comm, stream), "ncclAllGather");
}

Ctx::~Ctx() {
  (void)hipSetDevice(device);
  // (the streams may still carry all-gathers of the communicator: drain them before it goes)
  if (stream) (void)hipStreamSynchronize(stream);
  if (own_stream && own_stream != stream) (void)hipStreamSynchronize(own_stream);
  if (copy_stream) (void)hipStreamSynchronize(copy_stream);
  if (rccl_comm && rccl().CommDestroy) { (void)rccl().CommDestroy(rccl_comm); rccl_comm = nullptr; }
  for (hipEvent_t e : rr_ev) if (e) (void)hipEventDestroy(e);
  for (SegJob& J : seg_job) if (J.ev) (void)hipEventDestroy(J.ev);
  if (ev_mid) (void)hipEventDestroy(ev_mid);
  if (ev_early) (void)hipEventDestroy(ev_early);
  for (auto& t : pending) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
  for (auto e : pool) (void)hipEventDestroy(e);
  DevBuf* bufs[] = {&env_tri, &env_box, &env_plane, &rob_tri, &sx, &sy, &sz, &syaw, &spitch, &sroll, &stree, &spos,
                    &d_a, &d_b, &d_c, &d_d, &d_e, &d_f, &d_g, &d_h, &r_in, &r_out, &r_out2, &r_q, &r_cnt, &r_hidx,
                    &r_hdist, &r_sega, &r_segb, &r_items, &r_items2, &r_center, &r_qrec, &env_clear, &env_clear_edge, &env_cand, &g_cnt, &g_items, &g_ovfcnt, &g_ovf, &t_cnt, &t_items, &t_ovfcnt, &t_ovf, &t_occ, &g_lite, &g_ovf_lite, &t_lite, &t_ovf_lite, &env_tg_start, &env_tg_list, &r_sub, &env_ext, &rr_q1, &rr_q2, &rr_a, &rr_out, &rr_sq, &rr_np, &rr_alt, &rr_q2b, &rr_sqb, &seg_job[0].ids, &seg_job[0].a, &seg_job[0].b, &seg_job[0].c, &seg_job[1].ids, &seg_job[1].a,
                    &seg_job[1].b, &seg_job[1].c, &seg_job[2].ids, &seg_job[2].a, &seg_job[2].b, &seg_job[2].c, &seg_job[3].ids, &seg_job[3].a,
                    &seg_job[3].b, &seg_job[3].c};
  for (DevBuf* b : bufs) b->release();
  for (auto& b : level_box) b.release();
  PinBuf* pins[] = {&h_a, &h_b, &h_c, &h_d, &h_e, &h_f, &h_g, &h_h, &p_in, &p_out, &rr_hq, &rr_hout, &seg_job[0].hids, &seg_job[0].hc, &seg_job[1].hids, &seg_job[1].hc, &seg_job[2].hids, &seg_job[2].hc,
                    &seg_job[3].hids, &seg_job[3].hc};
  for (PinBuf* b : pins) b->release();
  if (copy_stream) (void)hipStreamDestroy(copy_stream);
  if (own_stream) (void)hipStreamDestroy(own_stream);
}

// Run every launch of this context on the caller's stream (multi-GPU: the stream the RCCL collectives of a round are
// issued on, so that kernels and collectives are ordered without host synchronisation).
void Ctx::set_stream(hipStream_t s) {
  HIPCHK(hipSetDevice(device));
  sync();
  stream = s ? s : own_stream;
}

hipEvent_t Ctx::get_event() {
  if (!pool.empty()) {
    hipEvent_t e = pool.back();
    pool.pop_back();
    return e;
  }
  hipEvent_t e;
  HIPCHK(hipEventCreate(&e));
  return e;
}
// HIP events between kernels cost ~5-10 us of idle GPU each, so the forest engine brackets its kernels only on
// every timer_stride-th round (timing_on); the batch entry points always time.
void Ctx::time_begin(int kind) {
  kernel_calls[kind] += 1;
  if (round_scope) round_calls[kind] += 1;
  timed_now = timing_on;
  if (!timed_now) return;
  Timed t{get_event(), get_event(), kind, round_scope};
  HIPCHK(hipEventRecord(t.a, stream));
  pending.push_back(t);
}
void Ctx::time_end() { if (timed_now) HIPCHK(hipEventRecord(pending.back().b, stream)); }
double Ctx::kernel_ms_total(int kind) const {   // batch calls as measured + the sampled rounds scaled to all rounds
  double ms = kernel_ms[kind];
  if (round_timed[kind]) ms += round_ms[kind] * ((double)round_calls[kind] / (double)round_timed[kind]);
  return ms;
}
void Ctx::sync() {
  HIPCHK(hipStreamSynchronize(stream));
  for (auto& t : pending) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, t.a, t.b));
    if (t.round) { round_ms[t.kind] += ms; round_timed[t.kind] += 1; }
    else kernel_ms[t.kind] += ms;
    kernel_launches[t.kind] += 1;
    pool.push_back(t.a);
    pool.push_back(t.b);
  }
  pending.clear();
}

// ------------------------------------------------------------------ collision models
static inline uint64_t spread21(uint64_t v) {
  v &= 0x1FFFFF;
  v = (v | v << 32) & 0x1F00000000FFFFULL;
  v = (v | v << 16) & 0x1F0000FF0000FFULL;
  v = (v | v << 8) & 0x100F00F00F00F00FULL;
  v = (v | v << 4) & 0x10C30C30C30C30C3ULL;
  v = (v | v << 2) & 0x1249249249249249ULL;
  return v;
}

// Replaces RAPID_model::BeginModel/AddTri/EndModel (src/environment.h:101-115,222).  The
// environment hierarchy is 64-ary so that one wavefront tests all children of a node at once:
// triangles are ordered along a Morton curve of their box centres, 64 consecutive triangles
// form a level-0 group, 64 consecutive groups a level-1 group, and so on until <= 64 remain.
void Ctx::upload_mesh(int role, const double* tri9, int n) {
  HIPCHK(hipSetDevice(device));
  if (role == SFFGPU_MESH_ROBOT) {
    if (n <= 0) throw HipError{"robot mesh must have at least one triangle"};
    rob_tri.ensure((size_t)n * 9 * sizeof(double));
    HIPCHK(hipMemcpy(rob_tri.p, tri9, (size_t)n * 9 * sizeof(double), hipMemcpyHostToDevice));
    robv.tri = rob_tri.as<double>();
    robv.n_tri = n;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int v = 0; v < n * 3; ++v)
      for (int i = 0; i < 3; ++i) {
        lo[i] = std::min(lo[i], tri9[3 * v + i]);
        hi[i] = std::max(hi[i], tri9[3 * v + i]);
      }
    double rad = 0;
    for (int i = 0; i < 3; ++i) { robv.lo[i] = lo[i]; robv.hi[i] = hi[i]; robv.center[i] = 0.5 * (lo[i] + hi[i]); }
    for (int v = 0; v < n * 3; ++v) {
      double d2 = 0;
      for (int i = 0; i < 3; ++i) d2 += (tri9[3 * v + i] - robv.center[i]) * (tri9[3 * v + i] - robv.center[i]);
      rad = std::max(rad, std::sqrt(d2));
    }
    robv.radius = rad;
    have_robot = true;
    h_rob.assign(tri9, tri9 + (size_t)n * 9);
    build_robot_extents();
    build_clearance();
    build_tri_grid();
    return;
  }
  if (role != SFFGPU_MESH_ENV) throw HipError{"unknown mesh role"};
  envv = sffk::EnvView{};
  have_env = true;
  clear_cells = 0;
  if (n <= 0) return;  // HasMap == false
  std::vector<double> box((size_t)n * 6);
  double glo[3] = {1e300, 1e300, 1e300}, ghi[3] = {-1e300, -1e300, -1e300};
  for (int t = 0; t < n; ++t)
    for (int a = 0; a < 3; ++a) {
      const double* P = tri9 + 9 * (size_t)t;
      double lo = std::min(P[a], std::min(P[3 + a], P[6 + a])), hi = std::max(P[a], std::max(P[3 + a], P[6 + a]));
      box[6 * (size_t)t + a] = lo;
      box[6 * (size_t)t + 3 + a] = hi;
      glo[a] = std::min(glo[a], lo);
      ghi[a] = std::max(ghi[a], hi);
    }
  for (int a = 0; a < 3; ++a) { env_lo[a] = glo[a]; env_hi[a] = ghi[a]; }
  env_maxabs = 1.0;
  for (int a = 0; a < 3; ++a) env_maxabs = std::max(env_maxabs, std::max(std::fabs(glo[a]), std::fabs(ghi[a])));
  std::vector<std::pair<uint64_t, int>> order(n);
  for (int t = 0; t < n; ++t) {
    uint64_t code = 0;
    for (int a = 0; a < 3; ++a) {
      double ext = ghi[a] - glo[a];
      double c = 0.5 * (box[6 * (size_t)t + a] + box[6 * (size_t)t + 3 + a]);
      double u = ext > 0 ? (c - glo[a]) / ext : 0.0;
      uint64_t q = (uint64_t)std::min(2097151.0, std::max(0.0, u * 2097151.0));
      code |= spread21(q) << a;
    }
    order[t] = {code, t};
  }
  std::sort(order.begin(), order.end());
  std::vector<double> tri_s((size_t)n * 9), box_s((size_t)n * 6);
  for (int k = 0; k < n; ++k) {
    memcpy(&tri_s[9 * (size_t)k], tri9 + 9 * (size_t)order[k].second, 9 * sizeof(double));
    memcpy(&box_s[6 * (size_t)k], &box[6 * (size_t)order[k].second], 6 * sizeof(double));
  }
  env_tri.ensure(tri_s.size() * sizeof(double));
  env_box.ensure(box_s.size() * sizeof(double));
  HIPCHK(hipMemcpy(env_tri.p, tri_s.data(), tri_s.size() * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(env_box.p, box_s.data(), box_s.size() * sizeof(double), hipMemcpyHostToDevice));
  std::vector<double> plane((size_t)n * 5);
  for (int k = 0; k < n; ++k) {
    const double* P = &tri_s[9 * (size_t)k];
    double e1[3] = {P[3] - P[0], P[4] - P[1], P[5] - P[2]}, e2[3] = {P[6] - P[0], P[7] - P[1], P[8] - P[2]}, nn[3];
    sffg::cross(e1, e2, nn);
    plane[5 * (size_t)k + 0] = nn[0];
    plane[5 * (size_t)k + 1] = nn[1];
    plane[5 * (size_t)k + 2] = nn[2];
    plane[5 * (size_t)k + 3] = (nn[0] * P[0] + nn[1] * P[1]) + nn[2] * P[2];
    plane[5 * (size_t)k + 4] = std::sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
  }
  env_plane.ensure(plane.size() * sizeof(double));
  HIPCHK(hipMemcpy(env_plane.p, plane.data(), plane.size() * sizeof(double), hipMemcpyHostToDevice));
  envv.tri = env_tri.as<double>();
  envv.tri_box = env_box.as<double>();
  envv.tri_plane = env_plane.as<double>();
  envv.n_tri = n;
  h_plane = plane;
  // levels
  std::vector<double> cur = box_s;
  int count = n, L = 0;
  while (true) {
    int groups = (count + 63) / 64;
    std::vector<double> nxt((size_t)groups * 6);
    for (int g = 0; g < groups; ++g) {
      double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
      for (int k = g * 64; k < std::min(count, g * 64 + 64); ++k)
        for (int a = 0; a < 3; ++a) {
          lo[a] = std::min(lo[a], cur[6 * (size_t)k + a]);
          hi[a] = std::max(hi[a], cur[6 * (size_t)k + 3 + a]);
        }
      for (int a = 0; a < 3; ++a) { nxt[6 * (size_t)g + a] = lo[a]; nxt[6 * (size_t)g + 3 + a] = hi[a]; }
    }
    if (L >= SFFK_MAX_LEVELS) throw HipError{"environment mesh too large for the box hierarchy"};
    level_box[L].ensure(nxt.size() * sizeof(double));
    HIPCHK(hipMemcpy(level_box[L].p, nxt.data(), nxt.size() * sizeof(double), hipMemcpyHostToDevice));
    envv.level_box[L] = level_box[L].as<double>();
    envv.level_count[L] = groups;
    ++L;
    cur.swap(nxt);
    count = groups;
    if (groups <= 64) break;
  }
  envv.n_levels = L;
  build_robot_extents();
  build_clearance();
  build_tri_grid();
}

// Extent of the un-rotated robot along every environment triangle's normal (kernels.h, EnvView::tri_ext): edge samples
// carry no rotation (src/problemStruct.h:157-165), so "the whole robot on one side of the triangle's plane" is one
// comparison per (sample, triangle) in the exact kernel.
void Ctx::build_robot_extents() {
  envv.tri_ext = nullptr;
  envv.cand = nullptr;
  if (!have_env || !have_robot || envv.n_tri <= 0 || h_rob.empty() || h_plane.size() != (size_t)envv.n_tri * 5) return;
  std::vector<double> ext((size_t)envv.n_tri * 2);
  const size_t nv = h_rob.size() / 3;
  for (int k = 0; k < envv.n_tri; ++k) {
    const double* pl = &h_plane[5 * (size_t)k];
    double lo = 1e300, hi = -1e300;
    for (size_t v = 0; v < nv; ++v) {
      const double d = (pl[0] * h_rob[3 * v] + pl[1] * h_rob[3 * v + 1]) + pl[2] * h_rob[3 * v + 2];
      lo = std::min(lo, d);
      hi = std::max(hi, d);
    }
    ext[2 * (size_t)k] = lo;
    ext[2 * (size_t)k + 1] = hi;
  }
  env_ext.ensure(ext.size() * sizeof(double));
  HIPCHK(hipMemcpy(env_ext.p, ext.data(), ext.size() * sizeof(double), hipMemcpyHostToDevice));
  envv.tri_ext = env_ext.as<double>();
  // everything the exact kernels stage per candidate triangle, as one record (kernels.h, EnvView::cand)
  std::vector<double> tb((size_t)envv.n_tri * 6), tt((size_t)envv.n_tri * 9), rec((size_t)envv.n_tri * 22);
  HIPCHK(hipMemcpy(tb.data(), env_box.p, tb.size() * sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(tt.data(), env_tri.p, tt.size() * sizeof(double), hipMemcpyDeviceToHost));
  for (int k = 0; k < envv.n_tri; ++k) {
    double* r = &rec[22 * (size_t)k];
    memcpy(r, &tb[6 * (size_t)k], 6 * sizeof(double));
    memcpy(r + 6, &h_plane[5 * (size_t)k], 5 * sizeof(double));
    memcpy(r + 11, &tt[9 * (size_t)k], 9 * sizeof(double));
    r[20] = ext[2 * (size_t)k];
    r[21] = ext[2 * (size_t)k + 1];
  }
  env_cand.ensure(rec.size() * sizeof(double));
  HIPCHK(hipMemcpy(env_cand.p, rec.data(), rec.size() * sizeof(double), hipMemcpyHostToDevice));
  envv.cand = env_cand.as<double>();
  if (const char* e = getenv("SFFGPU_NO_CAND")) if (atoi(e)) envv.cand = nullptr;   // (A/B: the four-array gather)
}

// Clearance bits over the environment box (kernels.h, EnvView): one bit per cell, set when a robot whose
// bounding-sphere centre lies anywhere in the cell cannot touch a triangle.  The cell edge follows the
// robot radius but the grid is capped at 2^27 cells (16 MB of bits) and by the build cost.
void Ctx::build_clearance() {
  envv.clear_bits = nullptr;
  envv.clear_bits_edge = nullptr;
  clear_cells = 0;
  if (!have_env || !have_robot || envv.n_tri <= 0) return;
  if (const char* e = getenv("SFFGPU_NO_CLEARANCE")) if (atoi(e)) return;
  // radius about the model origin: the bounding sphere (centre c, radius r) in any rotation stays inside |c| + r
  const double rr = robv.radius + std::sqrt(robv.center[0] * robv.center[0] + robv.center[1] * robv.center[1] +
                                            robv.center[2] * robv.center[2]) * (1 + 1e-9);
  double ext[3], vol_ext = 0;
  for (int a = 0; a < 3; ++a) { ext[a] = env_hi[a] - env_lo[a]; vol_ext = std::max(vol_ext, ext[a]); }
  if (!(vol_ext > 0) || !(rr >= 0)) return;
  double cap = 134217728.0;
  if (const char* e = getenv("SFFGPU_CLEAR_CELLS")) cap = std::max(512.0, atof(e));
  cap = std::min(cap, std::max(32768.0, 4e10 / (double)std::max(1, envv.level_count[0])));
  double hdiv = 2.0;
  if (const char* e = getenv("SFFGPU_CLEAR_HDIV")) hdiv = std::max(0.5, atof(e));
  double h = std::max(rr / hdiv, vol_ext * 1e-4);
  int n[3];
  double thr = 0, lin_slack = 0;
  const double reach = 0.4 * (1 + 1e-6);
  for (int it = 0; it < 64; ++it) {
    const double halfdiag = 0.5 * std::sqrt(3.0) * h;
    const double m0 = rr + halfdiag;
    // additive slack of every test about a cell centre: rounding of the centre / the sample, and (below) the fp32 placement
    lin_slack = 1e-8 * (3 * (env_maxabs + 2 * m0) + 1);
    thr = rr * (1 + 1e-9) + halfdiag * (1 + 1e-5) + lin_slack;
    double cells = 1, nmax = 1;
    for (int pass = 0; pass < 2; ++pass) {
      cells = 1;
      for (int a = 0; a < 3; ++a) {
        double c = std::ceil((ext[a] + 2 * (thr + reach)) / h) + 1;
        n[a] = (int)std::min(c, 1e9);
        cells *= c;
        nmax = std::max(nmax, c);
      }
      // the neighbour-query kernel places edge samples in fp32 cell units (k_query_classify): three roundings of
      // at most 2^-24 * cells-per-axis each; the bits cover 1e-6 * cells-per-axis cells of misplacement
      // ... and tests ONE sample for a group of eight consecutive ones: they lie within four sample spacings
      // (4 x 0.1 scaled units, src/problemStruct.h:121) of it, so the EDGE plane holds that reach on top
      if (pass == 0) { const double fp32 = 1e-6 * (nmax + 2) * h; thr += fp32; lin_slack += fp32; }
    }
    if (cells <= cap) break;
    h *= std::max(1.02, std::cbrt(cells / cap));
  }
  const long long cells = (long long)n[0] * n[1] * n[2];
  if (cells <= 0 || cells > (1LL << 31)) return;
  // (the grid spans the environment's box inflated by the larger of the two sphere radii: outside it everything is clear)
  for (int a = 0; a < 3; ++a) { envv.clear_org[a] = env_lo[a] - (thr + reach); envv.clear_n[a] = n[a]; }
  envv.clear_inv = 1.0 / h;
  sffk::ClearBuildArgs P{};
  P.thr_pose = thr;
  P.thr_edge = thr + reach;
  const double cell_half = 0.5 * h * (1 + 1e-5) + lin_slack;
  for (int a = 0; a < 3; ++a) {
    // pose plane: the robot in any rotation stays inside the cube of the bounding radius about its model origin
    P.pose_lo[a] = -(cell_half + rr * (1 + 1e-9));
    P.pose_hi[a] = cell_half + rr * (1 + 1e-9);
    // edge plane: the un-rotated robot's own box about the model origin, + the reach of a group of eight samples
    P.edge_lo[a] = -(cell_half + reach) + std::min(0.0, robv.lo[a]) * (1 + 1e-9);
    P.edge_hi[a] = cell_half + reach + std::max(0.0, robv.hi[a]) * (1 + 1e-9);
  }
  const long long padded = (cells + 255) / 256 * 256;
  env_clear.ensure((size_t)(padded / 8));
  env_clear_edge.ensure((size_t)(padded / 8));
  const auto t_build = std::chrono::steady_clock::now();
  sffk::launch_clear_build(stream, envv, P, env_clear.as<uint32_t>(), env_clear_edge.as<uint32_t>(), cells);
  HIPCHK(hipStreamSynchronize(stream));
  if (getenv("SFFGPU_PROFILE"))
    fprintf(stderr, "[sffgpu clearance bits] both planes built in %.2f ms\n",
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_build).count());
  envv.clear_bits = env_clear.as<uint32_t>();
  envv.clear_bits_edge = env_clear_edge.as<uint32_t>();
  clear_cells = cells;
  if (getenv("SFFGPU_PROFILE")) {   // share of blocked cells per plane
    std::vector<uint32_t> w((size_t)(padded / 32));
    long long blocked[2] = {0, 0};
    for (int pl = 0; pl < 2; ++pl) {
      HIPCHK(hipMemcpy(w.data(), pl ? env_clear_edge.p : env_clear.p, w.size() * 4, hipMemcpyDeviceToHost));
      for (long long k = 0; k < cells; ++k) blocked[pl] += !((w[(size_t)(k >> 5)] >> (k & 31)) & 1u);
    }
    fprintf(stderr, "[sffgpu clearance bits] %lld cells of edge %.3f (%d x %d x %d), blocked: pose plane %.2f %%, edge plane %.2f %%\n", cells, h,
            n[0], n[1], n[2], 100.0 * (double)blocked[0] / (double)cells, 100.0 * (double)blocked[1] / (double)cells);
  }
}

// Triangle grid over the environment's box (kernels.h, EnvView): cell edge = a third of the largest query box the
// collision kernels ask with (robot extent + one 64-sample chunk of an edge, 6.4 units), but at most 128 cells per
// axis.  Built on the device in two passes around a host prefix sum.
void Ctx::build_tri_grid() {
  envv.tg_start = nullptr;
  envv.tg_list = nullptr;
  if (!have_env || !have_robot || envv.n_tri <= 0 || envv.n_levels < 1) return;
  if (const char* e = getenv("SFFGPU_NO_TRIGRID")) if (atoi(e)) return;
  double rext = 0, ext = 0;
  for (int a = 0; a < 3; ++a) {
    rext = std::max(rext, robv.hi[a] - robv.lo[a]);
    ext = std::max(ext, env_hi[a] - env_lo[a]);
  }
  rext = std::max(rext, 2 * robv.radius);
  if (!(ext > 0)) return;
  double div = 3.0;
  if (const char* e = getenv("SFFGPU_TG_DIV")) div = std::max(1.0, atof(e));
  const double h = std::max((rext + 6.4) / div, ext / (div > 3.0 ? 256.0 : 128.0));
  long long cells = 1;
  for (int a = 0; a < 3; ++a) {
    envv.tg_org[a] = env_lo[a];
    envv.tg_n[a] = std::max(1, (int)std::ceil((env_hi[a] - env_lo[a]) / h + 1e-9));
    cells *= envv.tg_n[a];
  }
  envv.tg_inv = 1.0 / h;
  if (cells > (1LL << 24)) return;
  env_tg_start.ensure((size_t)(cells + 1) * 4);
  sffk::launch_tgrid_build(stream, envv, env_tg_start.as<int32_t>(), nullptr, false);
  std::vector<int32_t> cnt((size_t)cells + 1, 0);
  HIPCHK(hipMemcpyAsync(cnt.data(), env_tg_start.p, (size_t)cells * 4, hipMemcpyDeviceToHost, stream));
  HIPCHK(hipStreamSynchronize(stream));
  long long run = 0;
  for (long long c = 0; c < cells; ++c) { const int32_t m = cnt[(size_t)c]; cnt[(size_t)c] = (int32_t)run; run += m; }
  cnt[(size_t)cells] = (int32_t)run;
  if (run > (1LL << 30)) return;
  env_tg_list.ensure((size_t)std::max<long long>(run, 1) * 4);
  HIPCHK(hipMemcpyAsync(env_tg_start.p, cnt.data(), (size_t)(cells + 1) * 4, hipMemcpyHostToDevice, stream));
  sffk::launch_tgrid_build(stream, envv, env_tg_start.as<int32_t>(), env_tg_list.as<int32_t>(), true);
  HIPCHK(hipStreamSynchronize(stream));
  envv.tg_start = env_tg_start.as<int32_t>();
  envv.tg_list = env_tg_list.as<int32_t>();
}

// ------------------------------------------------------------------ node store
sffk::NodeStoreView Ctx::store_view() const {
  return sffk::NodeStoreView{sx.as<float>(), sy.as<float>(), sz.as<float>(), syaw.as<float>(), spitch.as<float>(),
                             sroll.as<float>(), stree.as<int32_t>(), spos.as<double>()};
}
static sffk::NodeStoreMut store_mut(const Ctx& c) {
  return sffk::NodeStoreMut{c.sx.as<float>(), c.sy.as<float>(), c.sz.as<float>(), c.syaw.as<float>(),
                            c.spitch.as<float>(), c.sroll.as<float>(), c.stree.as<int32_t>(), c.spos.as<double>()};
}
void Ctx::store_reserve(int capacity) {
  if (capacity <= store_cap) return;
  HIPCHK(hipSetDevice(device));
  sync();
  int cap = std::max(capacity, store_cap * 2);
  cap = (cap + 1023) & ~1023;
  DevBuf* cols[] = {&sx, &sy, &sz, &syaw, &spitch, &sroll};
  for (DevBuf* b : cols) b->ensure((size_t)cap * sizeof(float));
  stree.ensure((size_t)cap * sizeof(int32_t));
  spos.ensure((size_t)cap * 6 * sizeof(double));
  store_cap = cap;
}
void Ctx::store_reset(int capacity) {
  store_n = 0;
  grid_on = false;
  grid_inserted = 0;
  store_maxabs = 1.0;
  store_reserve(std::max(capacity, 1024));
}
void Ctx::store_append(const double* pos6, const int32_t* tree, int n, bool wait) {
  if (n <= 0) return;
  HIPCHK(hipSetDevice(device));
  store_reserve(store_n + n);
  // one packed block: positions, then tree ids
  const size_t pb = (size_t)n * 6 * sizeof(double), tb = (size_t)n * sizeof(int32_t);
  h_a.ensure(pb + tb);
  memcpy(h_a.p, pos6, pb);
  memcpy(h_a.as<char>() + pb, tree, tb);
  for (int i = 0; i < n * 6; ++i)
    if (i % 6 < 3) store_maxabs = std::max(store_maxabs, std::fabs(pos6[i]));
  d_a.ensure(pb + tb);
  HIPCHK(hipMemcpyAsync(d_a.p, h_a.p, pb + tb, hipMemcpyHostToDevice, stream));
  // with an up-to-date grid the new nodes are inserted by the same launch
  const bool fuse_grid = grid_on && grid_inserted == store_n;
  sffk::launch_store_write(stream, store_mut(*this), d_a.as<double>(), reinterpret_cast<const int32_t*>(d_a.as<char>() + pb),
                           nullptr, nullptr, n, store_n,
                           fuse_grid ? &gridv : nullptr);
  if (wait) sync();   // callers that keep the stream ordered (the forest engine) skip the wait
  store_n += n;
  if (fuse_grid) grid_inserted = store_n;
}

void Ctx::store_set_tree(const int32_t* ids, int n, int32_t tree) {
  if (n <= 0) return;
  HIPCHK(hipSetDevice(device));
  h_a.ensure((size_t)n * 4);
  d_a.ensure((size_t)n * 4);
  memcpy(h_a.p, ids, (size_t)n * 4);
  HIPCHK(hipMemcpyAsync(d_a.p, h_a.p, (size_t)n * 4, hipMemcpyHostToDevice, stream));
  sffk::launch_set_tree(stream, stree.as<int32_t>(), d_a.as<int32_t>(), n, tree);
  sync();
  if (grid_on) {   // the index's items carry their node's tree id: re-inserted with the new labels (merges are rare)
    double lim[6];
    memcpy(lim, grid_limits, sizeof lim);
    grid_setup(lim, grid_cell);
    grid_insert_new();
    grid_check(/*bulk=*/true);
  }
}

// ------------------------------------------------------------------ grid
void Ctx::grid_setup(const double limits[6], double cell) {
  HIPCHK(hipSetDevice(device));
  memcpy(grid_limits, limits, sizeof grid_limits);
  double ext[3] = {limits[1] - limits[0], limits[3] - limits[2], limits[5] - limits[4]};
  for (int tries = 0; tries < 64; ++tries) {
    double cells = 1;
    for (int a = 0; a < 3; ++a) cells *= std::floor(ext[a] / cell) + 1;
    if (cells <= 16777216.0) break;
    cell *= 1.26;
  }
  gridv = sffk::GridView{};
  gridv.ox = (float)limits[0];
  gridv.oy = (float)limits[2];
  gridv.oz = (float)limits[4];
  grid_cell = cell;
  gridv.inv_cell = (float)(1.0 / cell);
  gridv.nx = (int)std::floor(ext[0] / cell) + 1;
  gridv.ny = (int)std::floor(ext[1] / cell) + 1;
  gridv.nz = (int)std::floor(ext[2] / cell) + 1;
  gridv.bk = grid_bk;
  const size_t ncells = (size_t)gridv.nx * gridv.ny * gridv.nz;
  if (gridv_ovf_cap_next < 65536) gridv_ovf_cap_next = 65536;
  gridv.ovf_cap = gridv_ovf_cap_next;
  if (const char* e = getenv("SFFGPU_TEST_GRID_OVF")) gridv.ovf_cap = std::max(gridv.ovf_cap / 65536 * atoi(e), atoi(e));  // tests: tiny list
  g_cnt.ensure(ncells * sizeof(int32_t));
  g_items.ensure(ncells * gridv.bk * sizeof(sffk::GridItem));
  g_ovfcnt.ensure(16);
  g_ovf.ensure((size_t)gridv.ovf_cap * sizeof(sffk::GridItem));
  g_lite.ensure(ncells * gridv.bk * sizeof(sffk::GridItem32));
  g_ovf_lite.ensure((size_t)gridv.ovf_cap * sizeof(sffk::GridItem32));
  HIPCHK(hipMemsetAsync(g_cnt.p, 0, ncells * sizeof(int32_t), stream));
  HIPCHK(hipMemsetAsync(g_ovfcnt.p, 0, 16, stream));
  gridv.cnt = g_cnt.as<int32_t>();
  gridv.items = g_items.as<sffk::GridItem>();
  gridv.ovf_cnt = g_ovfcnt.as<int32_t>();
  gridv.ovf = g_ovf.as<sffk::GridItem>();
  gridv.lite = g_lite.as<sffk::GridItem32>();
  gridv.ovf_lite = g_ovf_lite.as<sffk::GridItem32>();
  // the round's own grid: same cells, its own buckets / overflow list, all counters zero between rounds
  tgridv = gridv;
  tgridv.bk = 8;
  tgridv.ovf_cap = std::max(1 << 20, tgrid_ovf_min);   // >= the samples of one round (Forest::Forest)
  t_cnt.ensure(ncells * sizeof(int32_t));
  t_items.ensure(ncells * tgridv.bk * sizeof(sffk::GridItem));
  t_ovfcnt.ensure(16);
  t_ovf.ensure((size_t)tgridv.ovf_cap * sizeof(sffk::GridItem));
  t_lite.ensure(ncells * tgridv.bk * sizeof(sffk::GridItem32));
  t_ovf_lite.ensure((size_t)tgridv.ovf_cap * sizeof(sffk::GridItem32));
  HIPCHK(hipMemsetAsync(t_cnt.p, 0, ncells * sizeof(int32_t), stream));
  HIPCHK(hipMemsetAsync(t_ovfcnt.p, 0, 16, stream));
  tgridv.cnt = t_cnt.as<int32_t>();
  tgridv.items = t_items.as<sffk::GridItem>();
  tgridv.ovf_cnt = t_ovfcnt.as<int32_t>();
  tgridv.ovf = t_ovf.as<sffk::GridItem>();
  tgridv.lite = t_lite.as<sffk::GridItem32>();
  tgridv.ovf_lite = t_ovf_lite.as<sffk::GridItem32>();
  t_occ.ensure((ncells / 32 + 2) * 4);
  HIPCHK(hipMemsetAsync(t_occ.p, 0, (ncells / 32 + 2) * 4, stream));
  tgridv.occ = t_occ.as<uint32_t>();
  grid_on = true;
  grid_inserted = 0;
}
void Ctx::grid_insert_new() {
  if (!grid_on || grid_inserted >= store_n) return;
  sffk::launch_grid_insert(stream, gridv, store_view(), grid_inserted, store_n - grid_inserted);
  grid_inserted = store_n;
}
// The shared overflow list is scanned by every query, so it must stay short.  When a quarter of it is in use
// (many nodes per xyz cell: dense forests, or forests that fill the angular dimensions) the grid is rebuilt
// with smaller cells — queries then visit more cells but shorter buckets — and all nodes are re-inserted.
void Ctx::grid_check(bool bulk) {
  if (!grid_on) return;
  int32_t v = 0;
  HIPCHK(hipMemcpyAsync(&v, g_ovfcnt.p, 4, hipMemcpyDeviceToHost, stream));
  HIPCHK(hipStreamSynchronize(stream));
  if (bulk) {
    // an index built over a store that already holds many nodes (sffgpu_nodes_index): nothing has been queried yet, so
    // a list that ran over only means "rebuild with smaller cells / deeper buckets" - until everything fits
    for (int tries = 0; tries < 40 && v > grid_rebuild_at(); ++tries) {
      const size_t cells_now = (size_t)gridv.nx * gridv.ny * gridv.nz;
      double cell = grid_cell;
      if (cells_now * 4 <= 16777216 && grid_cell * 0.63 >= 0.5 * grid_cell0) cell = grid_cell * 0.63;
      else if (grid_bk < grid_bk_max() && cells_now * (size_t)grid_bk * 2 * sizeof(sffk::GridItem) <= ((size_t)24 << 30)) grid_bk *= 2;
      else if (cells_now * 4 <= 16777216) cell = grid_cell * 0.63;
      else grid_grow_list();
      double lim[6];
      memcpy(lim, grid_limits, sizeof lim);
      grid_setup(lim, cell);
      grid_insert_new();
      ++grid_rebuilds;
      HIPCHK(hipMemcpyAsync(&v, g_ovfcnt.p, 4, hipMemcpyDeviceToHost, stream));
      HIPCHK(hipStreamSynchronize(stream));
    }
    if (v > gridv.ovf_cap) throw HipError{"grid overflow list exhausted"};
    return;
  }
  // entries beyond the capacity were dropped by grid_put / k_store_write: the queries of the wave that has just
  // finished may have missed nodes, so its results cannot be trusted (the capacity is sized so that one wave
  // cannot get here from below the rebuild threshold: ovf_cap - ovf_cap / 4 >= wave, see Forest::Forest)
  if (v > gridv.ovf_cap) throw HipError{"neighbour grid overflow list exhausted during a wave (nodes were dropped)"};
  if (v <= grid_rebuild_at()) return;
  const size_t cells_now = (size_t)gridv.nx * gridv.ny * gridv.nz;
  double cell = grid_cell;
  if (cells_now * 4 <= 16777216 && grid_cell * 0.63 >= 0.5 * grid_cell0) cell = grid_cell * 0.63;   // ~4x the cells
  else if (grid_bk < grid_bk_max() && cells_now * (size_t)grid_bk * 2 * sizeof(sffk::GridItem) <= ((size_t)24 << 30))
    grid_bk *= 2;                                           // cell count exhausted: deeper buckets (HBM is plentiful) ...
  else grid_grow_list();                                    // ... and only then a longer list
  double lim[6];
  memcpy(lim, grid_limits, sizeof lim);
  grid_setup(lim, cell);
  grid_insert_new();
  ++grid_rebuilds;
  HIPCHK(hipMemcpyAsync(&v, g_ovfcnt.p, 4, hipMemcpyDeviceToHost, stream));
  HIPCHK(hipStreamSynchronize(stream));
  if (v > gridv.ovf_cap) throw HipError{"grid overflow list exhausted"};
}

// cells and buckets cannot grow any further: the overflow list does (x 4, at most 2^26 entries = 4 GB of items), and from
// here on the re-cell trigger is a quarter of the list instead of 128 entries
void Ctx::grid_grow_list() {
  grid_exhausted = true;
  if (gridv.ovf_cap > (1 << 24)) throw HipError{"neighbour grid: more than 2^24 nodes do not fit their cells' buckets (too many nodes per xyz cell)"};
  gridv_ovf_cap_next = gridv.ovf_cap * 4;
}

// slack that makes the fp32 sweep filter a superset of the exact fp64 test: a few fp32 ulps of
// the largest coordinate magnitude in play
double Ctx::sweep_eps() const { return std::max(store_maxabs, env_maxabs) * std::ldexp(1.0, -20); }

// ------------------------------------------------------------------ batched primitives
void Ctx::collide_poses(const double* pos6, int n, uint8_t* hit, bool explicit_rt) {
  if (n <= 0) return;
  if (!have_env || !have_robot) throw HipError{"collide_poses: upload ENV and ROBOT meshes first"};
  HIPCHK(hipSetDevice(device));
  const size_t per = (explicit_rt ? 12 : 6) * sizeof(double);
  h_a.ensure((size_t)n * per);
  h_b.ensure((size_t)n);
  memcpy(h_a.p, pos6, (size_t)n * per);
  d_a.ensure((size_t)n * per);
  d_b.ensure((size_t)n);
  HIPCHK(hipMemcpyAsync(d_a.p, h_a.p, (size_t)n * per, hipMemcpyHostToDevice, stream));
  time_begin(T_COLLIDE);
  sffk::launch_collide_poses(stream, envv, robv, d_a.as<double>(), n, nullptr, d_b.as<uint8_t>(), explicit_rt);
  time_end();
  HIPCHK(hipMemcpyAsync(h_b.p, d_b.p, (size_t)n, hipMemcpyDeviceToHost, stream));
  sync();
  memcpy(hit, h_b.p, (size_t)n);
}

void Ctx::collide_segments(const double* a6, const double* b6, int n, uint8_t* is_free, int32_t* first_hit,
                           int32_t* n_samples) {
  collide_segments_core(a6, b6, nullptr, nullptr, n, is_free, first_hit, n_samples);
}
void Ctx::collide_segments_ids(const int32_t* ida, const int32_t* idb, int n, uint8_t* is_free, int32_t* first_hit,
                               int32_t* n_samples) {
  collide_segments_core(nullptr, nullptr, ida, idb, n, is_free, first_hit, n_samples);
}
void Ctx::collide_segments_refs(const int32_t* ida, const int32_t* idb, int n, uint8_t* is_free, int32_t* first_hit, int32_t* n_samples) {
  if (!rr_np_dev) throw HipError{"collide_segments_refs: no rrt_chain has run"};
  collide_segments_core(nullptr, nullptr, ida, idb, n, is_free, first_hit, n_samples, rr_np_dev);
}
void Ctx::collide_segments_core(const double* a6, const double* b6, const int32_t* ida, const int32_t* idb, int n,
                                uint8_t* is_free, int32_t* first_hit, int32_t* n_samples, const double* extra_dev) {
  if (n <= 0) return;
  if (!have_env || !have_robot) throw HipError{"collide_segments: upload ENV and ROBOT meshes first"};
  HIPCHK(hipSetDevice(device));
  // sample counts (src/problemStruct.h:155-156) and result presets are computed on the device, then the slot
  // table is compacted into (edge, chunk) work items for the persistent edge kernel
  const size_t pb = (size_t)n * 6 * sizeof(double);
  h_a.ensure(pb); h_b.ensure(pb); h_c.ensure((size_t)n * 12 + 64);
  if (a6) {
    memcpy(h_a.p, a6, pb);
    memcpy(h_b.p, b6, pb);
  } else {
    memcpy(h_a.p, ida, (size_t)n * 4);
    memcpy(h_b.p, idb, (size_t)n * 4);
  }
  d_a.ensure(pb); d_b.ensure(pb); d_c.ensure((size_t)n * 12 + 64);
  int32_t* d_ns = d_c.as<int32_t>();
  int32_t* d_fh = d_ns + n;
  int32_t* d_ov = d_fh + n;
  int32_t* d_ctrl = d_ov + n;
  if (a6) {
    HIPCHK(hipMemcpyAsync(d_a.p, h_a.p, pb, hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(d_b.p, h_b.p, pb, hipMemcpyHostToDevice, stream));
  } else {   // ids up, positions gathered from the store on the device
    d_d.ensure((size_t)n * 8);
    HIPCHK(hipMemcpyAsync(d_d.p, h_a.p, (size_t)n * 4, hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(d_d.as<int32_t>() + n, h_b.p, (size_t)n * 4, hipMemcpyHostToDevice, stream));
    sffk::launch_seg_gather(stream, spos.as<double>(), d_d.as<int32_t>(), d_d.as<int32_t>() + n, n, d_a.as<double>(),
                            d_b.as<double>(), extra_dev);
  }
  HIPCHK(hipMemsetAsync(d_ctrl, 0, 64, stream));
  const int list_cap = 8 * n + 65536;
  r_items.ensure((size_t)list_cap * SFFK_ITEM_BYTES);
  r_items2.ensure(((size_t)list_cap + (1u << 20)) * 8);   // (+ one window of the exact kernel, see mask_slot)
  time_begin(T_COLLIDE);
  sffk::launch_seg_prepare(stream, d_a.as<double>(), d_b.as<double>(), n, d_ns, d_fh, d_ov);
  sffk::launch_collide_segments_dyn(stream, envv, robv, d_a.as<double>(), d_b.as<double>(), d_ns, n, d_ctrl,
                                    r_items.p, list_cap, r_items2.p, d_fh, d_ov);
  time_end();
  HIPCHK(hipMemcpyAsync(h_c.p, d_c.p, (size_t)n * 12, hipMemcpyDeviceToHost, stream));
  sync();
  seg_finish(h_c.as<int32_t>(), n, a6, b6, d_a.p, d_b.p, is_free, first_hit, n_samples);
}

// the answer of an edge batch (n sample counts | n first hits or INT_MAX | n overflow marks) -> the caller's arrays; edges whose
// triangle candidate list ran over are re-run sample by sample through the pose kernel (dev_a / dev_b: the batch's end points on
// the device, fetched when the caller has none)
void Ctx::seg_finish(const int32_t* hn_in, int n, const double* a6, const double* b6, const void* dev_a, const void* dev_b, uint8_t* is_free,
                     int32_t* first_hit, int32_t* n_samples) {
  const size_t pb = (size_t)n * 6 * sizeof(double);
  // (one pass over the answer: a wave of the RRT* session brings tens of thousands of edges per call)
  bool any_ovf = false;
  {
    const int32_t* hn = hn_in;
    const int32_t* hf = hn + n;
    const int32_t* ho = hn + 2 * (size_t)n;
    for (int i = 0; i < n; ++i) {
      const int32_t v = hf[i] == 0x7fffffff ? -1 : hf[i];
      is_free[i] = v < 0 ? 1 : 0;
      if (first_hit) first_hit[i] = v;
      if (n_samples) n_samples[i] = hn[i];
      any_ovf |= ho[i] != 0;
    }
  }
  if (!any_ovf) return;
  // edges whose candidate list overflowed are re-run sample by sample through the pose kernel
  std::vector<int32_t> ns(n), fh(n);
  std::vector<uint8_t> ovf(n);
  {
    const int32_t* hn = hn_in;   // (still the answer: nothing has been enqueued on its buffer since)
    for (int i = 0; i < n; ++i) {
      ns[i] = hn[i];
      const int32_t v = hn[(size_t)n + i];
      fh[i] = v == 0x7fffffff ? -1 : v;
      ovf[i] = hn[2 * (size_t)n + i] != 0;
    }
  }
  std::vector<double> ga, gb;
  if (any_ovf && !a6) {   // (their end points only exist on the device: fetch the gathered arrays)
    ga.resize((size_t)n * 6);
    gb.resize((size_t)n * 6);
    HIPCHK(hipMemcpy(ga.data(), dev_a, pb, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(gb.data(), dev_b, pb, hipMemcpyDeviceToHost));
    a6 = ga.data();
    b6 = gb.data();
  }
  for (int i = 0; i < n && any_ovf; ++i) {
    if (!ovf[i]) continue;
    const double* a = a6 + 6 * (size_t)i;
    const double* b = b6 + 6 * (size_t)i;
    double parts = sffg::edge_parts(a, b);
    int cnt = ns[i];
    double dir[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
    std::vector<double> poses((size_t)cnt * 6, 0.0);
    for (int s = 1; s <= cnt; ++s) sffg::edge_sample_pos(a, dir, parts, s, &poses[6 * (size_t)(s - 1)]);
    std::vector<uint8_t> hits(cnt);
    collide_poses(poses.data(), cnt, hits.data());
    fh[i] = -1;
    for (int s = 0; s < cnt; ++s)
      if (hits[s]) { fh[i] = s + 1; break; }
  }
  for (int i = 0; i < n; ++i) {
    is_free[i] = fh[i] < 0 ? 1 : 0;
    if (first_hit) first_hit[i] = fh[i];
    if (n_samples) n_samples[i] = ns[i];
  }
}

// Several edge batches in flight (the RRT* wave's member edges in parts: the host builds the later parts' lists and replays the
// earlier parts' rows while the GPU checks the others).  seg_refs_begin enqueues and returns; seg_refs_end waits for that batch.
void Ctx::seg_refs_begin(int which, const int32_t* ida, const int32_t* idb, int n, int n_hint) {
  SegJob& J = seg_job[which];
  J.n = n;
  if (n <= 0) return;
  if (!rr_np_dev) throw HipError{"seg_refs_begin: no rrt_chain has run"};
  if (!have_env || !have_robot) throw HipError{"collide_segments: upload ENV and ROBOT meshes first"};
  HIPCHK(hipSetDevice(device));
  if (!J.ev) HIPCHK(hipEventCreateWithFlags(&J.ev, hipEventDisableTiming));
  const size_t pb = (size_t)n * 6 * sizeof(double);
  J.hids.ensure((size_t)n * 8);
  memcpy(J.hids.p, ida, (size_t)n * 4);
  memcpy(J.hids.as<int32_t>() + n, idb, (size_t)n * 4);
  J.ids.ensure((size_t)n * 8); J.a.ensure(pb); J.b.ensure(pb); J.c.ensure((size_t)n * 12 + 64); J.hc.ensure((size_t)n * 12);
  int32_t* d_ns = J.c.as<int32_t>();
  int32_t* d_fh = d_ns + n;
  int32_t* d_ov = d_fh + n;
  int32_t* d_ctrl = d_ov + n;
  HIPCHK(hipMemcpyAsync(J.ids.p, J.hids.p, (size_t)n * 8, hipMemcpyHostToDevice, stream));
  sffk::launch_seg_gather(stream, spos.as<double>(), J.ids.as<int32_t>(), J.ids.as<int32_t>() + n, n, J.a.as<double>(), J.b.as<double>(), rr_np_dev);
  HIPCHK(hipMemsetAsync(d_ctrl, 0, 64, stream));
  const int big = std::max(n, n_hint);   // (the work lists are shared: sized once for the larger batch, nothing is reallocated under a batch in flight)
  r_items.ensure((size_t)(8 * big + 65536) * SFFK_ITEM_BYTES);
  r_items2.ensure(((size_t)(8 * big + 65536) + (1u << 20)) * 8);
  const int list_cap = 8 * n + 65536;
  time_begin(T_COLLIDE);
  sffk::launch_seg_prepare(stream, J.a.as<double>(), J.b.as<double>(), n, d_ns, d_fh, d_ov);
  sffk::launch_collide_segments_dyn(stream, envv, robv, J.a.as<double>(), J.b.as<double>(), d_ns, n, d_ctrl, r_items.p, list_cap, r_items2.p, d_fh, d_ov);
  time_end();
  HIPCHK(hipMemcpyAsync(J.hc.p, J.c.p, (size_t)n * 12, hipMemcpyDeviceToHost, stream));
  HIPCHK(hipEventRecord(J.ev, stream));
}
void Ctx::seg_refs_end(int which, uint8_t* is_free, int32_t* first_hit, int32_t* n_samples) {
  SegJob& J = seg_job[which];
  if (J.n <= 0) return;
  HIPCHK(hipEventSynchronize(J.ev));
  seg_finish(J.hc.as<int32_t>(), J.n, nullptr, nullptr, J.a.p, J.b.p, is_free, first_hit, n_samples);
  J.n = 0;
}

// RRT session (csrc/rrt.cpp): the GPU half of one speculative wave as ONE enqueued chain and one wait.
// rrt_chain: the two nearest nodes of every steering target -> the steered new point -> per row (rr_enqueue) its pose, its
// parent edge, its k nearest nodes, the other trees' nodes around it -> which earlier new point of the wave would be nearer
// than the nearest node (k_rrt_mates) -> the slots that have one, in slot order (k_rrt_alt_list) -> the same rows for those
// REPAIRED slots, steered from that earlier new point instead (alt_cap rows; the unused ones are degenerate copies of row 0).
// rrt_chain_alt: the repaired rows as a second chain from a host-built list (SFFGPU_RRT_ONE_CHAIN=0).
// The new points of both stay in rr_np (rows 0..n-1, then the repaired ones): the wave's later edges name them by row.
namespace {
struct RrLayout {   // one result block, 8-byte parts first
  size_t o_np, o_md, o_cd, o_nd, o_sg, o_mi, o_mc, o_ci, o_cc, o_ni, o_nc, o_mt, o_al, o_ht, o_end, km, kc;
};
RrLayout rr_layout(int n, int kmax, bool conn, int conn_cap, int n_near, int n_alt_list) {
  RrLayout L{};
  const size_t K1 = 2;
  L.km = (size_t)std::max(kmax, 0);
  L.kc = conn ? (size_t)conn_cap : 0;
  size_t o = 0;
  L.o_np = o; o += (size_t)n * 48;
  L.o_md = o; o += (size_t)n * L.km * 8;
  L.o_cd = o; o += (size_t)n * L.kc * 8;
  L.o_nd = o; o += (size_t)n_near * K1 * 8;
  L.o_sg = o; o += ((size_t)n * 3 + 16) * 4;
  L.o_mi = o; o += (size_t)n * L.km * 4;
  L.o_mc = o; o += (size_t)n * 4;
  L.o_ci = o; o += (size_t)n * L.kc * 4;
  L.o_cc = o; o += (L.kc ? (size_t)n : 0) * 4;
  L.o_ni = o; o += (size_t)n_near * K1 * 4;
  L.o_nc = o; o += (size_t)n_near * 4;
  L.o_mt = o; o += (size_t)n_near * 4;
  L.o_al = o; o += (n_alt_list ? 2 * (size_t)n_alt_list + 2 : 0) * 4;   // slot[cap] | mate[cap] | listed, found
  L.o_ht = o; o += ((size_t)n + 7) / 8 * 8;
  L.o_end = o;
  return L;
}
float rrt_conn_r2f(double conn_r, double eps) {   // (the superset radius of Ctx::sweep_lists)
  const double ri = (conn_r + eps) * (1.0 + 1e-5);
  return (float)(ri * ri) * 1.000001f;
}
}  // namespace

// pose, parent edge, k nearest, other trees, (mates) of rows row0 .. row0 + n - 1 of rr_np into the block `db`; k_rrt_steer has
// written the rows' new points, nearest positions (rr_a), queries (rr_q2 / rr_sq) and the edge presets before
// (parts: 1 = pose + parent edge (+ mates) on the context's stream, 2 = the two queries - they only need the new points - on
// stream `qs` from the query records q2 / sq; 3 = both)
static void rr_enqueue(Ctx& c, char* db, const RrLayout& L, int row0, int n, int kmax, bool by_gridk, int conn_cap, bool mates, int parts = 3,
                       hipStream_t qs = nullptr, const sffk::KnnQuery* q2 = nullptr, const sffk::SweepQuery* sq = nullptr) {
  if (!qs) qs = c.stream;
  if (!q2) q2 = c.rr_q2.as<sffk::KnnQuery>();
  if (!sq) sq = c.rr_sq.as<sffk::SweepQuery>();
  const bool timed = qs == c.stream;
  double* r_np = c.rr_np.as<double>() + 6 * (size_t)row0;
  int32_t* d_ns = reinterpret_cast<int32_t*>(db + L.o_sg);
  int32_t* d_fh = d_ns + n;
  int32_t* d_ov = d_fh + n;
  int32_t* d_ctrl = d_ov + n;
  uint8_t* d_hit = reinterpret_cast<uint8_t*>(db + L.o_ht);
  const int list_cap = 8 * n + 65536;
  if (parts & 1) {
  c.time_begin(T_COLLIDE);
  // (the poses ride the edge kernels: the cull pass marks the ones that need the exact test, the exact kernel takes them first)
  sffk::launch_round_collide(c.stream, c.envv, c.robv, r_np, n, nullptr, d_hit, c.rr_a.as<double>(), r_np, d_ns, n, d_ctrl, c.r_items.p, list_cap,
                             c.r_items2.p, d_fh, d_ov, nullptr);
  c.time_end();
  if (mates) sffk::launch_rrt_mates(c.stream, c.rr_q1.as<sffk::KnnQuery>(), reinterpret_cast<double*>(db + L.o_nd), 2, r_np, d_hit, d_fh, d_ov, n,
                                    reinterpret_cast<int32_t*>(db + L.o_mt));
  }
  if (!(parts & 2)) return;
  if (kmax > 0) {
    if (timed) c.time_begin(T_SWEEP);
    if (by_gridk && c.grid_on && c.store_n >= kmax)
      sffk::launch_knn_grid(qs, c.gridv, nullptr, c.store_view(), q2, n, kmax, reinterpret_cast<int32_t*>(db + L.o_mi),
                            reinterpret_cast<double*>(db + L.o_md), reinterpret_cast<int32_t*>(db + L.o_mc), nullptr, nullptr, c.grid_cell,
                            8 * c.sweep_eps(), SFFK_KNN_MATES, c.store_n);
    else
      sffk::launch_knn_linear(qs, c.store_view(), c.store_n, q2, n, kmax, reinterpret_cast<int32_t*>(db + L.o_mi),
                              reinterpret_cast<double*>(db + L.o_md), reinterpret_cast<int32_t*>(db + L.o_mc), c.sweep_eps());
    if (timed) c.time_end();
  }
  if (L.kc) {
    if (timed) c.time_begin(T_SWEEP);
    sffk::launch_sweep(qs, c.store_view(), 0, c.store_n, sq, r_np, n, reinterpret_cast<int32_t*>(db + L.o_cc),
                       reinterpret_cast<int32_t*>(db + L.o_ci), reinterpret_cast<double*>(db + L.o_cd), conn_cap);
    if (timed) c.time_end();
  }
}

static void rr_unpack(const char* hb, const RrLayout& L, Ctx::RrtRows& R, int n, int kmax) {
  memcpy(R.np6, hb + L.o_np, (size_t)n * 48);
  memcpy(R.hit, hb + L.o_ht, (size_t)n);
  memcpy(R.seg, hb + L.o_sg, (size_t)n * 12);
  if (kmax > 0) {
    memcpy(R.mem_idx, hb + L.o_mi, (size_t)n * L.km * 4);
    memcpy(R.mem_d, hb + L.o_md, (size_t)n * L.km * 8);
    memcpy(R.mem_cnt, hb + L.o_mc, (size_t)n * 4);
  }
  if (L.kc) {
    memcpy(R.conn_idx, hb + L.o_ci, (size_t)n * L.kc * 4);
    memcpy(R.conn_d, hb + L.o_cd, (size_t)n * L.kc * 8);
    memcpy(R.conn_cnt, hb + L.o_cc, (size_t)n * 4);
  }
}

void Ctx::rrt_chain(const double* rnd6, const int32_t* tree, int n, double dist, bool by_grid1, int kmax, bool by_gridk,
                    int32_t* near_idx, double* near_d, int32_t* near_cnt, int32_t* mate, RrtRows& R, double conn_r, int conn_cap,
                    int alt_cap, int32_t* alt_slot, int32_t* alt_mate, int32_t* n_alt, RrtRows* R2) {
  if (n <= 0) return;
  if (!have_env || !have_robot) throw HipError{"rrt_chain: upload ENV and ROBOT meshes first"};
  HIPCHK(hipSetDevice(device));
  const int K1 = 2;
  if (alt_cap > n) alt_cap = n;
  if (!mate || !R2) alt_cap = 0;
  rr_hq.ensure((size_t)n * sizeof(sffk::KnnQuery));
  sffk::KnnQuery* hq = rr_hq.as<sffk::KnnQuery>();
  for (int i = 0; i < n; ++i) {
    memcpy(hq[i].pos, rnd6 + 6 * (size_t)i, sizeof hq[i].pos);
    hq[i].tree = tree ? tree[i] : -1;
    hq[i].max_id = std::numeric_limits<int32_t>::max();
    hq[i].k = K1;
    hq[i].mate_base = std::numeric_limits<int32_t>::max();
    hq[i].whole_tree = 0;
    hq[i].pad_ = 0;
  }
  const bool kc = conn_r > 0;
  const RrLayout L1 = rr_layout(n, kmax, kc, conn_cap, n, alt_cap), L2 = rr_layout(alt_cap, kmax, kc, conn_cap, 0, 0);
  rr_q1.ensure((size_t)n * sizeof(sffk::KnnQuery));
  rr_q2.ensure((size_t)n * sizeof(sffk::KnnQuery));
  rr_a.ensure((size_t)n * 48);
  rr_np.ensure((size_t)n * 2 * 48);   // (+ the repaired rows)
  if (kc) rr_sq.ensure((size_t)n * sizeof(sffk::SweepQuery));
  const size_t o_b2 = (L1.o_end + 63) / 64 * 64;   // (the repaired rows' block behind the slots': one copy back)
  rr_out.ensure(o_b2 + (alt_cap ? L2.o_end : 0));
  rr_hout.ensure(o_b2 + (alt_cap ? L2.o_end : 0));
  r_items.ensure((size_t)(8 * n + 65536) * SFFK_ITEM_BYTES);
  r_items2.ensure(((size_t)(8 * n + 65536) + (1u << 20)) * 8);
  char* db = rr_out.as<char>();
  int32_t* r_ni = reinterpret_cast<int32_t*>(db + L1.o_ni);
  double* r_nd = reinterpret_cast<double*>(db + L1.o_nd);
  int32_t* r_nc = reinterpret_cast<int32_t*>(db + L1.o_nc);
  HIPCHK(hipMemcpyAsync(rr_q1.p, rr_hq.p, (size_t)n * sizeof(sffk::KnnQuery), hipMemcpyHostToDevice, stream));
  if ((by_grid1 || (kmax > 0 && by_gridk)) && grid_on) {
    grid_insert_new();
    grid_check(/*bulk=*/true);
  }
  time_begin(T_SWEEP);
  if (by_grid1 && grid_on && store_n >= K1)
    sffk::launch_knn_grid(stream, gridv, nullptr, store_view(), rr_q1.as<sffk::KnnQuery>(), n, K1, r_ni, r_nd, r_nc, nullptr, nullptr, grid_cell,
                          8 * sweep_eps(), SFFK_KNN_MATES, store_n);
  else
    sffk::launch_knn_linear(stream, store_view(), store_n, rr_q1.as<sffk::KnnQuery>(), n, K1, r_ni, r_nd, r_nc, sweep_eps());
  time_end();
  const float r2f = kc ? rrt_conn_r2f(conn_r, sweep_eps()) : 0.f;
  sffk::launch_rrt_steer(stream, rr_q1.as<sffk::KnnQuery>(), r_ni, K1, spos.as<double>(), dist, rr_a.as<double>(), rr_np.as<double>(),
                         kmax > 0 ? rr_q2.as<sffk::KnnQuery>() : nullptr, kmax, n, kc ? rr_sq.as<sffk::SweepQuery>() : nullptr, conn_r, r2f,
                         reinterpret_cast<double*>(db + L1.o_np), reinterpret_cast<int32_t*>(db + L1.o_sg),
                         kc ? reinterpret_cast<int32_t*>(db + L1.o_cc) : nullptr);
  // the k nearest and the other trees' nodes of the new points need nothing of the pose / edge answers: they run on a second
  // stream beside them (RRT*: the k-nearest kernel is the longest of the chain)
  const bool fork = rr_fork && (kmax > 0 || kc) && copy_stream;
  if (fork) {
    if (!rr_ev[0]) for (hipEvent_t& e : rr_ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIPCHK(hipEventRecord(rr_ev[0], stream));
    HIPCHK(hipStreamWaitEvent(copy_stream, rr_ev[0], 0));
    rr_enqueue(*this, db, L1, 0, n, kmax, by_gridk, conn_cap, false, 2, copy_stream);
  }
  rr_enqueue(*this, db, L1, 0, n, kmax, by_gridk, conn_cap, mate != nullptr, fork ? 1 : 3);
  if (alt_cap) {
    // the repaired rows behind them, without a word from the host: the list of the slots that have a mate, the rows steered
    // from it (rr_a, rr_q2, rr_sq are free again: the stream keeps the order)
    char* db2 = db + o_b2;
    int32_t* d_al = reinterpret_cast<int32_t*>(db + L1.o_al);
    sffk::launch_rrt_alt_list(stream, reinterpret_cast<int32_t*>(db + L1.o_mt), n, alt_cap, d_al, d_al + alt_cap, d_al + 2 * alt_cap);
    // (their query records: arrays of their own - the slots' may still be read on the second stream)
    if (kmax > 0) rr_q2b.ensure((size_t)alt_cap * sizeof(sffk::KnnQuery));
    if (kc) rr_sqb.ensure((size_t)alt_cap * sizeof(sffk::SweepQuery));
    sffk::launch_rrt_steer(stream, rr_q1.as<sffk::KnnQuery>(), nullptr, 0, spos.as<double>(), dist, rr_a.as<double>(), rr_np.as<double>(),
                           kmax > 0 ? rr_q2b.as<sffk::KnnQuery>() : nullptr, kmax, alt_cap, kc ? rr_sqb.as<sffk::SweepQuery>() : nullptr, conn_r, r2f,
                           reinterpret_cast<double*>(db2 + L2.o_np), reinterpret_cast<int32_t*>(db2 + L2.o_sg),
                           kc ? reinterpret_cast<int32_t*>(db2 + L2.o_cc) : nullptr, d_al, d_al + alt_cap, n);
    if (fork) {
      HIPCHK(hipEventRecord(rr_ev[1], stream));
      HIPCHK(hipStreamWaitEvent(copy_stream, rr_ev[1], 0));
      rr_enqueue(*this, db2, L2, n, alt_cap, kmax, by_gridk, conn_cap, false, 2, copy_stream, rr_q2b.as<sffk::KnnQuery>(), rr_sqb.as<sffk::SweepQuery>());
    }
    rr_enqueue(*this, db2, L2, n, alt_cap, kmax, by_gridk, conn_cap, false, fork ? 1 : 3, nullptr, rr_q2b.as<sffk::KnnQuery>(), rr_sqb.as<sffk::SweepQuery>());
  }
  if (fork) {
    HIPCHK(hipEventRecord(rr_ev[2], copy_stream));
    HIPCHK(hipStreamWaitEvent(stream, rr_ev[2], 0));
  }
  HIPCHK(hipMemcpyAsync(rr_hout.p, rr_out.p, o_b2 + (alt_cap ? L2.o_end : 0), hipMemcpyDeviceToHost, stream));
  sync();
  rr_np_dev = rr_np.as<double>();
  rr_rows0 = n;
  const char* hb = rr_hout.as<char>();
  memcpy(near_idx, hb + L1.o_ni, (size_t)n * K1 * 4);
  memcpy(near_d, hb + L1.o_nd, (size_t)n * K1 * 8);
  memcpy(near_cnt, hb + L1.o_nc, (size_t)n * 4);
  if (mate) memcpy(mate, hb + L1.o_mt, (size_t)n * 4);
  rr_unpack(hb, L1, R, n, kmax);
  if (n_alt) *n_alt = 0;
  if (alt_cap) {
    const int32_t* al = reinterpret_cast<const int32_t*>(hb + L1.o_al);
    const int listed = al[2 * (size_t)alt_cap];
    memcpy(alt_slot, al, (size_t)listed * 4);
    memcpy(alt_mate, al + alt_cap, (size_t)listed * 4);
    *n_alt = listed;
    rr_unpack(hb + o_b2, L2, *R2, alt_cap, kmax);   // (all alt_cap rows: the caller's arrays have that many, seg's three columns that stride)
  }
}

void Ctx::rrt_chain_alt(const int32_t* slot, const int32_t* mate, int n_alt, double dist, int kmax, bool by_gridk, RrtRows& R,
                        double conn_r, int conn_cap) {
  if (n_alt <= 0) return;
  if (!rr_np_dev || n_alt > rr_rows0) throw HipError{"rrt_chain_alt: no rrt_chain has run (or more repaired slots than slots)"};
  HIPCHK(hipSetDevice(device));
  const bool kc = conn_r > 0;
  const RrLayout L = rr_layout(n_alt, kmax, kc, conn_cap, 0, 0);
  rr_hq.ensure((size_t)n_alt * 8);
  memcpy(rr_hq.as<int32_t>(), slot, (size_t)n_alt * 4);
  memcpy(rr_hq.as<int32_t>() + n_alt, mate, (size_t)n_alt * 4);
  rr_alt.ensure((size_t)n_alt * 8);
  rr_out.ensure(L.o_end);
  rr_hout.ensure(L.o_end);
  r_items.ensure((size_t)(8 * n_alt + 65536) * SFFK_ITEM_BYTES);
  r_items2.ensure(((size_t)(8 * n_alt + 65536) + (1u << 20)) * 8);
  char* db = rr_out.as<char>();
  HIPCHK(hipMemcpyAsync(rr_alt.p, rr_hq.p, (size_t)n_alt * 8, hipMemcpyHostToDevice, stream));
  sffk::launch_rrt_steer(stream, rr_q1.as<sffk::KnnQuery>(), nullptr, 0, spos.as<double>(), dist, rr_a.as<double>(), rr_np.as<double>(),
                         kmax > 0 ? rr_q2.as<sffk::KnnQuery>() : nullptr, kmax, n_alt, kc ? rr_sq.as<sffk::SweepQuery>() : nullptr, conn_r,
                         kc ? rrt_conn_r2f(conn_r, sweep_eps()) : 0.f, reinterpret_cast<double*>(db + L.o_np),
                         reinterpret_cast<int32_t*>(db + L.o_sg), kc ? reinterpret_cast<int32_t*>(db + L.o_cc) : nullptr, rr_alt.as<int32_t>(),
                         rr_alt.as<int32_t>() + n_alt, rr_rows0);
  rr_enqueue(*this, db, L, rr_rows0, n_alt, kmax, by_gridk, conn_cap, false);
  HIPCHK(hipMemcpyAsync(rr_hout.p, rr_out.p, L.o_end, hipMemcpyDeviceToHost, stream));
  sync();
  rr_unpack(rr_hout.as<char>(), L, R, n_alt, kmax);
}

void Ctx::sample_steer(const uint64_t* words, const double* center6, int n, double dist, int dim, const double* limits,
                       double* out6, uint8_t* in_limits) {
  if (n <= 0) return;
  HIPCHK(hipSetDevice(device));
  const size_t wb = (size_t)n * 6 * sizeof(uint64_t), pb = (size_t)n * 6 * sizeof(double);
  h_a.ensure(wb); h_b.ensure(pb); h_c.ensure(pb); h_d.ensure((size_t)n);
  memcpy(h_a.p, words, wb);
  memcpy(h_b.p, center6, pb);
  d_a.ensure(wb); d_b.ensure(pb); d_c.ensure(pb); d_d.ensure((size_t)n);
  HIPCHK(hipMemcpyAsync(d_a.p, h_a.p, wb, hipMemcpyHostToDevice, stream));
  HIPCHK(hipMemcpyAsync(d_b.p, h_b.p, pb, hipMemcpyHostToDevice, stream));
  sffk::SampleParams prm{};
  memcpy(prm.limits, limits, sizeof prm.limits);
  prm.rank = 0;
  prm.world = 1;
  time_begin(T_SAMPLE);
  sffk::launch_sample_steer(stream, d_a.as<uint64_t>(), nullptr, nullptr, d_b.as<double>(), n, dist, dim, prm,
                            d_c.as<double>(), d_d.as<uint8_t>(), nullptr, nullptr, 0, sffk::RoundTemps{});
  time_end();
  HIPCHK(hipMemcpyAsync(h_c.p, d_c.p, pb, hipMemcpyDeviceToHost, stream));
  HIPCHK(hipMemcpyAsync(h_d.p, d_d.p, (size_t)n, hipMemcpyDeviceToHost, stream));
  sync();
  memcpy(out6, h_c.p, pb);
  memcpy(in_limits, h_d.p, (size_t)n);
}

// One sweep launch + host-side ordering.  Returns the raw per-query totals.
double g_sweep_dbg[4] = {0, 0, 0, 0};   // sweep_lists: enqueue, wait, unpack ms; queries
void Ctx::sweep_lists(const double* q6, int nq, const std::vector<double>& r, const int32_t* tree,
                      const int32_t* max_id, const std::vector<uint8_t>& active, int cap, int n_store,
                      std::vector<int32_t>& cnt, std::vector<std::vector<HitRec>>& out, bool sort_lists) {
  Ctx& c = *this;
  auto tq0 = std::chrono::steady_clock::now();
  const double eps = c.sweep_eps();
  // only the active queries travel (the later passes of an adaptive-radius search hold a handful of them)
  std::vector<int> act;
  act.reserve(nq);
  for (int i = 0; i < nq; ++i)
    if (active[i]) act.push_back(i);
  const int na = (int)act.size();
  cnt.assign(nq, 0);
  out.resize(nq);   // (the lists keep their capacity from call to call)
  for (int i = 0; i < nq; ++i) out[i].clear();
  if (na == 0) return;
  c.h_a.ensure((size_t)na * sizeof(sffk::SweepQuery));
  c.h_b.ensure((size_t)na * 6 * sizeof(double));
  sffk::SweepQuery* hq = c.h_a.as<sffk::SweepQuery>();
  double* hp = c.h_b.as<double>();
  for (int j = 0; j < na; ++j) {
    const int i = act[j];
    const double* p = q6 + 6 * (size_t)i;
    sffk::SweepQuery q{};
    q.x = (float)p[0]; q.y = (float)p[1]; q.z = (float)p[2];
    q.yaw = (float)p[3]; q.pitch = (float)p[4]; q.roll = (float)p[5];
    q.r = r[i];
    double ri = (r[i] + eps) * (1.0 + 1e-5);
    double r2 = ri * ri;
    q.r2f = r2 > 1e37 ? 3.0e38f : (float)r2 * 1.000001f;
    q.tree = tree ? tree[i] : -1;
    q.max_id = max_id ? max_id[i] : std::numeric_limits<int32_t>::max();
    q.active = 1;
    hq[j] = q;
    memcpy(hp + 6 * (size_t)j, p, 6 * sizeof(double));
  }
  c.d_a.ensure((size_t)na * sizeof(sffk::SweepQuery));
  c.d_b.ensure((size_t)na * 6 * sizeof(double));
  c.d_c.ensure((size_t)na * sizeof(int32_t));
  c.d_d.ensure((size_t)na * cap * sizeof(int32_t));
  c.d_e.ensure((size_t)na * cap * sizeof(double));
  HIPCHK(hipMemcpyAsync(c.d_a.p, c.h_a.p, (size_t)na * sizeof(sffk::SweepQuery), hipMemcpyHostToDevice, c.stream));
  HIPCHK(hipMemcpyAsync(c.d_b.p, c.h_b.p, (size_t)na * 6 * sizeof(double), hipMemcpyHostToDevice, c.stream));
  HIPCHK(hipMemsetAsync(c.d_c.p, 0, (size_t)na * sizeof(int32_t), c.stream));
  c.time_begin(T_SWEEP);
  sffk::launch_sweep(c.stream, c.store_view(), 0, n_store, c.d_a.as<sffk::SweepQuery>(), c.d_b.as<double>(), na,
                     c.d_c.as<int32_t>(), c.d_d.as<int32_t>(), c.d_e.as<double>(), cap);
  c.time_end();
  c.h_c.ensure((size_t)na * sizeof(int32_t));
  c.h_d.ensure((size_t)na * cap * sizeof(int32_t));
  c.h_e.ensure((size_t)na * cap * sizeof(double));
  HIPCHK(hipMemcpyAsync(c.h_c.p, c.d_c.p, (size_t)na * sizeof(int32_t), hipMemcpyDeviceToHost, c.stream));
  HIPCHK(hipMemcpyAsync(c.h_d.p, c.d_d.p, (size_t)na * cap * sizeof(int32_t), hipMemcpyDeviceToHost, c.stream));
  HIPCHK(hipMemcpyAsync(c.h_e.p, c.d_e.p, (size_t)na * cap * sizeof(double), hipMemcpyDeviceToHost, c.stream));
  auto tq1 = std::chrono::steady_clock::now();
  c.sync();
  auto tq2 = std::chrono::steady_clock::now();
  const int32_t* hc = c.h_c.as<int32_t>();
  const double* hd = c.h_e.as<double>();
  const int32_t* hi = c.h_d.as<int32_t>();
  for (int j = 0; j < na; ++j) {
    const int i = act[j];
    cnt[i] = hc[j];
    const int m = std::min(cnt[i], cap);
    out[i].resize(m);
    for (int k = 0; k < m; ++k) out[i][k] = {hd[(size_t)j * cap + k], hi[(size_t)j * cap + k]};
    if (sort_lists) std::sort(out[i].begin(), out[i].end());
  }
  auto tq3 = std::chrono::steady_clock::now();
  g_sweep_dbg[0] += std::chrono::duration<double, std::milli>(tq1 - tq0).count();
  g_sweep_dbg[1] += std::chrono::duration<double, std::milli>(tq2 - tq1).count();
  g_sweep_dbg[2] += std::chrono::duration<double, std::milli>(tq3 - tq2).count();
  g_sweep_dbg[3] += na;
}

void Ctx::radius(const double* q6, int nq, const double* r, const int32_t* tree, const int32_t* max_id, int32_t* idx,
                 double* dist, int32_t* cnt, int cap) {
  if (nq <= 0) return;
  HIPCHK(hipSetDevice(device));
  std::vector<double> rv(r, r + nq);
  std::vector<uint8_t> active(nq, 1);
  std::vector<int32_t> c;
  std::vector<std::vector<HitRec>> out;
  sweep_lists(q6, nq, rv, tree, max_id, active, cap, store_n, c, out);
  for (int i = 0; i < nq; ++i) {
    cnt[i] = c[i];
    for (size_t k = 0; k < out[i].size(); ++k) {
      idx[(size_t)i * cap + k] = out[i][k].id;
      if (dist) dist[(size_t)i * cap + k] = out[i][k].d;
    }
  }
}

// k nearest through radius sweeps: the radius of each query is adapted (grow while fewer than k
// are inside, shrink when the hit list overflows) until the k smallest exact distances are known.
void Ctx::knn(const double* q6, int nq, int k, const int32_t* tree, const int32_t* max_id, int32_t* idx, double* dist,
              int32_t* cnt, bool tree_by_grid) {
  if (nq <= 0 || k <= 0) return;
  HIPCHK(hipSetDevice(device));
  if (k <= 64) {
    // device top-k: one wavefront per query keeps its k best in registers while it streams the store (k_knn_linear)
    h_a.ensure((size_t)nq * sizeof(sffk::KnnQuery));
    sffk::KnnQuery* hq = h_a.as<sffk::KnnQuery>();
    for (int i = 0; i < nq; ++i) {
      memcpy(hq[i].pos, q6 + 6 * (size_t)i, sizeof hq[i].pos);
      hq[i].tree = tree ? tree[i] : -1;
      hq[i].max_id = max_id ? max_id[i] : std::numeric_limits<int32_t>::max();
      hq[i].k = k;
      hq[i].mate_base = std::numeric_limits<int32_t>::max();
      hq[i].whole_tree = 0;
      hq[i].pad_ = 0;
    }
    // one result block (distances | indices | counts), so that the answer comes back in ONE copy
    const size_t o_dist = 0, o_idx = o_dist + (size_t)nq * k * 8, o_cnt = o_idx + (size_t)nq * k * 4, o_end = o_cnt + (size_t)nq * 4;
    d_a.ensure((size_t)nq * sizeof(sffk::KnnQuery));
    d_b.ensure(o_end);
    HIPCHK(hipMemcpyAsync(d_a.p, h_a.p, (size_t)nq * sizeof(sffk::KnnQuery), hipMemcpyHostToDevice, stream));
    // with an index over the store (sffgpu_nodes_index) and no per-tree restriction (a tree with fewer than k nodes
    // would make the shells grow over the whole grid) every query is answered from the cells around it
    const bool by_grid = grid_on && (tree == nullptr || tree_by_grid) && store_n >= k;
    if (by_grid) {
      grid_insert_new();
      grid_check(/*bulk=*/true);
    }
    char* db = d_b.as<char>();
    int32_t* r_idx = reinterpret_cast<int32_t*>(db + o_idx);
    double* r_dist = reinterpret_cast<double*>(db + o_dist);
    int32_t* r_cnt = reinterpret_cast<int32_t*>(db + o_cnt);
    time_begin(T_SWEEP);
    if (by_grid)
      sffk::launch_knn_grid(stream, gridv, nullptr, store_view(), d_a.as<sffk::KnnQuery>(), nq, k, r_idx, r_dist, r_cnt, nullptr, nullptr,
                            grid_cell, 8 * sweep_eps(), SFFK_KNN_MATES, store_n);
    else
      sffk::launch_knn_linear(stream, store_view(), store_n, d_a.as<sffk::KnnQuery>(), nq, k, r_idx, r_dist, r_cnt, sweep_eps());
    time_end();
    h_b.ensure(o_end);
    HIPCHK(hipMemcpyAsync(h_b.p, d_b.p, o_end, hipMemcpyDeviceToHost, stream));
    sync();
    const char* hb = h_b.as<char>();
    const int32_t* g_idx = reinterpret_cast<const int32_t*>(hb + o_idx);
    const double* g_dist = reinterpret_cast<const double*>(hb + o_dist);
    const int32_t* g_cnt = reinterpret_cast<const int32_t*>(hb + o_cnt);
    for (int i = 0; i < nq; ++i) {
      const int m = g_cnt[i];
      cnt[i] = m;
      for (int j = 0; j < m; ++j) {
        idx[(size_t)i * k + j] = g_idx[(size_t)i * k + j];
        if (dist) dist[(size_t)i * k + j] = g_dist[(size_t)i * k + j];
      }
    }
    return;
  }
  // k > 64: adaptive-radius sweeps (grow while fewer than k are inside, shrink when the hit list overflows)
  const int cap = std::max(4 * k, 256);
  // initial guess: radius of a ball expected to hold ~2k nodes at uniform density
  double ext = std::max(store_maxabs, 1.0) * 2.0;
  double r0 = ext * std::cbrt(2.0 * k / std::max(1, store_n)) + 1e-6;
  std::vector<double> r(nq, r0), lo(nq, 0.0), hi(nq, -1.0);
  std::vector<uint8_t> active(nq, 1);
  std::vector<std::vector<HitRec>> best(nq);
  std::vector<int32_t> found(nq, 0);
  const double RMAX = 1e30;
  for (int it = 0; it < 200; ++it) {
    bool any = false;
    for (int i = 0; i < nq; ++i) any |= active[i] != 0;
    if (!any) break;
    std::vector<int32_t> c;
    std::vector<std::vector<HitRec>> out;
    sweep_lists(q6, nq, r, tree, max_id, active, cap, store_n, c, out);
    for (int i = 0; i < nq; ++i) {
      if (!active[i]) continue;
      if (c[i] > cap) {            // too many: shrink
        hi[i] = r[i];
        r[i] = 0.5 * (lo[i] + hi[i]);
      } else if (c[i] >= k || r[i] >= RMAX) {
        best[i] = out[i];
        found[i] = std::min(c[i], k);
        active[i] = 0;
      } else {                     // too few: grow
        lo[i] = r[i];
        r[i] = hi[i] > 0 ? 0.5 * (lo[i] + hi[i]) : std::min(RMAX, r[i] * 2.0);
      }
    }
  }
  for (int i = 0; i < nq; ++i)
    if (active[i]) throw HipError{"knn: the radius search did not converge (more than 4k nodes at one distance?)"};
  for (int i = 0; i < nq; ++i) {
    cnt[i] = found[i];
    for (int j = 0; j < found[i]; ++j) {
      idx[(size_t)i * k + j] = best[i][j].id;
      if (dist) dist[(size_t)i * k + j] = best[i][j].d;
    }
  }
}

}  // namespace sff
