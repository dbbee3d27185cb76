// capi.cpp — extern "C" surface of libsffgpu.so (declared in include/sffgpu.h).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>

#include "engine.h"

namespace sff { void forest_profile_dump(); }
using namespace sff;

struct sffgpu_ctx {
  Ctx* c;
};
struct sffgpu_rrt {
  Rrt* r;
  sffgpu_ctx* owner;
};
struct sffgpu_forest {
  Forest* f;
  sffgpu_ctx* owner;
};

static std::string g_create_err;

#define GUARD(ctxp, body)                          \
  try {                                            \
    body;                                          \
    return SFFGPU_OK;                              \
  } catch (const HipError& e) {                    \
    (ctxp)->c->err = e.msg;                        \
    return SFFGPU_ERR_HIP;                         \
  } catch (const std::exception& e) {              \
    (ctxp)->c->err = e.what();                     \
    return SFFGPU_ERR_STATE;                       \
  }

extern "C" {

const char* sffgpu_version(void) { return "sffgpu 0.1 (gfx950)"; }

int sffgpu_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int sffgpu_create(int device, sffgpu_ctx** out) {
  if (!out) return SFFGPU_ERR_ARG;
  *out = nullptr;
  try {
    Ctx* c = new Ctx(device);
    *out = new sffgpu_ctx{c};
    return SFFGPU_OK;
  } catch (const HipError& e) {
    g_create_err = e.msg;
    return SFFGPU_ERR_HIP;
  }
}

void sffgpu_destroy(sffgpu_ctx* ctx) {
  if (!ctx) return;
  delete ctx->c;
  delete ctx;
}

const char* sffgpu_last_error(sffgpu_ctx* ctx) { return ctx ? ctx->c->err.c_str() : g_create_err.c_str(); }

int sffgpu_mesh_upload(sffgpu_ctx* ctx, int role, const double* tri9, int n_tri) {
  if (!ctx || n_tri < 0 || (n_tri > 0 && !tri9)) return SFFGPU_ERR_ARG;
  GUARD(ctx, ctx->c->upload_mesh(role, tri9, n_tri));
}

int sffgpu_collide_poses(sffgpu_ctx* ctx, const double* pos6, int n, uint8_t* hit) {
  if (!ctx || n < 0 || (n > 0 && (!pos6 || !hit))) return SFFGPU_ERR_ARG;
  GUARD(ctx, ctx->c->collide_poses(pos6, n, hit));
}

int sffgpu_collide_transforms(sffgpu_ctx* ctx, const double* rt12, int n, uint8_t* hit) {
  if (!ctx || n < 0 || (n > 0 && (!rt12 || !hit))) return SFFGPU_ERR_ARG;
  GUARD(ctx, ctx->c->collide_poses(rt12, n, hit, /*explicit_rt=*/true));
}

int sffgpu_collide_segments(sffgpu_ctx* ctx, const double* a6, const double* b6, int n, uint8_t* is_free,
                            int32_t* first_hit, int32_t* n_samples) {
  if (!ctx || n < 0 || (n > 0 && (!a6 || !b6 || !is_free))) return SFFGPU_ERR_ARG;
  GUARD(ctx, ctx->c->collide_segments(a6, b6, n, is_free, first_hit, n_samples));
}

int sffgpu_sample_steer(sffgpu_ctx* ctx, const uint64_t* words, const double* center6, int n, double dist, int dim,
                        const double limits[6], double* out6, uint8_t* in_limits) {
  if (!ctx || n < 0 || (dim != 2 && dim != 6) || (n > 0 && (!words || !center6 || !limits || !out6 || !in_limits)))
    return SFFGPU_ERR_ARG;
  GUARD(ctx, ctx->c->sample_steer(words, center6, n, dist, dim, limits, out6, in_limits));
}

int sffgpu_nodes_reset(sffgpu_ctx* ctx, int capacity) {
  if (!ctx || capacity < 0) return SFFGPU_ERR_ARG;
  GUARD(ctx, ctx->c->store_reset(capacity));
}
int sffgpu_nodes_append(sffgpu_ctx* ctx, const double* pos6, const int32_t* tree_id, int n) {
  if (!ctx || n < 0 || (n > 0 && (!pos6 || !tree_id))) return SFFGPU_ERR_ARG;
  GUARD(ctx, ctx->c->store_append(pos6, tree_id, n));
}
int sffgpu_nodes_index(sffgpu_ctx* ctx, const double limits[6], double cell) {
  if (!ctx || !limits || !(cell > 0)) return SFFGPU_ERR_ARG;
  GUARD(ctx, {
    Ctx& c = *ctx->c;
    c.grid_bk = 8;
    c.grid_cell0 = cell;
    c.grid_rebuilds = 0;
    c.grid_setup(limits, cell);
    c.grid_insert_new();
    c.grid_check(/*bulk=*/true);
  });
}
int sffgpu_nodes_count(sffgpu_ctx* ctx) { return ctx ? ctx->c->store_n : SFFGPU_ERR_ARG; }
int sffgpu_kernel_times(sffgpu_ctx* ctx, double ms[3], uint64_t launches[3]) {
  if (!ctx || !ms || !launches) return SFFGPU_ERR_ARG;
  for (int k = 0; k < 3; ++k) { ms[k] = ctx->c->kernel_ms_total(k); launches[k] = ctx->c->kernel_calls[k]; }
  return SFFGPU_OK;
}

int sffgpu_radius(sffgpu_ctx* ctx, const double* q6, int nq, const double* r, const int32_t* tree,
                  const int32_t* max_id, int32_t* idx, double* dist, int32_t* cnt, int cap) {
  if (!ctx || nq < 0 || cap <= 0 || (nq > 0 && (!q6 || !r || !idx || !cnt))) return SFFGPU_ERR_ARG;
  GUARD(ctx, ctx->c->radius(q6, nq, r, tree, max_id, idx, dist, cnt, cap));
}
int sffgpu_knn(sffgpu_ctx* ctx, const double* q6, int nq, int k, const int32_t* tree, const int32_t* max_id,
               int32_t* idx, double* dist, int32_t* cnt) {
  if (!ctx || nq < 0 || k <= 0 || (nq > 0 && (!q6 || !idx || !cnt))) return SFFGPU_ERR_ARG;
  GUARD(ctx, ctx->c->knn(q6, nq, k, tree, max_id, idx, dist, cnt));
}

int sffgpu_forest_create(sffgpu_ctx* ctx, const sffgpu_forest_cfg* cfg, const double* roots6, int n_roots,
                         sffgpu_forest** out) {
  if (!ctx || !cfg || !roots6 || n_roots <= 0 || !out) return SFFGPU_ERR_ARG;
  *out = nullptr;
  GUARD(ctx, {
    Forest* F = new Forest(ctx->c, *cfg, roots6, n_roots);
    sffgpu_forest* h = new sffgpu_forest;
    h->f = F;
    h->owner = ctx;
    *out = h;
  });
}
void sffgpu_forest_destroy(sffgpu_forest* f) {
  if (!f) return;
  sff::forest_profile_dump();
  delete f->f;
  delete f;
}
int sffgpu_forest_run(sffgpu_forest* f, int max_waves) {
  if (!f) return SFFGPU_ERR_ARG;
  try {
    f->f->run(max_waves);
  } catch (const HipError& e) {
    f->owner->c->err = e.msg;
    return SFFGPU_ERR_HIP;
  }
  return f->f->need_host_exchange ? SFFGPU_NEED_HOST_EXCHANGE : SFFGPU_OK;
}
int sffgpu_rccl_unique_id(uint8_t id128[128]) {
  if (!id128) return SFFGPU_ERR_ARG;
  try { Ctx::rccl_unique_id(id128); } catch (const HipError&) { return SFFGPU_ERR_HIP; }
  return SFFGPU_OK;
}
int sffgpu_ctx_set_allgather(sffgpu_ctx* ctx, sffgpu_allgather_fn fn, void* user, int rank, int world) {
  if (!ctx) return SFFGPU_ERR_ARG;
  if (fn && (world < 1 || rank < 0 || rank >= world)) return SFFGPU_ERR_ARG;
  GUARD(ctx, { ctx->c->sync(); ctx->c->xchg_fn = fn; ctx->c->xchg_user = user; ctx->c->xchg_rank = fn ? rank : 0; ctx->c->xchg_world = fn ? world : 1; });
}
int sffgpu_ctx_rccl_init(sffgpu_ctx* ctx, const uint8_t id128[128], int rank, int world) {
  if (!ctx || !id128) return SFFGPU_ERR_ARG;
  GUARD(ctx, ctx->c->rccl_init(id128, rank, world));
}
int sffgpu_forest_get_stats(sffgpu_forest* f, sffgpu_forest_stats* out) {
  if (!f || !out) return SFFGPU_ERR_ARG;
  GUARD(f->owner, f->f->fill_stats(out));
}
int sffgpu_forest_get_nodes(sffgpu_forest* f, double* pos6, int32_t* parent, int32_t* tree, int32_t* iter, double* cost,
                            double* dist_parent) {
  if (!f) return SFFGPU_ERR_ARG;
  Forest& F = *f->f;
  try { F.sync_host(); } catch (const HipError& e) { f->owner->c->err = e.msg; return SFFGPU_ERR_HIP; }
  for (size_t i = 0; i < F.nodes.size(); ++i) {
    const FNode& n = F.nodes[i];
    if (pos6) memcpy(pos6 + 6 * i, n.pos, sizeof n.pos);
    if (parent) parent[i] = n.parent;
    if (tree) tree[i] = n.tree;
    if (iter) iter[i] = (int32_t)n.iter;
    if (cost) cost[i] = n.d_root;
    if (dist_parent) dist_parent[i] = n.d_closest;
  }
  return SFFGPU_OK;
}
int sffgpu_forest_get_borders(sffgpu_forest* f, int32_t* ta, int32_t* tb, int32_t* n1, int32_t* n2, double* dist,
                              int cap) {
  if (!f) return SFFGPU_ERR_ARG;
  try { f->f->sync_host(); } catch (const HipError& e) { f->owner->c->err = e.msg; return SFFGPU_ERR_HIP; }
  int k = 0;
  for (auto& kv : f->f->borders)
    for (const Border& b : kv.second) {
      if (k < cap) {
        if (ta) ta[k] = kv.first.first;
        if (tb) tb[k] = kv.first.second;
        if (n1) n1[k] = b.n1;
        if (n2) n2[k] = b.n2;
        if (dist) dist[k] = b.dist;
      }
      ++k;
    }
  return k;
}
uint64_t sffgpu_forest_fingerprint(sffgpu_forest* f) {
  if (!f) return 0;
  try { f->f->sync_host(); } catch (const HipError& e) { f->owner->c->err = e.msg; return 0; }
  return f->f->fingerprint();
}

int sffgpu_forest_paths(sffgpu_forest* f, double* dist, int32_t* connected, int cap_connected) {
  if (!f || !dist) return SFFGPU_ERR_ARG;
  Forest& F = *f->f;
  try { F.sync_host(); } catch (const HipError& e) { f->owner->c->err = e.msg; return SFFGPU_ERR_HIP; }
  F.max_connected();                 // Solver::connectedTrees as Solve() leaves it (src/forest.h:196-206)
  F.get_paths();
  F.get_all_paths();
  for (int i = 0; i < F.num_roots; ++i)
    for (int j = 0; j < F.num_roots; ++j) dist[(size_t)i * F.num_roots + j] = i == j ? 0.0 : F.NM(i, j).dist;
  if (connected)
    for (size_t k = 0; k < F.connected.size() && (int)k < cap_connected; ++k) connected[k] = F.connected[k];
  return (int)F.connected.size();
}
int sffgpu_forest_smooth_paths(sffgpu_forest* f, double* dist) {
  if (!f || !dist) return SFFGPU_ERR_ARG;
  Forest& F = *f->f;
  if (F.nm.empty()) { f->owner->c->err = "smooth_paths: call sffgpu_forest_paths first"; return SFFGPU_ERR_STATE; }
  GUARD(f->owner, {
    F.smooth_paths();
    for (int i = 0; i < F.num_roots; ++i)
      for (int j = 0; j < F.num_roots; ++j) dist[(size_t)i * F.num_roots + j] = i == j ? 0.0 : F.NM(i, j).dist;
  });
}
int sffgpu_forest_path_plan(sffgpu_forest* f, int i, int j, int32_t* node_ids, int cap) {
  if (!f || i < 0 || j < 0 || i >= f->f->num_roots || j >= f->f->num_roots) return SFFGPU_ERR_ARG;
  Forest& F = *f->f;
  if (F.nm.empty() || i == j) return 0;
  const std::vector<int>& p = F.NM(i, j).plan;
  for (size_t k = 0; k < p.size() && (int)k < cap; ++k) node_ids[k] = p[k];
  return (int)p.size();
}

int sffgpu_rrt_create(sffgpu_ctx* ctx, const sffgpu_rrt_cfg* cfg, const double* roots6, int n_roots, sffgpu_rrt** out) {
  if (!ctx || !cfg || !roots6 || n_roots <= 0 || !out) return SFFGPU_ERR_ARG;
  *out = nullptr;
  GUARD(ctx, {
    Rrt* R = new Rrt(ctx->c, *cfg, roots6, n_roots);
    sffgpu_rrt* h = new sffgpu_rrt;
    h->r = R;
    h->owner = ctx;
    *out = h;
  });
}
void sffgpu_rrt_destroy(sffgpu_rrt* r) {
  if (!r) return;
  delete r->r;
  delete r;
}
int sffgpu_rrt_run(sffgpu_rrt* r, int max_iterations) {
  if (!r) return SFFGPU_ERR_ARG;
  GUARD(r->owner, r->r->run(max_iterations));
}
int sffgpu_rrt_get_stats(sffgpu_rrt* r, sffgpu_rrt_stats* out) {
  if (!r || !out) return SFFGPU_ERR_ARG;
  Rrt& R = *r->r;
  sffgpu_rrt_stats s = R.st;
  s.iterations = R.iter;
  s.solved = R.solved;
  s.n_nodes = (int)R.nodes.size();
  s.n_live_trees = (int)R.tree_frontier.size();
  int nl = 0;
  for (auto& l : R.links) nl += (int)l.size();
  s.n_links = nl;
  s.rng_draws = R.rng.draws;
  s.lazy_distance = R.lazy_distance;
  *out = s;
  return SFFGPU_OK;
}
int sffgpu_rrt_lazy_plan(sffgpu_rrt* r, int32_t* node_ids, int cap) {
  if (!r) return SFFGPU_ERR_ARG;
  Rrt& R = *r->r;
  if (R.lazy_last < 0) return 0;
  std::vector<int> chain;
  for (int n = R.lazy_last;; n = R.nodes[n].parent) { chain.push_back(n); if (R.nodes[n].d_root == 0) break; }   // Node::IsRoot()
  std::reverse(chain.begin(), chain.end());
  for (int k = 0; k < (int)chain.size() && k < cap; ++k) if (node_ids) node_ids[k] = chain[k];
  return (int)chain.size();
}
int sffgpu_rrt_get_nodes(sffgpu_rrt* r, double* pos6, int32_t* parent, int32_t* tree, int32_t* root_tree, int32_t* iter,
                         double* cost, double* dist_parent) {
  if (!r) return SFFGPU_ERR_ARG;
  Rrt& R = *r->r;
  for (size_t i = 0; i < R.nodes.size(); ++i) {
    const RNode& n = R.nodes[i];
    if (pos6) memcpy(pos6 + 6 * i, n.pos, sizeof n.pos);
    if (parent) parent[i] = n.parent;
    if (tree) tree[i] = n.tree;
    if (root_tree) root_tree[i] = n.root_tree;
    if (iter) iter[i] = (int32_t)n.iter;
    if (cost) cost[i] = n.d_root;
    if (dist_parent) dist_parent[i] = n.d_closest;
  }
  return SFFGPU_OK;
}
int sffgpu_rrt_get_links(sffgpu_rrt* r, int32_t* tree, int32_t* n1, int32_t* n2, double* dist, int cap) {
  if (!r) return SFFGPU_ERR_ARG;
  int k = 0;
  for (size_t t = 0; t < r->r->links.size(); ++t)
    for (const RLink& l : r->r->links[t]) {
      if (k < cap) {
        if (tree) tree[k] = (int32_t)t;
        if (n1) n1[k] = l.n1;
        if (n2) n2[k] = l.n2;
        if (dist) dist[k] = l.dist;
      }
      ++k;
    }
  return k;
}

int sffgpu_rrt_paths(sffgpu_rrt* r, double* dist, int32_t* connected, int cap_connected) {
  if (!r || !dist) return SFFGPU_ERR_ARG;
  Rrt& R = *r->r;
  R.get_paths();
  const int nt = (int)R.trees.size();
  for (int i = 0; i < nt; ++i)
    for (int j = 0; j < nt; ++j)
      dist[(size_t)i * nt + j] = i == j ? 0.0 : R.nm[(size_t)std::min(i, j) * nt + std::max(i, j)].dist;
  if (connected)
    for (size_t k = 0; k < R.connected.size() && (int)k < cap_connected; ++k) connected[k] = R.connected[k];
  return (int)R.connected.size();
}
int sffgpu_rrt_path_plan(sffgpu_rrt* r, int i, int j, int32_t* node_ids, int cap) {
  if (!r) return SFFGPU_ERR_ARG;
  Rrt& R = *r->r;
  const int nt = (int)R.trees.size();
  if (R.nm.empty() || i == j || i < 0 || j < 0 || i >= nt || j >= nt) return 0;
  const std::vector<int>& p = R.nm[(size_t)std::min(i, j) * nt + std::max(i, j)].plan;
  for (size_t k = 0; k < p.size() && (int)k < cap; ++k) node_ids[k] = p[k];
  return (int)p.size();
}
int sffgpu_rrt_smooth_paths(sffgpu_rrt* r) {
  if (!r) return SFFGPU_ERR_ARG;
  Rrt& R = *r->r;
  if (R.nm.empty()) { r->owner->c->err = "rrt smooth_paths: call sffgpu_rrt_paths first"; return SFFGPU_ERR_STATE; }
  try {
    R.smooth_paths();
  } catch (const HipError& e) {
    r->owner->c->err = e.msg;
    return SFFGPU_ERR_HIP;
  }
  return (int)R.link_plans.size();
}
int sffgpu_rrt_link_plan(sffgpu_rrt* r, int k, int32_t* node_ids, int cap) {
  if (!r) return SFFGPU_ERR_ARG;
  Rrt& R = *r->r;
  if (k < 0 || k >= (int)R.link_plans.size()) return 0;
  const std::vector<int>& p = R.link_plans[k];
  for (size_t q = 0; q < p.size() && (int)q < cap; ++q) node_ids[q] = p[q];
  return (int)p.size();
}
int sffgpu_forest_get_parent_history(sffgpu_forest* f, int32_t* node, int32_t* parent, int32_t* iteration, int cap) {
  if (!f || cap < 0) return SFFGPU_ERR_ARG;
  Forest& F = *f->f;
  if (!F.cfg.record_parents) { f->owner->c->err = "forest: created without record_parents"; return SFFGPU_ERR_ARG; }
  try { F.sync_host(); } catch (const HipError& e) { f->owner->c->err = e.msg; return SFFGPU_ERR_HIP; }
  if (F.hist_overflow) { f->owner->c->err = "forest: the device's parent-history buffer ran over"; return SFFGPU_ERR_CAPACITY; }
  if (!F.cfg.optimize) {   // plain SFF: a node keeps the parent it was created with
    const int n = (int)F.nodes.size();
    for (int i = 0; i < n && i < cap; ++i) {
      if (node) node[i] = i;
      if (parent) parent[i] = F.nodes[i].parent;
      if (iteration) iteration[i] = (int32_t)F.nodes[i].iter;
    }
    return n;
  }
  // (entries of one iteration touch distinct nodes; creation order of the entries is kept among equal keys)
  // Only what was appended since the last call is sorted, then merged in (the compat layer asks once per iter_<k> dump).
  {
    auto less = [](const Forest::HistRec& a, const Forest::HistRec& b) { return a.iter < b.iter; };
    if (F.hist_sorted > F.hist.size()) F.hist_sorted = 0;
    std::stable_sort(F.hist.begin() + F.hist_sorted, F.hist.end(), less);
    std::inplace_merge(F.hist.begin(), F.hist.begin() + F.hist_sorted, F.hist.end(), less);   // (stable: older entries first)
    F.hist_sorted = F.hist.size();
  }
  const int n = (int)F.hist.size();
  for (int i = 0; i < n && i < cap; ++i) {
    if (node) node[i] = F.hist[i].node;
    if (parent) parent[i] = F.hist[i].parent;
    if (iteration) iteration[i] = (int32_t)F.hist[i].iter;
  }
  return n;
}
int sffgpu_forest_get_frontier(sffgpu_forest* f, int32_t* node_ids, int cap) {
  if (!f) return SFFGPU_ERR_ARG;
  Forest& F = *f->f;
  try { F.sync_host(); } catch (const HipError& e) { f->owner->c->err = e.msg; return SFFGPU_ERR_HIP; }
  int k = 0;
  auto put = [&](int id) { if (k < cap && node_ids) node_ids[k] = id; ++k; };
  if (F.use_priority()) {
    for (auto& hs : F.heaps)
      if (!hs.empty()) for (int id : hs[0].v) put(id);
  } else {
    for (int id : F.frontier) put(id);
  }
  return k;
}

int sffgpu_forest_in_wave(sffgpu_forest* f) {
  if (!f) return SFFGPU_ERR_ARG;
  const Forest& F = *f->f;
  return ((F.dev.active && F.dev.host_stale) ? F.dev.last.in_wave != 0 : F.in_wave) ? 1 : 0;
}
int sffgpu_forest_round_begin(sffgpu_forest* f, int32_t* n_words, int32_t* done) {
  if (!f || !n_words || !done) return SFFGPU_ERR_ARG;
  GUARD(f->owner, {
    Forest& F = *f->f;
    *done = 0;
    *n_words = 0;
    if (F.dev.active) F.dev_to_host();   // the round protocol runs on the host path
    if (!F.in_wave && F.terminated()) {
      *done = 1;
    } else {
      F.round_begin();
      *n_words = (int32_t)F.records.size();
    }
  });
}
// ---- multi-GPU on the device-resident engine: the caller owns the collective of a round
int sffgpu_ctx_set_stream(sffgpu_ctx* ctx, void* hip_stream) {
  if (!ctx) return SFFGPU_ERR_ARG;
  GUARD(ctx, ctx->c->set_stream(static_cast<hipStream_t>(hip_stream)));
}
int sffgpu_forest_device_engine(sffgpu_forest* f) { return f ? (f->f->dev.on ? 1 : 0) : SFFGPU_ERR_ARG; }
long long sffgpu_forest_exchange_bytes(sffgpu_forest* f) { return f ? (long long)f->f->dev_exchange_bytes() : SFFGPU_ERR_ARG; }
int sffgpu_forest_dev_wave_begin(sffgpu_forest* f, int32_t* done) {
  if (!f || !done) return SFFGPU_ERR_ARG;
  GUARD(f->owner, *done = f->f->dev_wave_begin() ? 0 : 1);
}
int sffgpu_forest_dev_round_eval(sffgpu_forest* f, void* send_dev) {
  if (!f) return SFFGPU_ERR_ARG;
  // (SFFGPU_TEST_EXCHANGE_SELF: a one-rank forest packs and unpacks too - bench.py --force-dist prices the exchange with it)
  GUARD(f->owner, f->f->dev_enqueue_round_eval((f->f->cfg.world > 1 || f->f->test_exchange_self) ? send_dev : nullptr));
}
int sffgpu_forest_dev_round_commit(sffgpu_forest* f, const void* recv_dev) {
  if (!f) return SFFGPU_ERR_ARG;
  GUARD(f->owner, f->f->dev_enqueue_round_commit((f->f->cfg.world > 1 || f->f->test_exchange_self) ? recv_dev : nullptr));
}
int sffgpu_forest_dev_wave_end(sffgpu_forest* f, int32_t* fault) {
  if (!f || !fault) return SFFGPU_ERR_ARG;
  GUARD(f->owner, {
    f->f->dev_enqueue_end();
    *fault = f->f->dev_finish_wave(nullptr);
    if (*fault) {                      // (a list overflowed: the caller finishes the wave through the host protocol)
      ++f->f->st.host_fallback_waves;
      f->f->dev_to_host();
      f->f->on_list_fault();
    }
  });
}
int sffgpu_forest_rounds_per_wave(sffgpu_forest* f) { return f ? std::max(1, f->f->cfg.threshold_misses) : SFFGPU_ERR_ARG; }

int sffgpu_forest_round_records(sffgpu_forest* f, int32_t* words, int cap_words) {
  if (!f || !words) return SFFGPU_ERR_ARG;
  Forest& F = *f->f;
  if ((int)F.records.size() > cap_words) return SFFGPU_ERR_CAPACITY;
  memcpy(words, F.records.data(), F.records.size() * sizeof(int32_t));
  return SFFGPU_OK;
}
int sffgpu_forest_round_commit(sffgpu_forest* f, const int32_t* all_words, int total_words, const int32_t* words_per_rank, int world) {
  if (!f || !all_words || !words_per_rank || world < 1) return SFFGPU_ERR_ARG;
  GUARD(f->owner, f->f->round_commit(all_words, total_words, words_per_rank, world));
}

}  // extern "C"
