// rrt.cpp — RRT / RRT* / Multi-T-RRT engine of libsffgpu (reference src/rrt.h).
//
// Reference: RapidExpTree<T,R> — constructor src/rrt.h:47-83, Solve() :86-99, expandNode :128-322
// (nearest + steer :143-151, RRT* choose-parent / rewire :156-201, connect-and-merge :219-319).
// Round 1 keeps the reference's one-sample-per-iteration order (every iteration's nearest
// neighbour depends on the previous iteration's node) and runs each step's queries on the GPU
// in batches: nearest / k-nearest through the exact neighbour sweep, the new pose and every
// candidate edge of the iteration through the collision kernels.  A speculative multi-sample
// wave like the SFF engine's is the next step.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>

#include "engine.h"
#include "sff_geom.h"

namespace sff {

#define HIPCHK(x) hip_check((x), #x)

Rrt::Rrt(Ctx* c, const sffgpu_rrt_cfg& cf, const double* roots6, int n_roots) : ctx(c), cfg(cf) {
  if (const char* e = getenv("SFFGPU_RRT_CHAIN")) chain_on = atoi(e) != 0;
  if (const char* e = getenv("SFFGPU_RRT_FORK")) c->rr_fork = atoi(e) != 0;
  if (const char* e = getenv("SFFGPU_RRT_REPAIR")) repair_on = atoi(e) != 0;
  if (const char* e = getenv("SFFGPU_RRT_DRY")) dry_on = atoi(e) != 0;
  if (const char* e = getenv("SFFGPU_RRT_ONE_CHAIN")) one_chain = atoi(e) != 0;
  if (const char* e = getenv("SFFGPU_RRT_SPLIT")) split_parts = std::max(1, atoi(e));
  if (const char* e = getenv("SFFGPU_RRT_SMALL")) { small_cap = std::max(1, atoi(e)); }
  if (const char* e = getenv("SFFGPU_RRT_GROW")) grow_pct = std::max(100, atoi(e));
  if (cfg.dim != 2 && cfg.dim != 6) throw HipError{"rrt: dim must be 2 or 6"};
  if (n_roots < 1) throw HipError{"rrt: at least one root"};
  if (cfg.priority_bias != 0 && !cfg.has_goal) throw HipError{"rrt: goal bias needs a goal (src/main.cpp:330-331)"};
  if (!c->have_env || !c->have_robot) throw HipError{"rrt: upload ENV and ROBOT meshes first"};
  if (cfg.lazy_edge && (n_roots != 1 || cfg.has_goal || cfg.priority_bias != 0))
    throw HipError{"rrt: lazy_edge grows ONE tree from one root towards cfg.goal (has_goal = 0, no goal bias)"};
  rng.reseed(cfg.seed);
  for (uint64_t k = 0; k < cfg.rng_skip; ++k) (void)rng.next();
  const int nt = n_roots + (cfg.has_goal ? 1 : 0);
  trees.resize(nt);
  links.resize(nt);
  eaten.resize(nt);
  ctx->store_reset(std::max(4096, cfg.max_iterations + nt + 16));
  if (!getenv("SFFGPU_RRT_NO_GRID")) {
    // index over the store (the counterpart of flannIndex->buildIndex, src/rrt.h:56-62): used by knn() above
    const double cell = 1.01 * std::max(cfg.sampling_dist, cfg.dist_tree);
    ctx->grid_bk = 8;
    ctx->grid_cell0 = cell;
    ctx->grid_rebuilds = 0;
    ctx->gridv_ovf_cap_next = 65536;
    ctx->grid_setup(cfg.limits, cell);
  }
  for (int j = 0; j < n_roots; ++j) {          // src/rrt.h:48-62
    add_node(roots6 + 6 * (size_t)j, j, j, -1, 0, 0, 0);
    tree_frontier.push_back(j);
  }
  num_trees = n_roots - 1;                      // :63
  if (cfg.has_goal) {                           // :66-82
    goal_node = add_node(cfg.goal, n_roots, n_roots, -1, 0, 0, 0);
    tree_frontier.push_back(n_roots);
  }
  memset(&st, 0, sizeof st);
}

int Rrt::add_node(const double* pos, int root_tree, int tree, int parent, double dc, double dr, unsigned it) {
  RNode n;
  memcpy(n.pos, pos, sizeof n.pos);
  n.root_tree = root_tree;
  n.tree = tree;
  n.parent = parent;
  n.d_closest = dc;
  n.d_root = dr;
  n.iter = it;
  n.idx_in_tree = (int)trees[tree].size();
  int id = (int)nodes.size();
  nodes.push_back(n);
  trees[tree].push_back(id);
  int32_t t = tree;
  if (defer_append) {                            // wave engine: appended in one batch at the end of the wave
    pend_pos.insert(pend_pos.end(), pos, pos + 6);
    pend_tree.push_back(t);
  } else {
    ctx->store_append(pos, &t, 1);              // replaces flannIndex->addPoints (:215)
  }
  return id;
}

// LazyTSP::runRRT's goal test (src/lazy.h:258-273): the new node within treeDistance of the goal ends the edge's
// search - there is no edge check towards the goal
bool Rrt::lazy_goal_check(int new_id) {
  const double gd = sffg::dist6(cfg.goal, nodes[new_id].pos);
  if (!(gd < cfg.dist_tree)) return false;
  solved = true;
  lazy_distance = gd + nodes[new_id].d_root;
  lazy_last = new_id;
  return true;
}

RLink Rrt::make_link(int a, int b) {            // DistanceHolder(first, second), src/primitives.h:609-618
  double d = nodes[a].d_root + nodes[b].d_root + sffg::dist6(nodes[a].pos, nodes[b].pos);
  return {std::min(a, b), std::max(a, b), d};
}

// k nearest of one tree in the reference's order (distance, index in the tree's list)
// One tree that holds (nearly) every node - single-root RRT / RRT*, lazy edges, a forest after its merges: the
// queries are answered from the grid cells around them (k_knn_grid) instead of one sweep of the store per query.
// (Several live trees: a query for a small tree far away would grow its shells over the whole grid - linear sweep.)
bool Rrt::knn_by_grid(const int32_t* tree, int nq, int k) const {
  bool by_grid = ctx->grid_on && nq > 0 && (int)nodes.size() >= 2048;
  for (int i = 0; i < nq && by_grid; ++i) {
    const int t = tree ? tree[i] : -1;
    if (t >= 0 && (int)trees[t].size() * 10 < (int)nodes.size() * 9) by_grid = false;
    if (t >= 0 && (int)trees[t].size() < std::max(k, 1024)) by_grid = false;
  }
  return by_grid;
}

void Rrt::knn(const double* q, int nq, const int32_t* tree, int k, std::vector<std::vector<int>>& out) {
  std::vector<int32_t> idx((size_t)nq * k), cnt(nq);
  std::vector<double> dist((size_t)nq * k);
  const bool by_grid = knn_by_grid(tree, nq, k);
  ctx->knn(q, nq, k, tree, nullptr, idx.data(), dist.data(), cnt.data(), by_grid);
  out.assign(nq, {});
  for (int i = 0; i < nq; ++i) {
    struct E { double d; int order; int id; };
    std::vector<E> e;
    for (int j = 0; j < cnt[i]; ++j) {
      int id = idx[(size_t)i * k + j];
      e.push_back({dist[(size_t)i * k + j], nodes[id].idx_in_tree, id});
    }
    std::sort(e.begin(), e.end(), [](const E& a, const E& b) { return a.d < b.d || (a.d == b.d && a.order < b.order); });
    for (const E& x : e) out[i].push_back(x.id);
  }
  st.nn_queries += (uint64_t)nq;
}

static uint64_t seg_calls(int fh, int ns) { return fh > 0 ? (uint64_t)fh : (uint64_t)ns; }

// the iteration's steering target: the goal with probability priorityBias (:130-131), else
// RandGen::randomPointInSpace (src/randGen.h:124-146); Y is drawn before X (g++ evaluates the two arguments
// of point.set(...) right to left — pinned by tests/golden/ref_primitives.json)
void Rrt::draw_target(double rnd[6]) {
  using namespace sffg;
  if (cfg.priority_bias != 0 && uniform_real(rng.next(), 0.0, 1.0) <= cfg.priority_bias) {
    memcpy(rnd, nodes[goal_node].pos, 48);
    return;
  }
  double y = uniform_real(rng.next(), cfg.limits[2], cfg.limits[3]);
  double x = uniform_real(rng.next(), cfg.limits[0], cfg.limits[1]);
  rnd[0] = x; rnd[1] = y; rnd[2] = 0; rnd[3] = rnd[4] = rnd[5] = 0;
  if (cfg.dim == 6) {
    rnd[2] = uniform_real(rng.next(), cfg.limits[4], cfg.limits[5]);
    rnd[3] = uniform_real(rng.next(), -SFFG_PI, SFFG_PI);
    double phi = sffp::pacos(1 - 2 * uniform_real(rng.next(), 0.0, 1.0)) + SFFG_PI_2;
    if (uniform_real(rng.next(), 0.0, 1.0) < 0.5) { if (phi < 0) phi += SFFG_PI; else phi -= SFFG_PI; }
    rnd[4] = phi;
    rnd[5] = uniform_real(rng.next(), -SFFG_PI, SFFG_PI);
  }
}

void Rrt::expand(int tree_to_expand, unsigned iteration) {
  using namespace sffg;
  double rnd[6], np[6];
  draw_target(rnd);
  std::vector<std::vector<int>> res;
  int32_t tq = tree_to_expand;
  knn(rnd, 1, &tq, 1, res);                                                     // :143
  int nearest = res[0][0];
  steer(nodes[nearest].pos, rnd, cfg.sampling_dist, np);                        // :148
  uint8_t hit = 0, free_par = 0;
  int32_t fh = -1, ns = 0;
  ctx->collide_poses(np, 1, &hit);                                              // :149
  st.collide_calls += 1;
  if (hit) return;
  ctx->collide_segments(nodes[nearest].pos, np, 1, &free_par, &fh, &ns);
  st.path_free_calls += 1;
  st.collide_calls += seg_calls(fh, ns);
  if (!free_par) return;                                                        // :149-151
  int new_id;
  if (cfg.optimize) {                                                           // :156-201
    double best = dist6(np, nodes[nearest].pos) + nodes[nearest].d_root;
    const size_t krrt = (size_t)(2 * M_E * std::log10((double)nodes.size() + (cfg.lazy_edge ? 1.0 : 0.0)));   // src/lazy.h:199
    std::vector<int> kn;
    if (krrt > 0) {
      knn(np, 1, &tq, (int)krrt, res);
      kn = res[0];
    } else {
      st.nn_queries += 1;
    }
    // both directions of every candidate edge in one launch (choose-parent :169-175, rewire :181-201)
    const int m = (int)kn.size();
    std::vector<double> a((size_t)m * 12), b((size_t)m * 12);
    for (int j = 0; j < m; ++j) {
      memcpy(&a[12 * (size_t)j], np, 48);
      memcpy(&b[12 * (size_t)j], nodes[kn[j]].pos, 48);
      memcpy(&a[12 * (size_t)j + 6], nodes[kn[j]].pos, 48);
      memcpy(&b[12 * (size_t)j + 6], np, 48);
    }
    std::vector<uint8_t> fr((size_t)m * 2);
    std::vector<int32_t> fhs((size_t)m * 2), nss((size_t)m * 2);
    if (m) ctx->collide_segments(a.data(), b.data(), m * 2, fr.data(), fhs.data(), nss.data());
    for (int j = 0; j < m; ++j) {
      int nb = kn[j];
      double nd = dist6(np, nodes[nb].pos) + nodes[nb].d_root;
      if (nd < best - SFFG_TOL) {
        st.path_free_calls += 1;
        st.collide_calls += seg_calls(fhs[2 * j], nss[2 * j]);
        if (fr[2 * j]) { best = nd; nearest = nb; }
      }
    }
    new_id = add_node(np, nodes[nearest].root_tree, tree_to_expand, nearest, dist6(nodes[nearest].pos, np), best, iteration);
    for (int j = 0; j < m; ++j) {
      int nb = kn[j];
      double npd = dist6(nodes[nb].pos, np);
      double proposed = best + npd;
      if (proposed < nodes[nb].d_root - SFFG_TOL) {
        st.path_free_calls += 1;
        st.collide_calls += seg_calls(fhs[2 * j + 1], nss[2 * j + 1]);
        if (fr[2 * j + 1]) {
          nodes[nb].parent = new_id;
          nodes[nb].root_tree = nodes[new_id].root_tree;                       // :195
          nodes[nb].d_closest = npd;
          nodes[nb].d_root = proposed;
        }
      }
    }
  } else {                                                                      // :203
    new_id = add_node(np, nodes[nearest].root_tree, tree_to_expand, nearest, cfg.sampling_dist,
                      nodes[nearest].d_root + cfg.sampling_dist, iteration);
  }
  if (cfg.lazy_edge) { lazy_goal_check(new_id); return; }
  // :219-319 connect to / merge with the other live trees.  Merging changes neither the node set
  // of any third tree nor the new point, so the nearest node of every other live tree and its
  // edge check can be fetched up front in one sweep + one collision launch.
  std::vector<int32_t> others;
  for (int t : tree_frontier)
    if (t != tree_to_expand) others.push_back(t);
  const int no = (int)others.size();
  if (no == 0) return;
  std::vector<double> q((size_t)no * 6);
  for (int j = 0; j < no; ++j) memcpy(&q[6 * (size_t)j], np, 48);
  knn(q.data(), no, others.data(), 1, res);
  st.nn_queries -= (uint64_t)no;   // counted below, only for the trees the reference actually queries
  std::vector<double> a((size_t)no * 6), b((size_t)no * 6);
  std::vector<int> nbs(no);
  for (int j = 0; j < no; ++j) {
    nbs[j] = res[j][0];
    memcpy(&a[6 * (size_t)j], np, 48);
    memcpy(&b[6 * (size_t)j], nodes[nbs[j]].pos, 48);
  }
  std::vector<uint8_t> fr(no);
  std::vector<int32_t> fhs(no), nss(no);
  ctx->collide_segments(a.data(), b.data(), no, fr.data(), fhs.data(), nss.data());
  for (int i = 0; i < (int)tree_frontier.size(); ++i) {
    int tree = tree_frontier[i];
    if (tree == tree_to_expand) continue;
    int j = (int)(std::find(others.begin(), others.end(), tree) - others.begin());
    st.nn_queries += 1;
    int nb = nbs[j];
    double nd = dist6(nodes[nb].pos, np);
    if (!(nd < cfg.dist_tree)) continue;                                        // :231 (no TOLERANCE here)
    st.path_free_calls += 1;
    st.collide_calls += seg_calls(fhs[j], nss[j]);
    if (!fr[j]) continue;
    tree_to_expand = merge_or_link(tree_to_expand, new_id, nb, true, 0, 0, i);
  }
}

// src/rrt.h:233-316: link the new node to `nb` and merge the two trees (the one with the lower id eats the
// other); returns the surviving tree and fixes the caller's frontier cursor like the reference's --i.
int Rrt::merge_or_link(int tree_to_expand, int new_id, int nb, bool, int, int, int& i) {
  links[tree_to_expand].push_back(make_link(new_id, nb));                     // :233
    int nbt = nodes[nb].tree;
    int to = tree_to_expand < nbt ? tree_to_expand : nbt;
    int from = tree_to_expand < nbt ? nbt : tree_to_expand;
    std::vector<int32_t> moved(trees[from].begin(), trees[from].end());
    for (int id : trees[from]) {                                                // :240-250
      nodes[id].tree = to;
      nodes[id].idx_in_tree = (int)trees[to].size();
      trees[to].push_back(id);
    }
    if (!pend_tree.empty()) {                                                   // wave engine: nodes not on the device yet
      ctx->store_append(pend_pos.data(), pend_tree.data(), (int)pend_tree.size());
      pend_pos.clear();
      pend_tree.clear();
    }
    ctx->store_set_tree(moved.data(), (int)moved.size(), to);                   // the moved nodes now answer tree `to`
    for (RLink& l : links[to]) l = make_link(l.n1, l.n2);                       // :278-289
    for (const RLink& l : links[from]) links[to].push_back(make_link(l.n1, l.n2));  // :291-299
    eaten[to].push_back(from);                                                  // :305-308
    for (int t : eaten[from]) eaten[to].push_back(t);
    tree_frontier.erase(std::find(tree_frontier.begin(), tree_frontier.end(), from));  // :310-315
    tree_to_expand = to;
    solved = tree_frontier.size() == 1;
    --num_trees;
    --i;
    ++st.merges;
  return tree_to_expand;
}

// ---------------------------------------------------------------------------------------------------
// Speculative wave.  The reference runs one iteration at a time because iteration j's nearest neighbour may
// be the node iteration i < j has just added.  A wave evaluates B iterations against the FROZEN trees on the
// GPU (nearest node, steer, pose + parent edge, RRT* k-nearest edges, links to other trees) and then replays
// them in order on the host.  The replay checks, with the exact metric, whether a node accepted earlier in
// the same wave would have been the nearest neighbour of iteration j (or a tree merge happened): if so the
// wave is cut at j, the RNG is rewound to the start of iteration j, and j starts the next wave.  Whatever
// is committed is therefore exactly what the one-by-one loop produces, at every wave size.
// ---------------------------------------------------------------------------------------------------
namespace {
struct WCand {
  int tree;
  uint64_t draws_before;
  double rnd[6], np[6];
  int nearest;
  double d_near;
  bool pose_hit = false, par_free = false;
  int par_fh = -1, par_ns = 0;
  std::vector<int> members;                       // RRT*: k_max nearest store nodes of the tree, in order
  struct Edge { int other; bool free_f, free_b; int fh_f, ns_f, fh_b, ns_b; };
  std::vector<Edge> medges;                       // member edges (store members then mates), both directions
  struct Conn { int tree; int node; double d; int order; bool free; int fh, ns; };
  std::vector<Conn> conns;                        // nearest node of each OTHER tree within treeDistance
  int accepted = -1;
  // rows past the wave's slots are REPAIRED slots: evaluated as if the new point of row near_row (an earlier slot) were a node
  // and the slot's nearest one (that is what it is if that slot is accepted as speculated)
  int slot = -1, near_row = -1, alt_row = -1;
  bool cut_here = false;                          // nothing evaluated covers this slot if it is reached
  bool prepared = false;                          // its member lists / link candidates / edges are in the wave's batches
  int apos = -1;                                  // its place among the rows that got them
};
}  // namespace

static unsigned long long g_rrt_alt[3];   // SFFGPU_PROFILE: repaired rows evaluated, taken by the replay, waves cut for lack of one
static double g_rrt_sec[10];   // SFFGPU_PROFILE: ms in the sections of run_wave
int Rrt::run_wave(int B) {
  using namespace sffg;
  Ctx& c = *ctx;
  auto t_sec = std::chrono::steady_clock::now();
  auto lap = [&](int k) { const auto t = std::chrono::steady_clock::now(); g_rrt_sec[k] += std::chrono::duration<double, std::milli>(t - t_sec).count(); t_sec = t; };
  if (iter + B > cfg.max_iterations) B = cfg.max_iterations - iter;
  if (B <= 0) return 0;
  const Mt64 snapshot = rng;
  const int N0 = (int)nodes.size();
  std::vector<WCand> w(B);
  // ---- 1. per iteration: tree pick (:95) and steering target (:130-134), in the reference's draw order
  for (int j = 0; j < B; ++j) {
    w[j].draws_before = rng.draws;
    w[j].slot = j;
    w[j].tree = cfg.lazy_edge ? 0 : tree_frontier[rng.uniform_int(0, num_trees)];   // (src/lazy.h:181 draws no tree)
    draw_target(w[j].rnd);
  }
  const uint64_t draws_end = rng.draws;
  lap(0);
  const int kmax = cfg.optimize ? (int)(size_t)(2 * M_E * std::log10((double)(N0 + B) + (cfg.lazy_edge ? 1.0 : 0.0))) : 0;
  std::vector<int> alive;
  int nA = 0;
  bool chained = false;
  // (several live trees: the other trees' nodes around every new point ride the same chain, src/rrt.h:228-231)
  const int conn_cap = 32;
  bool conn_have = false;
  std::vector<int32_t> conn_i, conn_c;
  std::vector<double> conn_dd;
  bool have_mates = false;
  std::vector<int32_t> mate;
  if (chain_on && kmax <= 64) {
    // ---- 2-4 as ONE enqueued chain and one wait (Ctx::rrt_chain): the nearest node of the frozen tree (:143), the steered
    // point (:148), its pose and parent edge (:149-151), - RRT* - the k_max nearest store nodes of EVERY new point (:166;
    // those of the points that die in between are thrown away), - several trees - the other trees' nodes around it, and per slot
    // the earlier new point of the wave that would be its nearest node if it becomes one
    const int km = std::max(kmax, 1);
    std::vector<double> q((size_t)B * 6), np((size_t)B * 6), nd((size_t)B * 2), md((size_t)B * km);
    std::vector<int32_t> tq(B), ni((size_t)B * 2), nc(B), seg((size_t)B * 3), mi((size_t)B * km), mc(B);
    std::vector<uint8_t> hit(B);
    for (int j = 0; j < B; ++j) { memcpy(&q[6 * (size_t)j], w[j].rnd, 48); tq[j] = w[j].tree; }
    const bool conn_q = tree_frontier.size() > 1 && !getenv("SFFGPU_RRT_NO_CHAIN_CONN");
    if (conn_q) { conn_i.resize((size_t)B * conn_cap); conn_dd.resize((size_t)B * conn_cap); conn_c.resize(B); }
    mate.assign(B, -1);
    const bool by_gridk = kmax > 0 && knn_by_grid(tq.data(), B, kmax);
    const double conn_r = conn_q ? cfg.dist_tree : 0.0;
    Ctx::RrtRows R1{np.data(), hit.data(), seg.data(), mi.data(), md.data(), mc.data(), conn_i.data(), conn_dd.data(), conn_c.data()};
    // (the repaired rows ride the same chain: the device lists the slots that have a mate - at most alt_cap of them, a slot
    // beyond that ends the wave if it is reached - and evaluates them behind the others)
    const int alt_cap = one_chain && repair_on ? std::min(B, B / 4 + 32) : 0;
    const int acap = std::max(alt_cap, 1);
    std::vector<double> np2((size_t)acap * 6), md2((size_t)acap * km), cd2;
    std::vector<int32_t> seg2((size_t)acap * 3), mi2((size_t)acap * km), mc2(acap), ci2, cc2, d_slot(acap), d_mate(acap);
    std::vector<uint8_t> hit2(acap);
    if (conn_q) { ci2.resize((size_t)acap * conn_cap); cd2.resize((size_t)acap * conn_cap); cc2.resize(acap); }
    Ctx::RrtRows R2{np2.data(), hit2.data(), seg2.data(), mi2.data(), md2.data(), mc2.data(), ci2.data(), cd2.data(), cc2.data()};
    int32_t n_listed = 0;
    c.rrt_chain(q.data(), tq.data(), B, cfg.sampling_dist, knn_by_grid(tq.data(), B, 1), kmax, by_gridk, ni.data(), nd.data(), nc.data(),
                mate.data(), R1, conn_r, conn_cap, alt_cap, d_slot.data(), d_mate.data(), &n_listed, &R2);
    conn_have = conn_q;
    // (the nearest node is the first by (distance, position in its tree); the device orders by (distance, id): two nodes at
    // exactly the same distance - or a query nobody answered - send the wave through the separate calls below)
    chained = true;
    for (int j = 0; j < B && chained; ++j)
      if (nc[j] < 1 || (nc[j] >= 2 && nd[2 * (size_t)j] == nd[2 * (size_t)j + 1])) chained = false;
    if (chained) {
      lap(1);
      // one row of results -> its candidate (an edge whose triangle candidate list ran over: by itself through the batch call)
      auto fill_row = [&](WCand& cd, const double* a_pos, const double* np_r, uint8_t hit_r, const int32_t* seg_r, int n_r, int r) {
        memcpy(cd.np, np_r + 6 * (size_t)r, 48);
        cd.pose_hit = hit_r != 0;
        int ns_j = seg_r[r], fh_j = seg_r[(size_t)n_r + r] == 0x7fffffff ? -1 : seg_r[(size_t)n_r + r];
        if (seg_r[2 * (size_t)n_r + r]) {
          uint8_t fr1 = 0; int32_t fh1 = -1, ns1 = 0;
          c.collide_segments(a_pos, cd.np, 1, &fr1, &fh1, &ns1);
          fh_j = fh1; ns_j = ns1;
        }
        cd.par_free = fh_j < 0;
        cd.par_fh = fh_j;
        cd.par_ns = ns_j;
      };
      auto fill_members = [&](WCand& cd, const int32_t* mi_r, const double* md_r, const int32_t* mc_r, int r) {
        // (the device's order is (distance, id): already the reference's (distance, position in the tree) unless two nodes
        // are at one distance and the tree's list is not in id order - after a merge)
        const int n_m = mc_r[r];
        const int32_t* ids = mi_r + (size_t)r * kmax;
        const double* ds = md_r + (size_t)r * kmax;
        bool in_order = true;
        for (int m = 1; m < n_m && in_order; ++m)
          in_order = ds[m - 1] < ds[m] || (ds[m - 1] == ds[m] && nodes[ids[m - 1]].idx_in_tree < nodes[ids[m]].idx_in_tree);
        cd.members.assign(ids, ids + n_m);
        if (in_order) return;
        struct E { double d; int order; int id; };
        std::vector<E> e;
        for (int m = 0; m < n_m; ++m) e.push_back({ds[m], nodes[ids[m]].idx_in_tree, ids[m]});
        std::sort(e.begin(), e.end(), [](const E& a, const E& b) { return a.d < b.d || (a.d == b.d && a.order < b.order); });
        cd.members.clear();
        for (const E& x : e) cd.members.push_back(x.id);
      };
      for (int j = 0; j < B; ++j) {
        WCand& cd = w[j];
        cd.slot = j;
        cd.nearest = ni[2 * (size_t)j];
        cd.d_near = dist6(cd.rnd, nodes[cd.nearest].pos);
        fill_row(cd, nodes[cd.nearest].pos, np.data(), hit[j], seg.data(), B, j);
      }
      // ---- the repaired slots: a slot whose nearest node would be an earlier new point of the wave is evaluated a second
      // time from that point (Ctx::rrt_chain_alt) - the replay takes that row when the earlier slot is accepted as speculated
      have_mates = true;
      // rows behind the wave's slots, one per listed (slot, mate); a mate the device took for alive and that is not (its
      // edge's candidate list had run over) leaves its row unused and the slot uncovered
      std::vector<int32_t> a_slot, a_mate;
      std::vector<uint8_t> a_use;
      int seg_rows = 0;
      if (alt_cap > 0) {
        std::vector<uint8_t> listed(B, 0);
        for (int r = 0; r < n_listed; ++r) {
          const int js = d_slot[r], i = d_mate[r];
          if (js < 0 || js >= B || i != mate[js]) throw HipError{"rrt: repaired-slot list out of step (internal error)"};
          listed[js] = 1;
          const bool use = !w[i].pose_hit && w[i].par_free;
          if (!use) w[js].cut_here = true;
          a_slot.push_back(js); a_mate.push_back(i); a_use.push_back(use ? 1 : 0);
        }
        for (int js = 0; js < B; ++js)
          if (mate[js] >= 0 && !listed[js]) w[js].cut_here = true;      // (more slots with a mate than rows)
        seg_rows = alt_cap;
      } else {
        for (int js = 0; js < B; ++js) {
          const int i = mate[js];
          if (i < 0) continue;
          if (!repair_on || w[i].pose_hit || !w[i].par_free) { w[js].cut_here = true; continue; }   // (what the device took for alive is not)
          a_slot.push_back(js); a_mate.push_back(i); a_use.push_back(1);
        }
        seg_rows = (int)a_slot.size();
      }
      const int nalt = (int)a_slot.size();
      g_rrt_alt[0] += (unsigned long long)nalt;
      if (nalt > 0) {
        if (alt_cap == 0) {   // (the second chain, from the host's list)
          np2.resize((size_t)nalt * 6); md2.resize((size_t)nalt * km); seg2.resize((size_t)nalt * 3); mi2.resize((size_t)nalt * km); mc2.resize(nalt);
          hit2.resize(nalt);
          if (conn_q) { ci2.resize((size_t)nalt * conn_cap); cd2.resize((size_t)nalt * conn_cap); cc2.resize(nalt); }
          Ctx::RrtRows R3{np2.data(), hit2.data(), seg2.data(), mi2.data(), md2.data(), mc2.data(), ci2.data(), cd2.data(), cc2.data()};
          c.rrt_chain_alt(a_slot.data(), a_mate.data(), nalt, cfg.sampling_dist, kmax, by_gridk, R3, conn_r, conn_cap);
        }
        w.resize((size_t)B + nalt);
        for (int r = 0; r < nalt; ++r) {
          WCand& cd = w[(size_t)B + r];
          const WCand& sl = w[a_slot[r]];
          cd.tree = sl.tree;
          cd.draws_before = sl.draws_before;
          memcpy(cd.rnd, sl.rnd, 48);
          cd.slot = a_slot[r];
          cd.near_row = a_mate[r];
          cd.nearest = -1;                                   // (the node row near_row becomes: known in the replay)
          if (!a_use[r]) { cd.pose_hit = true; cd.d_near = 0; memcpy(cd.np, sl.np, 48); continue; }   // (nobody takes this row)
          cd.d_near = dist6(cd.rnd, w[a_mate[r]].np);
          fill_row(cd, w[a_mate[r]].np, np2.data(), hit2[r], seg2.data(), seg_rows, r);
          w[a_slot[r]].alt_row = B + r;
          if (kmax > 0 && !cd.pose_hit && cd.par_free) fill_members(cd, mi2.data(), md2.data(), mc2.data(), r);
        }
        if (conn_q) {   // (one list for all rows)
          conn_i.insert(conn_i.end(), ci2.begin(), ci2.end());
          conn_dd.insert(conn_dd.end(), cd2.begin(), cd2.end());
          conn_c.insert(conn_c.end(), cc2.begin(), cc2.end());
        }
      }
      lap(2);
      // ---- which row of which slot will the replay take, and where will it stop?  That depends on the nearest-node logic
      // alone (a row that is alive becomes a node whatever its other edges say): the replay's walk is done once ahead, dry, and
      // only the rows it takes get member lists, link candidates and edges - nothing past the cut, not both rows of a slot.
      // (Tree merges end the real replay earlier; a row the replay reaches unprepared ends the wave there.)
      if (dry_on) {
        std::vector<int> alt_acc;
        std::vector<char> took(w.size(), 0);
        for (int j = 0; j < B; ++j) {
          if (w[j].cut_here) break;
          int row = j;
          const int cand = mate[j];
          if (cand >= 0) {
            if (!took[cand] || w[j].alt_row < 0) break;
            row = w[j].alt_row;
          }
          const WCand& cr = w[row];
          bool conflict = false;
          for (int i : alt_acc) {
            if (w[i].tree != cr.tree) continue;
            if (std::fabs(w[i].np[0] - cr.rnd[0]) > cr.d_near) continue;
            if (dist6(cr.rnd, w[i].np) <= cr.d_near) { conflict = true; break; }
          }
          if (conflict) break;
          if (cr.pose_hit || !cr.par_free) continue;
          took[row] = 1;
          w[row].prepared = true;
          alive.push_back(row);
          if (row != j) alt_acc.push_back(row);
        }
      } else {
        // (rows in the order of their slots, a slot's repaired row after its speculated one)
        for (int j = 0; j < B; ++j) {
          if (!w[j].pose_hit && w[j].par_free) { alive.push_back(j); w[j].prepared = true; }
          const int a = w[j].alt_row;
          if (a >= 0 && !w[a].pose_hit && w[a].par_free) { alive.push_back(a); w[a].prepared = true; }
        }
      }
      nA = (int)alive.size();
      if (kmax > 0)
        for (int j : alive)
          if (j < B) fill_members(w[j], mi.data(), md.data(), mc.data(), j);
      lap(3);
    }
  }
  if (!chained) {
    // ---- 2. nearest node of the frozen tree (:143), steer (:148)
    {
      std::vector<double> q((size_t)B * 6);
      std::vector<int32_t> tq(B);
      for (int j = 0; j < B; ++j) { memcpy(&q[6 * (size_t)j], w[j].rnd, 48); tq[j] = w[j].tree; }
      std::vector<std::vector<int>> res;
      const uint64_t keep = st.nn_queries;
      knn(q.data(), B, tq.data(), 1, res);
      st.nn_queries = keep;   // accounted per committed iteration below
      for (int j = 0; j < B; ++j) {
        w[j].nearest = res[j][0];
        w[j].d_near = dist6(w[j].rnd, nodes[w[j].nearest].pos);
        steer(nodes[w[j].nearest].pos, w[j].rnd, cfg.sampling_dist, w[j].np);
      }
    }
    lap(1);
    // ---- 3. new pose + parent edge (:149-151)
    {
      std::vector<double> p((size_t)B * 6), a((size_t)B * 6);
      for (int j = 0; j < B; ++j) { memcpy(&p[6 * (size_t)j], w[j].np, 48); memcpy(&a[6 * (size_t)j], nodes[w[j].nearest].pos, 48); }
      std::vector<uint8_t> hit(B), fr(B);
      std::vector<int32_t> fh(B), ns(B);
      c.collide_poses(p.data(), B, hit.data());
      c.collide_segments(a.data(), p.data(), B, fr.data(), fh.data(), ns.data());
      for (int j = 0; j < B; ++j) {
        w[j].pose_hit = hit[j] != 0;
        w[j].par_free = fr[j] != 0;
        w[j].par_fh = fh[j];
        w[j].par_ns = ns[j];
      }
    }
    lap(2);
    for (int j = 0; j < B; ++j)
      if (!w[j].pose_hit && w[j].par_free) alive.push_back(j);
    nA = (int)alive.size();
    // ---- 4. RRT*: k_max nearest store nodes around every surviving new point (:166)
    if (kmax > 0 && nA > 0) {
      std::vector<double> q((size_t)nA * 6);
      std::vector<int32_t> tq(nA);
      for (int k = 0; k < nA; ++k) { memcpy(&q[6 * (size_t)k], w[alive[k]].np, 48); tq[k] = w[alive[k]].tree; }
      std::vector<std::vector<int>> res;
      const uint64_t keep = st.nn_queries;
      knn(q.data(), nA, tq.data(), kmax, res);
      st.nn_queries = keep;
      for (int k = 0; k < nA; ++k) w[alive[k]].members = res[k];
    }
    lap(3);
  }
  // ---- 5. other trees: every node within treeDistance of the new point; per tree the nearest one (:228-231)
  auto note_conn = [&](WCand& cd, int id, double d) {
    const int t = nodes[id].tree;
    if (t == cd.tree) return;
    for (auto& cn : cd.conns)
      if (cn.tree == t) {
        if (d < cn.d || (d == cn.d && nodes[id].idx_in_tree < cn.order)) { cn.node = id; cn.d = d; cn.order = nodes[id].idx_in_tree; }
        return;
      }
    cd.conns.push_back({t, id, d, nodes[id].idx_in_tree, false, -1, 0});
  };
  if (chained && conn_have)
    for (int k = 0; k < nA && conn_have; ++k)
      if (conn_c[alive[k]] > conn_cap) conn_have = false;   // (a list ran over: the separate query below, with its growing lists)
  if (nA > 0 && tree_frontier.size() > 1 && chained && conn_have) {
    for (int k = 0; k < nA; ++k) {
      const int j = alive[k];
      for (int h = 0; h < conn_c[j]; ++h) note_conn(w[j], conn_i[(size_t)j * conn_cap + h], conn_dd[(size_t)j * conn_cap + h]);
    }
  } else if (nA > 0 && tree_frontier.size() > 1) {
    std::vector<double> q((size_t)nA * 6), rr(nA, cfg.dist_tree);
    for (int k = 0; k < nA; ++k) memcpy(&q[6 * (size_t)k], w[alive[k]].np, 48);
    int cap = 256;
    std::vector<int32_t> idx, cnt(nA);
    std::vector<double> dd;
    for (int tries = 0; tries < 6; ++tries) {
      idx.assign((size_t)nA * cap, -1);
      dd.assign((size_t)nA * cap, 0.0);
      c.radius(q.data(), nA, rr.data(), nullptr, nullptr, idx.data(), dd.data(), cnt.data(), cap);
      int mx = 0;
      for (int k = 0; k < nA; ++k) mx = std::max(mx, (int)cnt[k]);
      if (mx <= cap) break;
      cap = mx + 64;
    }
    for (int k = 0; k < nA; ++k) {
      WCand& cd = w[alive[k]];
      for (int h = 0; h < cnt[k]; ++h) note_conn(cd, idx[(size_t)k * cap + h], dd[(size_t)k * cap + h]);
    }
  }
  lap(4);
  // ---- 6. all remaining edges in one launch: RRT* member edges in both directions (store members and the
  // earlier new points of the same tree that may enter the k-nearest set), links to other trees (store
  // nearest and earlier new points of other trees within treeDistance)
  // (a chained wave: the new points are still on the device - an edge travels as two references, a store node's id or
  // -1 - j for the wave's j-th new point, 8 bytes instead of 96)
  std::vector<double> ea, eb;
  std::vector<int32_t> ra, rb;
  auto add_edge = [&](const double* a, const double* b, int ref_a, int ref_b) {
    if (chained) { ra.push_back(ref_a); rb.push_back(ref_b); return (int)ra.size() - 1; }
    int id = (int)(ea.size() / 6);
    ea.insert(ea.end(), a, a + 6);
    eb.insert(eb.end(), b, b + 6);
    return id;
  };
  struct Ref { int cand, kind, idx; };   // kind 0 = member fwd, 1 = member bwd, 2 = store conn, 3 = mate conn
  std::vector<Ref> refs;
  struct MateConn { int cand, mate; int edge; };
  std::vector<MateConn> mate_conns;
  {   // (room for everything up front; the surviving points' x and tree side by side for the two mate scans)
    const size_t guess = (size_t)nA * (size_t)(2 * std::max(kmax, 0) + 4) + 16;
    refs.reserve(guess);
    if (chained) { ra.reserve(guess); rb.reserve(guess); } else { ea.reserve(6 * guess); eb.reserve(6 * guess); }
  }
  std::vector<double> ax(nA);
  std::vector<int> at(nA), as(nA);
  for (int k = 0; k < nA; ++k) { ax[k] = w[alive[k]].np[0]; at[k] = w[alive[k]].tree; as[k] = w[alive[k]].slot; }
  const bool several_trees = tree_frontier.size() > 1;
  // RRT*: the earlier new points that may enter a point's k-nearest set are looked up in a coarse xyz grid that fills as the
  // rows are walked (a wave of hundreds of rows: every earlier row against every later one was most of this section)
  const int G = 16;
  double g_lo[3], g_inv[3];
  for (int a = 0; a < 3; ++a) {
    g_lo[a] = cfg.limits[2 * a];
    const double ext = cfg.limits[2 * a + 1] - cfg.limits[2 * a];
    g_inv[a] = ext > 0 ? G / ext : 0.0;
  }
  auto g_cell = [&](double v, int a) { const int cidx = (int)std::floor((v - g_lo[a]) * g_inv[a]); return cidx < 0 ? 0 : cidx >= G ? G - 1 : cidx; };
  std::vector<int> g_head, g_next, g_found;
  if (kmax > 0) { g_head.assign((size_t)G * G * G, -1); g_next.assign(nA, -1); }
  auto build_rows = [&](int k_first, int k_end) {
  for (int k = k_first; k < k_end; ++k) {
    const int j = alive[k];
    WCand& cd = w[j];
    cd.apos = k;
    if (kmax > 0) {
      // distance of the k_max-th store member bounds which mates can enter the set
      double dk = std::numeric_limits<double>::infinity();
      if ((int)cd.members.size() >= kmax) dk = dist6(cd.np, nodes[cd.members[kmax - 1]].pos);
      cd.medges.reserve(cd.members.size() + 8);
      for (int id : cd.members) cd.medges.push_back({id, false, false, -1, 0, -1, 0});
      int c0[3], c1[3];
      for (int a = 0; a < 3; ++a) {
        c0[a] = std::isfinite(dk) ? g_cell(cd.np[a] - dk, a) : 0;
        c1[a] = std::isfinite(dk) ? g_cell(cd.np[a] + dk, a) : G - 1;
      }
      g_found.clear();
      for (int cx = c0[0]; cx <= c1[0]; ++cx)
        for (int cy = c0[1]; cy <= c1[1]; ++cy)
          for (int cz = c0[2]; cz <= c1[2]; ++cz)
            for (int kk = g_head[((size_t)cx * G + cy) * G + cz]; kk >= 0; kk = g_next[kk]) {
              if (at[kk] != cd.tree || as[kk] == cd.slot) continue;   // (not the slot's own other row)
              if (dist6(cd.np, w[alive[kk]].np) <= dk) g_found.push_back(kk);
            }
      std::sort(g_found.begin(), g_found.end());
      for (int kk : g_found) cd.medges.push_back({-1 - alive[kk], false, false, -1, 0, -1, 0});
      {
        const size_t cell = ((size_t)g_cell(cd.np[0], 0) * G + g_cell(cd.np[1], 1)) * G + g_cell(cd.np[2], 2);
        g_next[k] = g_head[cell];
        g_head[cell] = k;
      }
      for (size_t e = 0; e < cd.medges.size(); ++e) {
        const double* op = cd.medges[e].other >= 0 ? nodes[cd.medges[e].other].pos : w[-1 - cd.medges[e].other].np;
        const int oref = cd.medges[e].other;   // (a store node's id, or -1 - i: the same encoding)
        refs.push_back({j, 0, (int)e});
        add_edge(cd.np, op, -1 - j, oref);          // isPathFree(newPoint, neighbor)  :172
        refs.push_back({j, 1, (int)e});
        add_edge(op, cd.np, oref, -1 - j);          // isPathFree(neighbor, newPoint)  :184
      }
    }
    for (size_t e = 0; e < cd.conns.size(); ++e) {
      refs.push_back({j, 2, (int)e});
      add_edge(cd.np, nodes[cd.conns[e].node].pos, -1 - j, cd.conns[e].node);   // isPathFree(newPoint, neighbor)  :231
    }
    for (int kk = 0; several_trees && kk < k; ++kk) {
      if (at[kk] == cd.tree || std::fabs(ax[kk] - cd.np[0]) >= cfg.dist_tree || as[kk] == cd.slot) continue;
      const int i = alive[kk];
      if (dist6(w[i].np, cd.np) < cfg.dist_tree) {
        mate_conns.push_back({j, i, add_edge(cd.np, w[i].np, -1 - j, -1 - i)});
        refs.push_back({j, 3, (int)mate_conns.size() - 1});
      }
    }
  }
  };
  // RRT* waves of some size: the edges go to the GPU in batches - the later batches' lists are built and the earlier
  // batches' rows replayed while the GPU checks the others (Ctx::seg_refs_begin / _end)
  // (two: a batch is a chain of eight launches and copies, ~80 us whatever its size - four batches measured no better)
  const int n_parts = chained && split_parts > 1 && kmax > 0 && nA >= 64 ? std::min(split_parts, 4) : 1;
  int part_k[5], part_e[5];          // rows / edges where the batches start
  for (int q = 0; q <= n_parts; ++q) part_k[q] = (int)((long long)nA * q / n_parts);
  part_e[0] = 0;
  for (int q = 0; q < n_parts; ++q) {
    build_rows(part_k[q], part_k[q + 1]);
    part_e[q + 1] = chained ? (int)ra.size() : (int)(ea.size() / 6);
    if (n_parts > 1) c.seg_refs_begin(q, ra.data() + part_e[q], rb.data() + part_e[q], part_e[q + 1] - part_e[q], (n_parts + 1) * (part_e[1] + 1024));
  }
  lap(8);
  const int nE = part_e[n_parts];
  std::vector<uint8_t> efr(nE);
  std::vector<int32_t> efh(nE), ens(nE);
  auto deal = [&](int e0, int e1) {   // the edges' answers -> the rows' lists
    for (int e = e0; e < e1; ++e) {
      const Ref& r = refs[e];
      WCand& cd = w[r.cand];
      if (r.kind == 0) { cd.medges[r.idx].free_f = efr[e] != 0; cd.medges[r.idx].fh_f = efh[e]; cd.medges[r.idx].ns_f = ens[e]; }
      else if (r.kind == 1) { cd.medges[r.idx].free_b = efr[e] != 0; cd.medges[r.idx].fh_b = efh[e]; cd.medges[r.idx].ns_b = ens[e]; }
      else if (r.kind == 2) { cd.conns[r.idx].free = efr[e] != 0; cd.conns[r.idx].fh = efh[e]; cd.conns[r.idx].ns = ens[e]; }
    }
  };
  int parts_home = 0;                // batches whose answers have been dealt out
  auto take_parts = [&](int upto) {  // ... up to and including batch `upto`
    for (; parts_home <= upto && parts_home < n_parts; ++parts_home) {
      const int e0 = part_e[parts_home], e1 = part_e[parts_home + 1];
      c.seg_refs_end(parts_home, efr.data() + e0, efh.data() + e0, ens.data() + e0);
      deal(e0, e1);
    }
  };
  if (n_parts > 1) {
    take_parts(0);
    lap(9);
  } else {
    if (nE && chained) c.collide_segments_refs(ra.data(), rb.data(), nE, efr.data(), efh.data(), ens.data());
    else if (nE) c.collide_segments(ea.data(), eb.data(), nE, efr.data(), efh.data(), ens.data());
    lap(9);
    deal(0, nE);
    parts_home = 1;
  }
  lap(5);
  // ---- 7. replay in order; cut the wave at the first iteration the speculation does not cover
  defer_append = true;
  int done = 0;
  bool merged = false;
  std::vector<int> acc;   // wave indices of the iterations that became nodes
  struct KN { double d; int order; int node; const WCand::Edge* e; };
  std::vector<KN> kn;
  kn.reserve(128);
  std::vector<int> acc_alt;   // the repaired rows among them (their points are not what k_rrt_mates saw)
  for (int j = 0; j < B && !merged && !solved; ++j) {
    // would a node added earlier in this wave have been the nearest neighbour? (ties go to the older node)
    int row = j;
    bool conflict = false;
    if (have_mates) {
      // the device named the nearest earlier new point that looked alive (mate): the slot's repaired row stands if that point
      // became a node as speculated; no repaired node of the wave may be as near as the row's nearest node
      if (w[j].cut_here) { ++g_rrt_alt[2]; break; }
      const int cand = mate[j];
      if (cand >= 0) {
        if (w[cand].accepted < 0 || w[j].alt_row < 0) { ++g_rrt_alt[2]; break; }
        row = w[j].alt_row;
        ++g_rrt_alt[1];
      }
      const WCand& cr = w[row];
      for (int i : acc_alt) {
        if (w[i].tree != cr.tree) continue;
        if (std::fabs(w[i].np[0] - cr.rnd[0]) > cr.d_near) continue;
        if (dist6(cr.rnd, w[i].np) <= cr.d_near) { conflict = true; break; }
      }
    } else {
      const WCand& cr = w[j];
      for (int i : acc) {
        if (w[i].tree != cr.tree) continue;
        if (std::fabs(w[i].np[0] - cr.rnd[0]) >= cr.d_near) continue;
        if (dist6(cr.rnd, w[i].np) < cr.d_near) { conflict = true; break; }
      }
    }
    if (conflict) break;
    WCand& cd = w[row];
    if (have_mates && !cd.prepared && !cd.pose_hit && cd.par_free) { ++g_rrt_alt[2]; break; }   // (the dry walk stopped before it)
    while (parts_home < n_parts && cd.apos >= part_k[parts_home]) take_parts(parts_home);
    ++done;
    ++iter;
    const unsigned iteration = (unsigned)iter;
    st.nn_queries += 1;                                                          // :143
    st.collide_calls += 1;                                                       // :149
    if (cd.pose_hit) continue;
    st.path_free_calls += 1;
    st.collide_calls += seg_calls(cd.par_fh, cd.par_ns);
    if (!cd.par_free) continue;
    int tree_to_expand = cd.tree;
    int nearest = cd.near_row >= 0 ? w[cd.near_row].accepted : cd.nearest;
    int new_id;
    if (cfg.optimize) {                                                          // :156-201
      double best = dist6(cd.np, nodes[nearest].pos) + nodes[nearest].d_root;
      const size_t krrt = (size_t)(2 * M_E * std::log10((double)nodes.size() + (cfg.lazy_edge ? 1.0 : 0.0)));
      st.nn_queries += 1;
      kn.clear();
      for (const WCand::Edge& e : cd.medges) {
        int nd = e.other >= 0 ? e.other : w[-1 - e.other].accepted;
        if (nd < 0) continue;
        kn.push_back({dist6(cd.np, nodes[nd].pos), nodes[nd].idx_in_tree, nd, &e});
      }
      std::sort(kn.begin(), kn.end(), [](const KN& a, const KN& b) { return a.d < b.d || (a.d == b.d && a.order < b.order); });
      if (kn.size() > krrt) kn.resize(krrt);
      if (kn.size() < std::min(krrt, trees[tree_to_expand].size()))
        throw HipError{"rrt: k-nearest candidate set incomplete (internal error)"};
      for (const KN& x : kn) {                                                   // :168-175
        double nd = x.d + nodes[x.node].d_root;   // (x.d = dist6(cd.np, the node))
        if (nd < best - SFFG_TOL) {
          st.path_free_calls += 1;
          st.collide_calls += seg_calls(x.e->fh_f, x.e->ns_f);
          if (x.e->free_f) { best = nd; nearest = x.node; }
        }
      }
      new_id = add_node(cd.np, nodes[nearest].root_tree, tree_to_expand, nearest, dist6(nodes[nearest].pos, cd.np), best, iteration);
      for (const KN& x : kn) {                                                   // :181-201
        double npd = dist6(nodes[x.node].pos, cd.np);
        double proposed = best + npd;
        if (proposed < nodes[x.node].d_root - SFFG_TOL) {
          st.path_free_calls += 1;
          st.collide_calls += seg_calls(x.e->fh_b, x.e->ns_b);
          if (x.e->free_b) {
            nodes[x.node].parent = new_id;
            nodes[x.node].root_tree = nodes[new_id].root_tree;
            nodes[x.node].d_closest = npd;
            nodes[x.node].d_root = proposed;
          }
        }
      }
    } else {                                                                     // :203
      new_id = add_node(cd.np, nodes[nearest].root_tree, tree_to_expand, nearest, cfg.sampling_dist,
                        nodes[nearest].d_root + cfg.sampling_dist, iteration);
    }
    cd.accepted = new_id;
    acc.push_back(row);
    if (row != j) acc_alt.push_back(row);
    if (cfg.lazy_edge) { lazy_goal_check(new_id); continue; }   // (solved ends the replay loop)
    // :219-319 links to the other live trees, in frontier order
    for (int i = 0; i < (int)tree_frontier.size(); ++i) {
      const int tree = tree_frontier[i];
      if (tree == tree_to_expand) continue;
      st.nn_queries += 1;
      // nearest node of that tree: the frozen tree's candidate vs the nodes it gained earlier in this wave
      int nb = -1, nb_order = 0, fh = -1, ns = 0;
      double nbd = 0;
      bool nb_free = false;
      for (const WCand::Conn& cn : cd.conns)
        if (cn.tree == tree) { nb = cn.node; nbd = cn.d; nb_order = cn.order; nb_free = cn.free; fh = cn.fh; ns = cn.ns; }
      for (const MateConn& mc : mate_conns) {
        if (mc.cand != row || w[mc.mate].tree != tree || w[mc.mate].accepted < 0) continue;
        const int id = w[mc.mate].accepted;
        const double d = dist6(nodes[id].pos, cd.np);
        if (nb < 0 || d < nbd || (d == nbd && nodes[id].idx_in_tree < nb_order)) {
          nb = id; nbd = d; nb_order = nodes[id].idx_in_tree; nb_free = efr[mc.edge] != 0; fh = efh[mc.edge]; ns = ens[mc.edge];
        }
      }
      if (nb < 0) continue;                        // nothing of that tree within treeDistance: :231 is false
      if (!(dist6(nodes[nb].pos, cd.np) < cfg.dist_tree)) continue;
      st.path_free_calls += 1;
      st.collide_calls += seg_calls(fh, ns);
      if (!nb_free) continue;
      tree_to_expand = merge_or_link(tree_to_expand, new_id, nb, true, 0, 0, i);
      merged = true;                               // tree ids / frontier changed: later picks are stale
    }
  }
  take_parts(n_parts - 1);   // (a replay that ended early: the batches still have to come home before their buffers are used again)
  lap(6);
  // ---- 8. commit: device store, RNG position
  if (!pend_tree.empty()) {
    // (after a chained wave nothing touches the staging buffer before the next wave's chain has waited for the stream)
    c.store_append(pend_pos.data(), pend_tree.data(), (int)pend_tree.size(), /*wait=*/!chained);
    pend_pos.clear();
    pend_tree.clear();
  }
  defer_append = false;
  if (done < B) {
    const uint64_t target = w[done].draws_before;
    rng = snapshot;
    while (rng.draws < target) rng.next();
  } else if (rng.draws != draws_end) {
    throw HipError{"rrt: RNG bookkeeping error"};
  }
  lap(7);
  st.waves += 1;
  st.speculated += (uint64_t)B;
  st.committed += (uint64_t)done;
  return done;
}

void Rrt::run(int max_iters) {
  auto t0 = std::chrono::steady_clock::now();
  int done = 0;
  int B = 1;
  while (!(solved || iter == cfg.max_iterations)) {                             // :93
    if (max_iters > 0 && done >= max_iters) break;
    if (cfg.wave == 1) {
      ++done;
      ++iter;
      int tree = cfg.lazy_edge ? 0 : tree_frontier[rng.uniform_int(0, num_trees)];   // :95
      expand(tree, (unsigned)iter);
      continue;
    }
    // wave size: fixed by the caller, or adapted to how much of the last wave survived
    int want = cfg.wave > 1 ? cfg.wave : B;
    if (max_iters > 0) want = std::min(want, max_iters - done);
    const int got = run_wave(std::max(1, want));
    done += got;
    if (cfg.wave <= 0) {
      if (got >= want) B = std::min(4096, B * 2);
      else {
        // (a cut wave: the next one speculates grow_pct % of what survived; a SMALL wave costs the chain's latency whatever
        // its size, so it may as well carry a few times that)
        const int by_rule = (int)((long long)got * grow_pct / 100) + 1;
        B = std::max(1, std::min(4096, std::max(by_rule, std::min(small_mul * got + 1, small_cap))));
      }
    }
    if (got == 0 && want > 0 && iter >= cfg.max_iterations) break;
  }
  if (chain_on) ctx->sync();   // (the last wave's append)
  st.total_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (getenv("SFFGPU_PROFILE")) {
    fprintf(stderr, "[sffgpu rrt run_wave ms] draws %.1f | chain %.1f | repaired slots %.1f | k nearest %.1f | other trees %.1f | edge lists %.1f + edges on the GPU %.1f + results %.1f | replay %.1f | append %.1f  (%llu waves)\n",
            g_rrt_sec[0], g_rrt_sec[1], g_rrt_sec[2], g_rrt_sec[3], g_rrt_sec[4], g_rrt_sec[8], g_rrt_sec[9], g_rrt_sec[5], g_rrt_sec[6], g_rrt_sec[7], (unsigned long long)st.waves);
    fprintf(stderr, "[sffgpu rrt repaired slots] %llu evaluated, %llu taken, %llu waves cut at a slot without one\n", g_rrt_alt[0], g_rrt_alt[1], g_rrt_alt[2]);
    for (double& x : g_rrt_sec) x = 0;
    for (auto& x : g_rrt_alt) x = 0;
  }
}

// RapidExpTree::getConnectedTrees (src/rrt.h:381-393) + getPaths (:324-352)
// RapidExpTree::smoothPaths (src/rrt.h:354-379): walk every link plan from its far end (index g) and connect it
// to the EARLIEST node t < g-1 whose straight edge is free, dropping the nodes in between.  The reference
// tests t = 0, 1, ... one isPathFree at a time; here all candidate edges of one g go to the GPU in one batch and
// the first free one is taken (isPathFree is pure).  It shortens the plans stored in the central tree's links;
// neighboringMatrix holds copies made earlier (:350) and every writer reads the matrix, so - as in the
// reference - the saved paths and costs do not change.
void Rrt::smooth_paths() {
  for (std::vector<int>& plan : link_plans) {
    int g = (int)plan.size() - 1;
    while (g > 0) {
      const int m = g - 1;   // candidates t = 0 .. g-2
      std::vector<uint8_t> fr(std::max(m, 0));
      std::vector<int32_t> fh(std::max(m, 0)), nsv(std::max(m, 0));
      if (m > 0) {
        std::vector<double> a((size_t)m * 6), b((size_t)m * 6);
        for (int t = 0; t < m; ++t) {
          memcpy(&a[6 * (size_t)t], nodes[plan[t]].pos, 48);
          memcpy(&b[6 * (size_t)t], nodes[plan[g]].pos, 48);
        }
        ctx->collide_segments(a.data(), b.data(), m, fr.data(), fh.data(), nsv.data());
      }
      int t = 0;
      bool changed = false;
      while (t < g - 1) {
        st.path_free_calls += 1;
        st.collide_calls += fh[t] > 0 ? (uint64_t)fh[t] : (uint64_t)nsv[t];
        if (fr[t]) { changed = true; break; }
        ++t;
      }
      if (changed) plan.erase(plan.begin() + t + 1, plan.begin() + g);
      g = t;
    }
  }
}

void Rrt::get_paths() {
  const int nt = (int)trees.size();
  nm.assign((size_t)nt * nt, PathHolder());
  connected.clear();
  size_t max_conn = 0;
  int central = 0;
  const int num_roots = cfg.has_goal ? num_trees + 2 : num_trees + 1;   // :384 (numTrees was decremented per merge)
  for (int i = 0; i < num_roots && i < nt; ++i)
    if (eaten[i].size() > max_conn) {
      max_conn = eaten[i].size();
      central = i;
      connected = eaten[i];
      connected.push_back(i);
    }
  link_plans.clear();
  for (const RLink& link : links[central]) {
    PathHolder h;
    h.n1 = link.n1;
    h.n2 = link.n2;
    h.dist = link.dist;
    std::vector<int> chain;
    for (int n = link.n1;; n = nodes[n].parent) {   // push_front up to the root (IsRoot: DistanceToRoot == 0)
      chain.push_back(n);
      if (nodes[n].d_root == 0) break;
    }
    h.plan.assign(chain.rbegin(), chain.rend());
    for (int n = link.n2;; n = nodes[n].parent) {
      h.plan.push_back(n);
      if (nodes[n].d_root == 0) break;
    }
    const int a = nodes[link.n1].root_tree, b = nodes[link.n2].root_tree;   // :350 neighboringMatrix(Root ids)
    nm[(size_t)std::min(a, b) * nt + std::max(a, b)] = h;
    link_plans.push_back(h.plan);   // the link's own DistanceHolder::plan; the matrix keeps a copy (:350)
  }
  // Solver::getAllPaths (src/problemStruct.h:184-253), called right after getPaths (src/rrt.h:106-107)
  auto NM = [&](int i, int j) -> PathHolder& { return nm[(size_t)std::min(i, j) * nt + std::max(i, j)]; };
  const int nc = (int)connected.size();
  for (int k = 0; k < nc; ++k) {
    const int id3 = connected[k];
    for (int i = 0; i < nc; ++i) {
      const int id1 = connected[i];
      if (i == k || NM(id1, id3).n1 < 0) continue;
      for (int j = 0; j < nc; ++j) {
        const int id2 = connected[j];
        if (i == j || NM(id2, id3).n1 < 0) continue;
        const PathHolder h1 = NM(id1, id3), h2 = NM(id2, id3);
        std::vector<int> plan1 = h1.plan, plan2 = h2.plan;
        int node1, node2;
        if (nodes[h1.n1].root_tree == id1) node1 = h1.n1; else { node1 = h1.n2; std::reverse(plan1.begin(), plan1.end()); }
        if (nodes[h2.n1].root_tree == id2) node2 = h2.n1; else { node2 = h2.n2; std::reverse(plan2.begin(), plan2.end()); }
        int last = -1;
        while (!plan1.empty() && !plan2.empty() && plan1.back() == plan2.back()) {
          last = plan1.back();
          plan1.pop_back();
          plan2.pop_back();
        }
        std::vector<int> fin(plan1.begin(), plan1.end());
        fin.push_back(last);
        fin.insert(fin.end(), plan2.rbegin(), plan2.rend());
        double d = 0;
        for (size_t q = 1; q < fin.size(); ++q) d += sffg::dist6(nodes[fin[q - 1]].pos, nodes[fin[q]].pos);
        if (d < NM(id1, id2).dist - SFFG_TOL) {
          PathHolder h;
          h.dist = d;
          if (node1 < node2) { h.n1 = node1; h.n2 = node2; h.plan = fin; }
          else { h.n1 = node2; h.n2 = node1; h.plan.assign(fin.rbegin(), fin.rend()); }
          NM(id1, id2) = h;
        }
      }
    }
  }
}

}  // namespace sff
