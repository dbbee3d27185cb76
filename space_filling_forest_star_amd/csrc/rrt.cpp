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

#include "engine.h"
#include "sff_geom.h"

namespace sff {

#define HIPCHK(x) hip_check((x), #x)

Rrt::Rrt(Ctx* c, const sffgpu_rrt_cfg& cf, const double* roots6, int n_roots) : ctx(c), cfg(cf) {
  if (cfg.dim != 2 && cfg.dim != 6) throw HipError{"rrt: dim must be 2 or 6"};
  if (n_roots < 1) throw HipError{"rrt: at least one root"};
  if (cfg.priority_bias != 0 && !cfg.has_goal) throw HipError{"rrt: goal bias needs a goal (src/main.cpp:330-331)"};
  if (!c->have_env || !c->have_robot) throw HipError{"rrt: upload ENV and ROBOT meshes first"};
  rng.reseed(cfg.seed);
  const int nt = n_roots + (cfg.has_goal ? 1 : 0);
  trees.resize(nt);
  links.resize(nt);
  eaten.resize(nt);
  ctx->store_reset(std::max(4096, cfg.max_iterations + nt + 16));
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
  ctx->store_append(pos, &t, 1);                // replaces flannIndex->addPoints (:215)
  return id;
}

RLink Rrt::make_link(int a, int b) {            // DistanceHolder(first, second), src/primitives.h:609-618
  double d = nodes[a].d_root + nodes[b].d_root + sffg::dist6(nodes[a].pos, nodes[b].pos);
  return {std::min(a, b), std::max(a, b), d};
}

// k nearest of one tree in the reference's order (distance, index in the tree's list)
void Rrt::knn(const double* q, int nq, const int32_t* tree, int k, std::vector<std::vector<int>>& out) {
  std::vector<int32_t> idx((size_t)nq * k), cnt(nq);
  std::vector<double> dist((size_t)nq * k);
  ctx->knn(q, nq, k, tree, nullptr, idx.data(), dist.data(), cnt.data());
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

void Rrt::expand(int tree_to_expand, unsigned iteration) {
  using namespace sffg;
  double rnd[6], np[6];
  if (cfg.priority_bias != 0 && uniform_real(rng.next(), 0.0, 1.0) <= cfg.priority_bias) {   // :130-131
    memcpy(rnd, nodes[goal_node].pos, sizeof rnd);
  } else {
    // RandGen::randomPointInSpace (src/randGen.h:124-146); Y is drawn before X (g++ evaluates the
    // two arguments of point.set(...) right to left — pinned by tests/golden/ref_primitives.json)
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
    const size_t krrt = (size_t)(2 * M_E * std::log10((double)nodes.size()));
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
  }
}

void Rrt::run(int max_iters) {
  auto t0 = std::chrono::steady_clock::now();
  int done = 0;
  while (!(solved || iter == cfg.max_iterations)) {                             // :93
    if (max_iters > 0 && done >= max_iters) break;
    ++done;
    ++iter;
    int tree = tree_frontier[rng.uniform_int(0, num_trees)];                    // :95
    expand(tree, (unsigned)iter);
  }
  st.total_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace sff
