// forest.cpp — wave-parallel SpaceForest (SFF) engine of libsffgpu.
//
// Reference: SpaceForest<T,R> — constructor src/forest.h:57-110, Solve() :113-202, expandNode
// :240-376, maxConnected :379-418.  The reference expands ONE frontier node per outer iteration;
// this engine draws `wave` frontier slots at once, evaluates every slot's sample on the GPU
// against the frozen node store (+ the earlier samples of the same round), and then replays
// the reference's accept/reject logic on the host in slot order, looking the collision and
// neighbour answers up instead of computing them.  The result is, by construction, what the
// reference loop would produce if it ran the same slots one after another; with wave == 1 it is
// the reference loop itself (same RNG consumption order).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>

#include "engine.h"
#include "sff_geom.h"

namespace sff {

#define HIPCHK(x) hip_check((x), #x)
using Clock = std::chrono::steady_clock;
static double ms_since(Clock::time_point t0) {
  return std::chrono::duration<double, std::milli>(Clock::now() - t0).count();
}

Forest::Forest(Ctx* c, const sffgpu_forest_cfg& cf, const double* roots6, int n_roots) : ctx(c), cfg(cf) {
  if (cfg.wave < 1) cfg.wave = 1;
  if (cfg.dim != 2 && cfg.dim != 6) throw HipError{"forest: dim must be 2 or 6"};
  if (cfg.has_goal) throw HipError{"forest: single-goal mode (Problem::hasGoal) is not implemented on the GPU path yet"};
  if (cfg.world < 1) cfg.world = 1;
  if (cfg.rank < 0 || cfg.rank >= cfg.world) throw HipError{"forest: rank outside [0, world)"};
  if (n_roots < 1) throw HipError{"forest: at least one root"};
  if (!c->have_env || !c->have_robot) throw HipError{"forest: upload ENV and ROBOT meshes first"};
  rng.reseed(cfg.seed);
  num_roots = n_roots;
  trees.resize(num_roots);
  ctx->store_reset(std::max(cfg.node_budget, 4096) + cfg.wave + 64);
  std::vector<int32_t> tids(n_roots);
  for (int j = 0; j < n_roots; ++j) {          // src/forest.h:60-76
    int id = add_node(roots6 + 6 * (size_t)j, j, -1, 0, 0, 0);
    frontier.push_back(id);
    tids[j] = j;
  }
  ctx->store_append(roots6, tids.data(), n_roots);
  memset(&st, 0, sizeof st);
  knn_r = 2.5 * cfg.sampling_dist;
}

int Forest::add_node(const double* pos, int tree, int parent, double dclosest, double droot, unsigned it) {
  FNode n;
  memcpy(n.pos, pos, sizeof n.pos);
  n.tree = tree;
  n.parent = parent;
  n.d_closest = dclosest;
  n.d_root = droot;
  n.iter = it;
  n.idx_in_tree = (int)trees[tree].size();
  int id = (int)nodes.size();
  nodes.push_back(n);
  trees[tree].push_back(id);
  return id;
}

std::vector<Border>& Forest::border(int i, int j) {  // SymmetricMatrix, src/primitives.h:572-596
  if (i > j) std::swap(i, j);
  return borders[{i, j}];
}

// src/forest.h:379-418
int Forest::max_connected() {
  int max_conn = 0, remaining = num_roots;
  std::vector<char> conn(num_roots, 0);
  int unconnected = 0;
  while (max_conn < remaining) {
    connected.clear();
    std::vector<int> stack{unconnected};
    conn[unconnected] = 1;
    while (!stack.empty()) {
      int root = stack.front();
      stack.erase(stack.begin());
      connected.push_back(root);
      for (int i = 0; i < num_roots; ++i) {
        if (root == i) continue;
        auto it = borders.find({std::min(root, i), std::max(root, i)});
        if (it != borders.end() && !it->second.empty() && !conn[i]) {
          conn[i] = 1;
          stack.insert(stack.begin(), i);
        }
      }
    }
    max_conn = (int)connected.size();
    for (int i = 0; i < num_roots; ++i)
      if (!conn[i]) { unconnected = i; break; }
    remaining -= max_conn;
  }
  return max_conn;
}

// node selection for every slot of the wave, src/forest.h:136-151 (non-priority mode)
void Forest::begin_wave() {
  slots.clear();
  const bool use_closed = !closed.empty() && empty_frontier;
  const int pool = use_closed ? (int)closed.size() : (int)frontier.size();
  const int n_slots = std::max(1, std::min(cfg.wave, pool));
  for (int s = 0; s < n_slots; ++s) {
    Slot sl;
    if (use_closed) {
      sl.node = closed[rng.uniform_int(0, (int)closed.size() - 1)];
      sl.from_closed = true;
    } else {
      sl.node = frontier[rng.uniform_int(0, (int)frontier.size() - 1)];
      sl.from_closed = false;
    }
    sl.failing = true;
    slots.push_back(sl);
  }
  round = 0;
  in_wave = true;
  ++st.waves;
}

// src/forest.h:160-201
void Forest::end_wave() {
  for (Slot& sl : slots) {
    if (sl.failing && !sl.from_closed) {
      auto it = std::find(frontier.begin(), frontier.end(), sl.node);
      if (it != frontier.end()) {
        frontier.erase(it);
        nodes[sl.node].force_children = true;
        closed.push_back(sl.node);
      }
    }
  }
  empty_frontier = frontier.empty();
  if (!solved) {
    bool conn = max_connected() == num_roots;
    solved = (!cfg.has_goal && empty_frontier && conn);
  } else {
    max_connected();
  }
  in_wave = false;
}

// ---------------------------------------------------------------------------------------
// A round in two phases.
//   round_begin : (replicated) pick the active slots, draw the engine words, sample ALL of them on
//                 this GPU; (sharded) sweep / classify / collision-check only the candidates this
//                 rank owns (candidate i -> rank i % world) and serialise their answers.
//   round_commit: (replicated) take the answers of all ranks, replay expandNode in slot order,
//                 append the accepted nodes to the host forest and the device store.
// With world == 1 the local answers are consumed directly.
// ---------------------------------------------------------------------------------------
static const int32_t REC_MAGIC = 0x53464652;  // "SFFR"

void Forest::round_begin() {
  Ctx& c = *ctx;
  HIPCHK(hipSetDevice(c.device));
  if (pending_round) throw HipError{"forest: round_begin called twice without round_commit"};
  if (!in_wave) {
    if (terminated()) return;
    begin_wave();
  }
  auto t_host = Clock::now();
  double wait_ms = 0;
  auto timed_sync = [&]() {
    auto t0 = Clock::now();
    c.sync();
    wait_ms += ms_since(t0);
  };
  // ---- active slots of this round (src/forest.h:155: i < ThresholdMisses && expandResult && iter < max)
  cands.clear();
  for (int s = 0; s < (int)slots.size(); ++s) {
    if (!slots[s].failing) continue;
    if (iter + (int)cands.size() >= cfg.max_iterations) break;
    Cand cd;
    cd.slot = s;
    cd.expanded = slots[s].node;
    cands.push_back(cd);
  }
  const int n = (int)cands.size();
  ++round;
  pending_round = true;
  records.clear();
  records.push_back(REC_MAGIC);
  records.push_back(n);
  if (n == 0) return;
  iter0 = iter;
  iter += n;
  N0 = (int)nodes.size();
  c.store_reserve(N0 + n + 4);

  // ---- draw the raw engine words in reference order; sample + steer every candidate
  const int words_per = cfg.dim == 2 ? 1 : 6;
  c.h_a.ensure((size_t)n * 6 * sizeof(uint64_t));
  c.h_b.ensure((size_t)n * sizeof(int32_t));
  uint64_t* hw = c.h_a.as<uint64_t>();
  int32_t* hp = c.h_b.as<int32_t>();
  for (int i = 0; i < n; ++i) {
    for (int k = 0; k < 6; ++k) hw[6 * (size_t)i + k] = k < words_per ? rng.next() : 0;
    hp[i] = cands[i].expanded;
  }
  const int CAP = 64;
  const size_t pb = (size_t)n * 6 * sizeof(double);
  c.d_a.ensure((size_t)n * 6 * sizeof(uint64_t));
  c.d_b.ensure((size_t)n * sizeof(int32_t));
  c.d_c.ensure(pb);                                   // new positions
  c.d_d.ensure((size_t)n);                            // in-limits
  c.d_e.ensure((size_t)n * sizeof(double));           // parent distance
  c.d_f.ensure((size_t)n * sizeof(sffk::SweepQuery)); // sweep queries
  c.d_g.ensure((size_t)n * sizeof(int32_t) + (size_t)n * CAP * sizeof(int32_t));  // cnt | hit idx
  c.d_h.ensure((size_t)n * CAP * sizeof(double));     // hit dist
  int32_t* d_cnt = c.d_g.as<int32_t>();
  int32_t* d_hidx = d_cnt + n;
  HIPCHK(hipMemcpyAsync(c.d_a.p, hw, (size_t)n * 6 * sizeof(uint64_t), hipMemcpyHostToDevice, c.stream));
  HIPCHK(hipMemcpyAsync(c.d_b.p, hp, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, c.stream));
  HIPCHK(hipMemsetAsync(d_cnt, 0, (size_t)n * sizeof(int32_t), c.stream));
  sffk::SampleParams prm{};
  memcpy(prm.limits, cfg.limits, sizeof prm.limits);
  prm.dist_tree = cfg.dist_tree;
  prm.sweep_abs_eps = c.sweep_eps();
  prm.rank = cfg.rank;
  prm.world = cfg.world;
  c.time_begin(T_SAMPLE);
  sffk::launch_sample_steer(c.stream, c.d_a.as<uint64_t>(), c.d_b.as<int32_t>(), c.spos.as<double>(), nullptr, n,
                            cfg.sampling_dist, cfg.dim, prm, c.d_c.as<double>(), c.d_d.as<uint8_t>(),
                            c.d_e.as<double>(), c.d_f.as<sffk::SweepQuery>(), N0);
  // the round's samples become temporary store entries [N0, N0+n) so that the same sweep also
  // finds, for every sample, the EARLIER samples of this round (query i sees ids < N0 + i)
  sffk::NodeStoreMut mut{c.sx.as<float>(), c.sy.as<float>(), c.sz.as<float>(), c.syaw.as<float>(),
                         c.spitch.as<float>(), c.sroll.as<float>(), c.stree.as<int32_t>(), c.spos.as<double>()};
  sffk::launch_store_write(c.stream, mut, c.d_c.as<double>(), nullptr, c.d_b.as<int32_t>(), c.d_d.as<uint8_t>(), n, N0);
  c.time_end();
  // the sweep only serves the queries of this rank's shard (the others are marked inactive)
  c.time_begin(T_SWEEP);
  sffk::launch_sweep(c.stream, c.store_view(), N0 + n, c.d_f.as<sffk::SweepQuery>(), c.d_c.as<double>(), n, d_cnt,
                     d_hidx, c.d_h.as<double>(), CAP);
  c.time_end();
  st.sweeps += 1;
  st.sweep_nodes += (uint64_t)(N0 + n);
  st.sweep_queries += (uint64_t)((n - cfg.rank + cfg.world - 1) / cfg.world);
  c.h_c.ensure(pb);
  c.h_d.ensure((size_t)n);
  c.h_e.ensure((size_t)n * sizeof(double));
  c.h_g.ensure((size_t)n * sizeof(int32_t) + (size_t)n * CAP * sizeof(int32_t));
  c.h_h.ensure((size_t)n * CAP * sizeof(double));
  HIPCHK(hipMemcpyAsync(c.h_c.p, c.d_c.p, pb, hipMemcpyDeviceToHost, c.stream));
  HIPCHK(hipMemcpyAsync(c.h_d.p, c.d_d.p, (size_t)n, hipMemcpyDeviceToHost, c.stream));
  HIPCHK(hipMemcpyAsync(c.h_e.p, c.d_e.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c.stream));
  HIPCHK(hipMemcpyAsync(c.h_g.p, c.d_g.p, (size_t)n * sizeof(int32_t) + (size_t)n * CAP * sizeof(int32_t),
                        hipMemcpyDeviceToHost, c.stream));
  HIPCHK(hipMemcpyAsync(c.h_h.p, c.d_h.p, (size_t)n * CAP * sizeof(double), hipMemcpyDeviceToHost, c.stream));
  timed_sync();

  const double* hpos = c.h_c.as<double>();
  const uint8_t* hlim = c.h_d.as<uint8_t>();
  const double* hpd = c.h_e.as<double>();
  const int32_t* hcnt = c.h_g.as<int32_t>();
  const int32_t* hidx = hcnt + n;
  const double* hdist = c.h_h.as<double>();
  auto mine_shard = [&](int i) { return i % cfg.world == cfg.rank; };

  // ---- classify neighbours, build the pose / edge task lists (own shard only)
  std::vector<double> pose_tasks, seg_a, seg_b;
  auto add_seg = [&](const double* a, const double* b) {
    int id = (int)(seg_a.size() / 6);
    seg_a.insert(seg_a.end(), a, a + 6);
    seg_b.insert(seg_b.end(), b, b + 6);
    return id;
  };
  std::vector<int> overflow_q;
  for (int i = 0; i < n; ++i) {
    Cand& cd = cands[i];
    memcpy(cd.pos, hpos + 6 * (size_t)i, sizeof cd.pos);
    cd.in_lim = hlim[i] != 0;
    cd.pdist = hpd[i];
    if (!cd.in_lim || !mine_shard(i)) continue;
    if (hcnt[i] > CAP) overflow_q.push_back(i);
  }
  // rare: a hit list overflowed -> redo those queries with a big list through the generic path
  std::vector<std::vector<std::pair<double, int>>> big(n);
  if (!overflow_q.empty()) {
    const int BIG = 4096;
    int m = (int)overflow_q.size();
    std::vector<double> q6((size_t)m * 6), rr(m);
    std::vector<int32_t> mx(m), cn(m), ix((size_t)m * BIG);
    std::vector<double> dd((size_t)m * BIG);
    std::vector<int32_t> hcnt_copy(hcnt, hcnt + n);
    for (int k = 0; k < m; ++k) {
      int i = overflow_q[k];
      memcpy(&q6[6 * (size_t)k], cands[i].pos, 6 * sizeof(double));
      rr[k] = std::max(cands[i].pdist, cfg.dist_tree);
      mx[k] = N0 + i;
    }
    int keep = c.store_n;
    c.store_n = N0 + n;  // include the temporaries
    c.radius(q6.data(), m, rr.data(), nullptr, mx.data(), ix.data(), dd.data(), cn.data(), BIG);
    c.store_n = keep;
    for (int k = 0; k < m; ++k) {
      if (cn[k] > BIG) throw HipError{"forest: neighbour list overflow (> 4096 hits)"};
      for (int j = 0; j < cn[k]; ++j) big[overflow_q[k]].push_back({dd[(size_t)k * BIG + j], ix[(size_t)k * BIG + j]});
    }
  }
  std::vector<char> overflowed(n, 0);
  for (int i : overflow_q) overflowed[i] = 1;
  for (int i = 0; i < n; ++i) {
    Cand& cd = cands[i];
    if (!cd.in_lim || !mine_shard(i)) continue;
    const FNode& ex = nodes[cd.expanded];
    cd.pose_task = (int)(pose_tasks.size() / 6);
    pose_tasks.insert(pose_tasks.end(), cd.pos, cd.pos + 6);
    cd.seg_parent = add_seg(ex.pos, cd.pos);
    const int mine = ex.tree;
    std::vector<Nb> all;
    auto consider = [&](double d, int id) {
      Nb nb;
      nb.d = d;
      if (id < N0) {
        nb.id = id;
        nb.tree = nodes[id].tree;
        nb.order = nodes[id].idx_in_tree;
      } else {
        int cc = id - N0;
        if (!cands[cc].in_lim) return;
        nb.id = -1 - cc;
        nb.tree = nodes[cands[cc].expanded].tree;
        nb.order = 0x40000000 + cc;
      }
      nb.same_tree = nb.tree == mine;
      nb.seg = -1;
      if (nb.same_tree) {
        if (ex.force_children || !(d < cd.pdist - SFFG_TOL)) return;   // src/forest.h:276
      } else {
        if (!(d < cfg.dist_tree - SFFG_TOL)) return;                    // src/forest.h:283
      }
      all.push_back(nb);
    };
    if (overflowed[i]) {
      for (auto& h : big[i]) consider(h.first, h.second);
    } else {
      const int32_t* hc = c.h_g.as<int32_t>();  // (buffers may have been re-allocated by the overflow path)
      const int32_t* hi = hc + n;
      const double* hd = c.h_h.as<double>();
      for (int k = 0; k < hc[i]; ++k) consider(hd[(size_t)i * CAP + k], hi[(size_t)i * CAP + k]);
    }
    std::sort(all.begin(), all.end(), [](const Nb& a, const Nb& b) {
      if (a.tree != b.tree) return a.tree < b.tree;
      if (a.d != b.d) return a.d < b.d;
      return a.order < b.order;
    });
    // everything after the first STORE neighbour of another tree is unreachable (:296-299)
    for (Nb& nb : all) {
      const double* npos = nb.id >= 0 ? nodes[nb.id].pos : cands[-1 - nb.id].pos;
      if (nb.same_tree) nb.seg = add_seg(npos, cd.pos);     // isPathFree(neighbour, newPoint)  :276
      else nb.seg = add_seg(ex.pos, npos);                  // isPathFree(expanded, neighbour)  :288
      cd.nbs.push_back(nb);
      if (!nb.same_tree && nb.id >= 0) break;
    }
  }
  (void)hidx; (void)hdist;

  // ---- collision launches
  const int n_pose = (int)(pose_tasks.size() / 6), n_seg = (int)(seg_a.size() / 6);
  std::vector<uint8_t> pose_hit(n_pose), seg_free(n_seg);
  std::vector<int32_t> seg_fh(n_seg), seg_ns(n_seg);
  if (n_pose) {
    auto t0 = Clock::now();
    c.collide_poses(pose_tasks.data(), n_pose, pose_hit.data());
    c.collide_segments(seg_a.data(), seg_b.data(), n_seg, seg_free.data(), seg_fh.data(), seg_ns.data());
    wait_ms += ms_since(t0);
  }
  st.poses_executed += n_pose;
  st.segments_executed += n_seg;
  for (int k = 0; k < n_seg; ++k) st.samples_executed += (uint64_t)seg_ns[k];

  // ---- answers of the owned candidates -> Cand fields
  for (int i = 0; i < n; ++i) {
    Cand& cd = cands[i];
    if (!cd.in_lim || !mine_shard(i)) continue;
    cd.answered = true;
    cd.pose_hit = pose_hit[cd.pose_task] != 0;
    cd.par_free = seg_free[cd.seg_parent] != 0;
    cd.par_fh = seg_fh[cd.seg_parent];
    cd.par_ns = seg_ns[cd.seg_parent];
    for (Nb& nb : cd.nbs) {
      nb.free = seg_free[nb.seg] != 0;
      nb.fh = seg_fh[nb.seg];
      nb.ns = seg_ns[nb.seg];
    }
  }

  // ---- SFF* (src/forest.h:307-351): candidates that no STORE neighbour rejects may be accepted;
  // for them fetch the potential k-nearest set of their tree and answer both edge directions
  if (cfg.optimize) {
    std::vector<int> maybe;
    for (int i = 0; i < n; ++i) {
      Cand& cd = cands[i];
      if (!cd.answered || cd.pose_hit || !cd.par_free) continue;
      bool rejected = false;
      for (const Nb& nb : cd.nbs) {
        if (nb.id < 0) continue;
        if (nb.same_tree) { if (nb.free) { rejected = true; break; } }
        else { rejected = true; break; }
      }
      if (!rejected) maybe.push_back(i);
    }
    const int m = (int)maybe.size();
    if (m) {
      auto t0 = Clock::now();
      const int KCAP = 256;
      std::vector<double> q6((size_t)m * 6), r(m, knn_r), lo(m, 0.0), hi(m, -1.0);
      std::vector<int32_t> qtree(m), qmax(m), kmax(m);
      std::vector<uint8_t> active(m, 1);
      std::vector<std::vector<HitRec>> lists(m);
      for (int k = 0; k < m; ++k) {
        const Cand& cd = cands[maybe[k]];
        memcpy(&q6[6 * (size_t)k], cd.pos, 6 * sizeof(double));
        qtree[k] = nodes[cd.expanded].tree;
        qmax[k] = N0 + maybe[k];
        // k = 2e log10(#nodes) (:309) can only grow with the nodes accepted earlier in this round
        kmax[k] = (int32_t)(size_t)(2 * M_E * std::log10((double)(N0 + maybe[k])));
        if (kmax[k] <= 0) active[k] = 0;
      }
      const double RMAX = 1e30;
      for (int it = 0; it < 200; ++it) {
        bool any = false;
        for (int k = 0; k < m; ++k) any |= active[k] != 0;
        if (!any) break;
        std::vector<int32_t> cnt;
        std::vector<std::vector<HitRec>> out;
        c.sweep_lists(q6.data(), m, r, qtree.data(), qmax.data(), active, KCAP, N0 + n, cnt, out);
        st.sweeps += 1;
        st.sweep_nodes += (uint64_t)(N0 + n);
        for (int k = 0; k < m; ++k) {
          if (!active[k]) continue;
          st.sweep_queries += 1;
          if (cnt[k] > KCAP) { hi[k] = r[k]; r[k] = 0.5 * (lo[k] + hi[k]); continue; }
          int store_hits = 0;
          for (const HitRec& h : out[k]) store_hits += h.id < N0;
          if (store_hits >= kmax[k] || r[k] >= RMAX) {
            lists[k] = out[k];
            active[k] = 0;
          } else {
            lo[k] = r[k];
            r[k] = hi[k] > 0 ? 0.5 * (lo[k] + hi[k]) : std::min(RMAX, r[k] * 1.5);
          }
        }
      }
      double rsum = 0;
      int rcount = 0;
      for (int k = 0; k < m; ++k) {
        Cand& cd = cands[maybe[k]];
        if (kmax[k] <= 0) continue;
        // store members: the kmax nearest; wave-mates: closer than the kmax-th store member
        double dk = std::numeric_limits<double>::infinity();
        int seen = 0;
        for (const HitRec& h : lists[k]) {
          if (h.id >= N0) continue;
          if (++seen == kmax[k]) { dk = h.d; break; }
        }
        seen = 0;
        for (const HitRec& h : lists[k]) {
          Member mb;
          if (h.id < N0) {
            if (seen >= kmax[k]) continue;
            ++seen;
            mb.id = h.id;
          } else {
            if (!(h.d <= dk)) continue;
            mb.id = -1 - (h.id - N0);
          }
          cd.members.push_back(mb);
        }
        if (dk < 1e29) { rsum += dk; ++rcount; }
      }
      if (rcount) knn_r = 1.15 * rsum / rcount;
      // both directions of every member edge
      std::vector<double> sa, sb;
      for (int k = 0; k < m; ++k) {
        Cand& cd = cands[maybe[k]];
        for (Member& mb : cd.members) {
          const double* mp = mb.id >= 0 ? nodes[mb.id].pos : cands[-1 - mb.id].pos;
          mb.seg_f = (int)(sa.size() / 6);
          sa.insert(sa.end(), cd.pos, cd.pos + 6);   // isPathFree(newPoint, neighbor)   :323
          sb.insert(sb.end(), mp, mp + 6);
          mb.seg_b = (int)(sa.size() / 6);
          sa.insert(sa.end(), mp, mp + 6);           // isPathFree(neighbor, newPoint)   :336
          sb.insert(sb.end(), cd.pos, cd.pos + 6);
        }
      }
      const int ns2 = (int)(sa.size() / 6);
      if (ns2) {
        std::vector<uint8_t> fr(ns2);
        std::vector<int32_t> fh(ns2), nsv(ns2);
        c.collide_segments(sa.data(), sb.data(), ns2, fr.data(), fh.data(), nsv.data());
        st.segments_executed += ns2;
        for (int k = 0; k < ns2; ++k) st.samples_executed += (uint64_t)nsv[k];
        for (int k = 0; k < m; ++k)
          for (Member& mb : cands[maybe[k]].members) {
            mb.fwd_free = fr[mb.seg_f] != 0; mb.fwd_fh = fh[mb.seg_f]; mb.fwd_ns = nsv[mb.seg_f];
            mb.bwd_free = fr[mb.seg_b] != 0; mb.bwd_fh = fh[mb.seg_b]; mb.bwd_ns = nsv[mb.seg_b];
          }
      }
      wait_ms += ms_since(t0);
    }
  }

  // ---- the int32 record stream of the owned candidates
  for (int i = 0; i < n; ++i) {
    Cand& cd = cands[i];
    if (!cd.answered) continue;
    records.push_back(i);
    records.push_back((cd.pose_hit ? 1 : 0) | (cd.par_free ? 2 : 0));
    records.push_back(cd.par_fh);
    records.push_back(cd.par_ns);
    records.push_back((int32_t)cd.nbs.size());
    records.push_back((int32_t)cd.members.size());
    for (const Nb& nb : cd.nbs) {
      records.push_back(nb.id);
      records.push_back(nb.tree);
      records.push_back((nb.same_tree ? 1 : 0) | (nb.free ? 2 : 0));
      records.push_back(nb.fh);
      records.push_back(nb.ns);
    }
    for (const Member& mb : cd.members) {
      records.push_back(mb.id);
      records.push_back((mb.fwd_free ? 1 : 0) | (mb.bwd_free ? 2 : 0));
      records.push_back(mb.fwd_fh);
      records.push_back(mb.fwd_ns);
      records.push_back(mb.bwd_fh);
      records.push_back(mb.bwd_ns);
    }
  }
  st.host_ms += ms_since(t_host) - wait_ms;
}

// all = concatenation of every rank's record stream (rank order), counts in int32 words
void Forest::round_commit(const int32_t* all, const int32_t* counts, int world) {
  Ctx& c = *ctx;
  HIPCHK(hipSetDevice(c.device));
  if (!pending_round) throw HipError{"forest: round_commit without round_begin"};
  if (world != cfg.world) throw HipError{"forest: round_commit world size mismatch"};
  auto t_host = Clock::now();
  double wait_ms = 0;
  const int n = (int)cands.size();
  // ---- absorb the other ranks' answers
  size_t off = 0;
  for (int r = 0; r < world; ++r) {
    const int32_t* p = all + off;
    const int32_t* end = p + counts[r];
    off += (size_t)counts[r];
    if (counts[r] < 2 || p[0] != REC_MAGIC || p[1] != n) throw HipError{"forest: ranks disagree on the round (diverged state)"};
    p += 2;
    while (p < end) {
      int i = p[0];
      if (i < 0 || i >= n || i % world != r) throw HipError{"forest: malformed record stream"};
      Cand& cd = cands[i];
      int flags = p[1], nn = p[4], nm = p[5];
      if (r != cfg.rank) {
        cd.answered = true;
        cd.pose_hit = flags & 1;
        cd.par_free = (flags & 2) != 0;
        cd.par_fh = p[2];
        cd.par_ns = p[3];
        cd.nbs.resize(nn);
        for (int k = 0; k < nn; ++k) {
          const int32_t* q = p + 6 + 5 * k;
          Nb& nb = cd.nbs[k];
          nb.id = q[0];
          nb.tree = q[1];
          nb.same_tree = q[2] & 1;
          nb.free = (q[2] & 2) != 0;
          nb.fh = q[3];
          nb.ns = q[4];
        }
        cd.members.resize(nm);
        for (int k = 0; k < nm; ++k) {
          const int32_t* q = p + 6 + 5 * (size_t)nn + 6 * k;
          Member& mb = cd.members[k];
          mb.id = q[0];
          mb.fwd_free = q[1] & 1;
          mb.bwd_free = (q[1] & 2) != 0;
          mb.fwd_fh = q[2]; mb.fwd_ns = q[3]; mb.bwd_fh = q[4]; mb.bwd_ns = q[5];
        }
      }
      p += 6 + 5 * (size_t)nn + 6 * (size_t)nm;
    }
  }
  // ---- replay expandNode in slot order (src/forest.h:240-376)
  auto calls = [](int fh, int ns) -> uint64_t {  // Collide calls isPathFree makes (early exit at the first hit)
    return fh > 0 ? (uint64_t)fh : (uint64_t)ns;
  };
  std::vector<double> app_pos;
  std::vector<int32_t> app_tree;
  for (int i = 0; i < n; ++i) {
    Cand& cd = cands[i];
    Slot& sl = slots[cd.slot];
    const unsigned iteration = (unsigned)(iter0 + i + 1);
    if (!cd.in_lim) continue;                                  // :246 !result
    if (!cd.answered) throw HipError{"forest: a candidate has no answer record (missing rank?)"};
    st.collide_calls += 1;
    if (cd.pose_hit) continue;                                 // :246 env.Collide(newPoint)
    st.path_free_calls += 1;
    st.collide_calls += calls(cd.par_fh, cd.par_ns);
    if (!cd.par_free) continue;                                // :246 !isPathFree(expanded, newPoint)
    st.nn_queries += (uint64_t)trees.size();                   // :262-267 one radiusSearch per tree
    const int expanded = cd.expanded;
    const int mine = nodes[expanded].tree;
    bool reject = false;
    for (const Nb& nb : cd.nbs) {
      int nb_node;
      if (nb.id >= 0) nb_node = nb.id;
      else {
        nb_node = cands[-1 - nb.id].accepted_id;
        if (nb_node < 0) continue;                             // that sample never became a node
      }
      st.path_free_calls += 1;
      st.collide_calls += calls(nb.fh, nb.ns);
      if (nb.same_tree) {
        if (nb.free) { reject = true; break; }                 // :276-280 overcrowded
      } else {
        if (nb.free) {                                         // :288-294
          std::vector<Border>& bp = border(nb.tree, mine);
          int a = std::min(nb_node, expanded), b = std::max(nb_node, expanded);
          bool found = false;
          for (const Border& x : bp) if (x.n1 == a && x.n2 == b) { found = true; break; }
          if (!found) {
            double d = nodes[nb_node].d_root + nodes[expanded].d_root + sffg::dist6(nodes[nb_node].pos, nodes[expanded].pos);
            bp.push_back({a, b, d});
          }
        }
        reject = true;                                         // :296-299
        break;
      }
    }
    if (reject) continue;
    int id;
    if (cfg.optimize) {                                        // :307-351 SFF*: choose parent, rewire
      double best = sffg::dist6(cd.pos, nodes[expanded].pos) + nodes[expanded].d_root;
      const size_t ksff = (size_t)(2 * M_E * std::log10((double)nodes.size()));  // Node::globId (:309)
      struct KN { double d; int order; int node; const Member* mb; };
      std::vector<KN> knn;
      for (const Member& mb : cd.members) {
        int nd = mb.id >= 0 ? mb.id : cands[-1 - mb.id].accepted_id;
        if (nd < 0) continue;
        knn.push_back({sffg::dist6(cd.pos, nodes[nd].pos), nodes[nd].idx_in_tree, nd, &mb});
      }
      std::sort(knn.begin(), knn.end(), [](const KN& a, const KN& b) { return a.d < b.d || (a.d == b.d && a.order < b.order); });
      if (knn.size() > ksff) knn.resize(ksff);
      if (knn.size() < std::min(ksff, trees[mine].size()))
        throw HipError{"forest: k-nearest candidate set incomplete (internal error)"};
      int parent = expanded;
      for (const KN& kn : knn) {                               // :320-327
        double nd = sffg::dist6(cd.pos, nodes[kn.node].pos) + nodes[kn.node].d_root;
        if (nd < best - SFFG_TOL) {
          st.path_free_calls += 1;
          st.collide_calls += calls(kn.mb->fwd_fh, kn.mb->fwd_ns);
          if (kn.mb->fwd_free) { best = nd; parent = kn.node; }
        }
      }
      id = add_node(cd.pos, mine, parent, sffg::dist6(cd.pos, nodes[parent].pos), best, iteration);  // :329
      for (const KN& kn : knn) {                               // :332-350
        double npd = sffg::dist6(nodes[kn.node].pos, cd.pos);
        double proposed = best + npd;
        if (proposed < nodes[kn.node].d_root - SFFG_TOL) {
          st.path_free_calls += 1;
          st.collide_calls += calls(kn.mb->bwd_fh, kn.mb->bwd_ns);
          if (kn.mb->bwd_free) {
            nodes[kn.node].parent = id;
            nodes[kn.node].d_closest = npd;
            nodes[kn.node].d_root = proposed;                  // descendants keep their old cost (Appendix A.6)
          }
        }
      }
      st.nn_queries += 1;                                      // :317 knnSearch
    } else {
      id = add_node(cd.pos, mine, expanded, cd.pdist, cd.pdist + nodes[expanded].d_root, iteration);  // :353
    }
    cd.accepted_id = id;
    frontier.push_back(id);                                    // :365
    sl.failing = false;
    app_pos.insert(app_pos.end(), cd.pos, cd.pos + 6);
    app_tree.push_back(mine);
  }
  // ---- commit the accepted nodes to the device store (replaces flannIndex->addPoints, :367)
  if (n > 0) c.store_n = N0;
  if (!app_tree.empty()) {
    auto t0 = Clock::now();
    c.store_append(app_pos.data(), app_tree.data(), (int)app_tree.size());
    wait_ms += ms_since(t0);
  }
  pending_round = false;
  // ---- wave bookkeeping (src/forest.h:155: at most ThresholdMisses attempts per slot)
  bool any_failing = false;
  for (const Slot& s : slots) any_failing |= s.failing;
  if (round >= cfg.threshold_misses || !any_failing || solved || iter >= cfg.max_iterations) end_wave();
  st.host_ms += ms_since(t_host) - wait_ms;
}

void Forest::run(int max_waves) {
  if (cfg.world != 1) throw HipError{"forest: run() drives a single-GPU forest; use round_begin/round_commit"};
  auto t0 = Clock::now();
  const uint64_t w0 = st.waves;
  while (true) {
    if (!in_wave) {
      if (terminated()) break;
      if (max_waves > 0 && (int)(st.waves - w0) >= max_waves) break;
    }
    round_begin();
    int32_t cnt = (int32_t)records.size();
    round_commit(records.data(), &cnt, 1);
  }
  st.total_ms += ms_since(t0);
}

uint64_t Forest::fingerprint() const {
  uint64_t x = 1469598103934665603ULL;
  auto mix = [&](const void* p, size_t n) {
    const unsigned char* c = (const unsigned char*)p;
    for (size_t i = 0; i < n; ++i) { x ^= c[i]; x *= 1099511628211ULL; }
  };
  for (const FNode& n : nodes) {
    int32_t v[3] = {n.parent, n.tree, (int32_t)n.iter};
    mix(v, sizeof v);
    mix(n.pos, sizeof n.pos);
  }
  return x;
}

}  // namespace sff
