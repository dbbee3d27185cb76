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
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>

#include "engine.h"
#include "sff_geom.h"

namespace sff {

#define HIPCHK(x) hip_check((x), #x)
using Clock = std::chrono::steady_clock;
static double g_sec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
extern double g_sweep_dbg[4];
static double g_star[4] = {0, 0, 0, 0};   // SFF* second stage: k-nearest sweeps, member lists, edge batch, sweep iterations
static double g_wait[3] = {0, 0, 0};   // blocked on: early copy, final sync; [2] = second read pass
static uint64_t g_items[4] = {0, 0, 0, 0};   // rounds, edge work items, items after the cull, poses after the cull
static uint64_t g_cnt[4] = {0, 0, 0, 0};   // candidates, skipped by the replay, settled on the device, accepted
static const bool g_prof = getenv("SFFGPU_PROFILE") != nullptr;
struct Sec {
  int k;
  Clock::time_point t0;
  explicit Sec(int kk) : k(kk), t0(Clock::now()) {}
  ~Sec() { g_sec[k] += std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }
};
void forest_profile_dump() {
  if (g_prof && g_items[0]) fprintf(stderr, "[sffgpu per round] edge work items (64-sample chunks) %.0f\n", (double)g_items[1] / g_items[0]);
  if (g_prof) fprintf(stderr, "[sffgpu candidates] %llu skipped %llu settled %llu\n", (unsigned long long)g_cnt[0], (unsigned long long)g_cnt[1], (unsigned long long)g_cnt[2]);
  if (g_prof && g_star[3] > 0) fprintf(stderr, "[sffgpu sweep_lists ms] enqueue %.1f wait %.1f unpack %.1f, %.0f queries\n", g_sweep_dbg[0], g_sweep_dbg[1], g_sweep_dbg[2], g_sweep_dbg[3]);
  if (g_prof && g_star[3] > 0) fprintf(stderr, "[sffgpu SFF* stage 2 ms] k-nearest sweeps %.1f (%.0f passes) member lists %.1f edge batch %.1f\n", g_star[0], g_star[3], g_star[1], g_star[2]);
  if (g_prof) fprintf(stderr, "[sffgpu waits ms] early copy %.1f final sync %.1f | read pass 2 %.1f\n", g_wait[0], g_wait[1], g_wait[2]);
  if (g_prof) fprintf(stderr, "[sffgpu host ms] prep %.1f launch %.1f read %.1f records %.1f deser %.1f replay %.1f append %.1f endwave %.1f\n", g_sec[0], g_sec[1], g_sec[2], g_sec[3], g_sec[4], g_sec[5], g_sec[6], g_sec[7]);
}
static double ms_since(Clock::time_point t0) {
  return std::chrono::duration<double, std::milli>(Clock::now() - t0).count();
}

// ---- priority frontier heap: the reference's BubbleDown / BubbleUp / pop-at-index (src/heap.h:122-238)
double PHeap::cost(int i) const { return sffg::dist6((*nodes)[v[i]].pos, ref); }   // Distance(node, refPoint)
void PHeap::bubble_down(int index) {
  const int size = (int)v.size();
  const int l = 2 * index + 1, r = 2 * index + 2;
  if (l >= size) return;
  int mi = index;
  if (cost(index) > cost(l)) mi = l;
  if (r < size && cost(mi) > cost(r)) mi = r;
  if (mi != index) { std::swap(v[index], v[mi]); bubble_down(mi); }
}
void PHeap::bubble_up(int index) {
  if (index == 0) return;
  const int p = (index - 1) / 2;
  if (cost(p) > cost(index)) { std::swap(v[p], v[index]); bubble_up(p); }
}
void PHeap::push(int n) { v.push_back(n); bubble_up((int)v.size() - 1); }
int PHeap::pop() {
  int mn = v[0];
  v[0] = v.back();
  v.pop_back();
  bubble_down(0);
  return mn;
}
int PHeap::pop_at(int id) {
  const int size = (int)v.size();
  const double old_cost = cost(id);
  int val = -1;
  if (id == size - 1) { val = v.back(); v.pop_back(); }
  else if (size > id) {
    const double new_cost = cost(size - 1);
    val = v[id];
    v[id] = v[size - 1];
    v.pop_back();
    if (new_cost < old_cost) bubble_up(id); else bubble_down(id);
  }
  return val;
}
bool Forest::tree_frontiers_empty(int t) const {   // Tree::EmptyFrontiers (src/primitives.h:542-556)
  for (const PHeap& h : heaps[t]) if (!h.v.empty()) return false;
  return true;
}
bool Forest::all_frontiers_empty() const {
  for (size_t t = 0; t < heaps.size(); ++t) if (!tree_frontiers_empty((int)t)) return false;
  return true;
}

Forest::Forest(Ctx* c, const sffgpu_forest_cfg& cf, const double* roots6, int n_roots) : ctx(c), cfg(cf) {
  if (cfg.wave < 1) cfg.wave = 1;
  if (cfg.dim != 2 && cfg.dim != 6) throw HipError{"forest: dim must be 2 or 6"};
  if (cfg.world < 1) cfg.world = 1;
  if (cfg.rank < 0 || cfg.rank >= cfg.world) throw HipError{"forest: rank outside [0, world)"};
  if (n_roots < 1) throw HipError{"forest: at least one root"};
  if (!c->have_env || !c->have_robot) throw HipError{"forest: upload ENV and ROBOT meshes first"};
  rng.reseed(cfg.seed);
  if (cfg.wave >= 64) rng_ahead.assign((size_t)8 * cfg.wave + 4096, 0);
  num_roots = n_roots + (cfg.has_goal ? 1 : 0);
  trees.resize(num_roots);
  {   // no reallocation (and copy) of the node records while the forest grows
    const size_t cap = (size_t)std::min(std::max(cfg.node_budget, 4096), 1 << 26) + (size_t)cfg.wave + 64;
    nodes.reserve(cap);
    nflag.reserve(cap);
  }
  {   // which engine commits the rounds: the device-resident one (forest_dev.cpp) for plain SFF and SFF*,
      // the host replay below for goal / priority modes (and when SFFGPU_ENGINE=host asks for it)
    const char* e = getenv("SFFGPU_ENGINE");
    const std::string want = e ? e : "";
    // (every wave size: at wave = 1 - the reference's own order - the device engine's one graph launch + one status
    // read per wave is 1.3x the host replay's per-round synchronisation, at wave 64 1.6x: profiles/r3_small_waves.txt)
    dev.on = device_eligible() && want != "host" && cfg.wave <= 64 * SFFK_DEV_MAX_GROUPS;
  }
  // (device engine: a wave of new nodes past the budget plus the round's temporaries behind them)
  ctx->store_reset(std::max(cfg.node_budget, 4096) + (dev.on ? 2 * cfg.wave + 128 : cfg.wave + 64));
  std::vector<int32_t> tids(n_roots);
  for (int j = 0; j < n_roots; ++j) {          // src/forest.h:60-76
    int id = add_node(roots6 + 6 * (size_t)j, j, -1, 0, 0, 0);
    if (cfg.record_parents) hist.push_back({id, -1, 0u});
    frontier.push_back(id);
    nflag[id] |= 2;
    tids[j] = j;
  }
  {
    // cell edge >= the planner's neighbour radius max(parentDistance ~ SamplingDistance, treeDistance)
    double cell = 1.01 * std::max(cfg.sampling_dist, cfg.dist_tree) + 4 * ctx->sweep_eps();
    ctx->grid_rebuilds = 0;
    ctx->grid_exhausted = false;
    ctx->grid_bk = 8;
    if (const char* e = getenv("SFFGPU_TEST_GRID_BK")) ctx->grid_bk = std::max(1, std::min(8, atoi(e)));   // tests: tiny buckets to start with
    ctx->grid_cell0 = cell;
    query_wide = cfg.dim != 2 && std::min(cfg.sampling_dist, cfg.dist_tree) < 2.0 * 3.14159265358979323846;
    if (use_priority()) query_wide = true; // (the heaps crowd the samples around the trees' best nodes: more than 24 hits happen)
    // the overflow list is checked once per wave and re-celled at a quarter full: three quarters of it must hold
    // whatever TWO waves can add (at most `wave` nodes each; the device engine keeps one wave enqueued ahead of the
    // one whose status it reads), so that no insert is ever dropped between two checks
    if (cfg.wave > (1 << 27)) throw HipError{"forest: wave too large"};
    ctx->gridv_ovf_cap_next = std::max(65536, 4 * cfg.wave);
    ctx->tgrid_ovf_min = cfg.wave + 64;
    ctx->grid_setup(cfg.limits, cell);
  }
  ctx->store_append(roots6, tids.data(), n_roots);
  if (cfg.has_goal) {                          // src/forest.h:91-109: searched like any tree, never expanded
    goal_node = add_node(cfg.goal, n_roots, -1, 0, 0, 0);
    if (cfg.record_parents) hist.push_back({goal_node, -1, 0u});
    int32_t gt = n_roots;
    ctx->store_append(cfg.goal, &gt, 1);
  }
  ctx->grid_insert_new();
  if (use_priority()) {            // Tree::AddFrontier as called at src/forest.h:78-88 / :104-108
    heaps.resize(num_roots);
    auto add_heap = [&](int tree, int ref_node) {
      PHeap hp;
      hp.nodes = &nodes;
      memcpy(hp.ref, nodes[ref_node].pos, sizeof hp.ref);
      for (int id : trees[tree]) hp.v.push_back(id);
      for (int k = (int)hp.v.size() - 1; k >= 0; --k) hp.bubble_down(k);   // Heap::sort
      heaps[tree].push_back(hp);
    };
    if (!cfg.has_goal) {
      for (int i = 0; i < n_roots; ++i)
        for (int j = 0; j < n_roots; ++j)
          if (i != j) add_heap(i, trees[j][0]);
    } else {
      for (int i = 0; i < n_roots; ++i) add_heap(i, goal_node);
    }
  }
  memset(&st, 0, sizeof st);
  knn_r = 2.5 * cfg.sampling_dist;
  if (const char* e = getenv("SFFGPU_TEST_HITCAP")) hit_cap = std::min(64, std::max(1, atoi(e)));  // one lane per hit
  dev.ord_enabled = !(getenv("SFFGPU_NO_ORDER") && atoi(getenv("SFFGPU_NO_ORDER")) != 0);   // spatial order of a wave's slots (sffk::OrderView)
  if (const char* e = getenv("SFFGPU_ORDER_MIN_WAVE")) dev.ord_min_wave = std::max(2, atoi(e));
  if (hit_cap < 24) query_wide = true;   // (tests shrink the hit list of the wide kernel)
  if (const char* e = getenv("SFFGPU_TEST_NBCAP")) nb_cap = std::max(1, atoi(e));
  if (const char* e = getenv("SFFGPU_TEST_STAR_PASSES")) star_pass_limit = std::max(1, atoi(e));
  if (getenv("SFFGPU_TEST_EXCHANGE_SELF")) test_exchange_self = true;
  if (const char* e = getenv("SFFGPU_STAR_TAIL")) star_tail = atoi(e) != 0;
  // waves of one slot: the speculative kernel's shape (k_spec_waves; forest_dev.cpp: spec_setup)
  if (const char* e = getenv("SFFGPU_SPEC")) dev.spec_off = atoi(e) == 0;
  if (const char* e = getenv("SFFGPU_SPEC_DEPTH")) dev.spec_depth = atoi(e);
  if (const char* e = getenv("SFFGPU_SPEC_SETS")) dev.spec_sets_want = atoi(e);
  if (const char* e = getenv("SFFGPU_TEST_SPEC_STALL")) dev.spec_test_stall = atoi(e);
  if (const char* e = getenv("SFFGPU_SPEC_PIPE")) dev.spec_pipe = atoi(e) != 0;
  if (const char* e = getenv("SFFGPU_NO_DEV_TRIG")) dev.dev_trig_off = atoi(e) != 0;
  if (const char* e = getenv("SFFGPU_STAR_TAIL_WGS")) star_tail_wgs = std::max(1, atoi(e));
  if (const char* e = getenv("SFFGPU_TEST_STAR_STALL")) star_tail_stall = std::max(0, atoi(e));
  if (const char* e = getenv("SFFGPU_NO_GRAPH")) dev.graph_enabled = atoi(e) == 0;
  else if (cfg.optimize) {
    // rocprofv3 (ROCm 7.2) crashes while tracing replays of the SFF* wave graph (~130 kernel nodes; the plain SFF graph
    // of ~35 nodes traces fine): under the profiler SFF* waves are launched kernel by kernel
    const char* pre = getenv("LD_PRELOAD");
    if (pre && strstr(pre, "rocprofiler")) {
      dev.graph_enabled = false;
      fprintf(stderr, "[sffgpu] profiler library in LD_PRELOAD: SFF* waves are launched kernel by kernel (SFFGPU_NO_GRAPH=0 / 1 overrides)\n");
    }
  }
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
  nflag.push_back(0);
  trees[tree].push_back(id);
  return id;
}

std::vector<Border>& Forest::border(int i, int j) {  // SymmetricMatrix, src/primitives.h:572-596
  if (i > j) std::swap(i, j);
  return borders[{i, j}];
}

// src/forest.h:379-418
template <class HasBorder>
int Forest::max_connected_over(HasBorder has_border, std::vector<int>& out_connected) const {
  int max_conn = 0, remaining = num_roots;
  std::vector<char> conn(num_roots, 0);
  int unconnected = 0;
  while (max_conn < remaining) {
    out_connected.clear();
    std::vector<int> stack{unconnected};
    conn[unconnected] = 1;
    while (!stack.empty()) {
      int root = stack.front();
      stack.erase(stack.begin());
      out_connected.push_back(root);
      for (int i = 0; i < num_roots; ++i) {
        if (root == i) continue;
        if (!conn[i] && has_border(std::min(root, i), std::max(root, i))) {
          conn[i] = 1;
          stack.insert(stack.begin(), i);
        }
      }
    }
    max_conn = (int)out_connected.size();
    for (int i = 0; i < num_roots; ++i)
      if (!conn[i]) { unconnected = i; break; }
    remaining -= max_conn;
  }
  return max_conn;
}
int Forest::max_connected() {
  return max_connected_over([&](int a, int b) {
    auto it = borders.find({a, b});
    return it != borders.end() && !it->second.empty();
  }, connected);
}

// what sffgpu_forest_get_stats reports; while the device engine owns the state the figures come from its status
// block (and the small pair matrix), without downloading the forest
void Forest::fill_stats(sffgpu_forest_stats* out) {
  sffgpu_forest_stats s = st;
  if (dev.active && dev.host_stale) {
    const sffk::DevCtrl& k = dev.last;
    std::vector<uint8_t> pair((size_t)num_roots * num_roots);
    HIPCHK(hipMemcpy(pair.data(), dev.pair.p, pair.size(), hipMemcpyDeviceToHost));
    std::vector<int> conn;
    const int mc = max_connected_over([&](int a, int b) { return pair[(size_t)a * num_roots + b] != 0; }, conn);
    s.iterations = k.iter;
    s.solved = (k.solved || (!cfg.has_goal && mc == num_roots)) ? 1 : 0;
    s.n_nodes = k.n_nodes;
    s.frontier_size = k.frontier_n;
    s.closed_size = k.closed_n;
    s.n_connected = (int)conn.size();
    s.n_borders = k.n_borders;
    s.collide_calls = k.collide_calls;
    s.path_free_calls = k.path_free_calls;
    s.nn_queries = k.nn_queries;
    s.poses_executed = k.poses_executed;
    s.segments_executed = k.segments_executed;
    s.samples_executed = k.samples_executed;
    s.waves = k.waves;
    s.sweeps = k.rounds;
    s.sweep_nodes = k.round_nodes;
    s.sweep_queries = k.round_queries;
    s.star_rounds = k.star_rounds;
    s.star_passes = k.star_passes;
    s.star_members = k.star_members;
    s.star_rewires = k.star_rewires;
    s.spec_steps = k.spec_steps;
    s.spec_evaluated = k.spec_evaluated;
    s.spec_committed = k.spec_committed;
  } else {
    s.iterations = iter;
    bool sv = solved;
    if (!sv && !cfg.has_goal) sv = max_connected() == num_roots;  // src/forest.h:204-206
    s.solved = sv;
    s.n_nodes = (int)nodes.size();
    s.frontier_size = (int)frontier.size();
    s.closed_size = (int)closed.size();
    s.n_connected = (int)connected.size();
    int nb = 0;
    for (auto& kv : borders) nb += (int)kv.second.size();
    s.n_borders = nb;
  }
  s.n_trees = (int)trees.size();
  if (dev.active || dev.inited) {
    s.query_clock_ms = (double)dev.last.q_ticks / (double)ctx->wall_clock_khz;
    s.query_clock_launches = dev.last.q_launches;
  }
  s.grid_rebuilds = (uint64_t)ctx->grid_rebuilds;
  s.sweep_ms = ctx->kernel_ms_total(T_SWEEP);
  s.collide_ms = ctx->kernel_ms_total(T_COLLIDE);
  s.sample_ms = ctx->kernel_ms_total(T_SAMPLE);
  s.commit_ms = ctx->kernel_ms_total(T_COMMIT);
  s.exchange_ms = ctx->kernel_ms_total(T_EXCHANGE);
  s.graph_launches = dev.graph_launches;
  *out = s;
}

Forest::~Forest() {
  DevBuf* bufs[] = {&dev.ctrl, &dev.parent, &dev.d_root, &dev.d_closest, &dev.iter, &dev.nflag, &dev.frontier, &dev.closed,
                    &dev.claim, &dev.slot_node, &dev.slot_fail, &dev.act_slot, &dev.b_n1, &dev.b_n2, &dev.b_ta, &dev.b_tb,
                    &dev.b_dist, &dev.bt_key, &dev.bt_val, &dev.pair, &dev.ring, &dev.ulist,
                    &dev.d_parent, &dev.d_parent2, &dev.d_force, &dev.fault_pending, &dev.frontier2, &dev.rm_words, &dev.rm_pref,
                    &dev.slot_pos, &dev.act_slot2, &dev.trig, &dev.s_ktab, &dev.s_tree_cnt, &dev.s_head, &dev.s_mcnt, &dev.s_mid, &dev.s_md,
                    &dev.s_next, &dev.s_prop, &dev.s_best, &dev.s_psel, &dev.s_dcl, &dev.s_cnt, &dev.s_accs, &dev.s_hdr, &dev.s_changed,
                    &dev.s_ew, &dev.s_ida, &dev.s_idb, &dev.s_sub, &dev.s_segns, &dev.s_fh, &dev.s_sovf, &dev.s_evs, &dev.s_evn, &dev.s_eve,
                    &dev.s_evd, &dev.s_acc, &dev.s_backup, &dev.s_items, &dev.s_dbg, &dev.s_hist, &dev.w_acc, &dev.acc_pref, &dev.ustate32, &dev.wg_pub, &dev.commit_seq, &dev.kc_trace, &x_send, &x_recv,
                    &dev.hp_base, &dev.hp_size, &dev.hp_v, &dev.hp_key, &dev.hp_pos, &dev.hp_ref, &dev.hp_gen, &dev.hp_cnt, &dev.slot_tree, &dev.slot_heap, &dev.slot_idx, &dev.slot_word, &dev.hp_plan,
                    &dev.ord_hist, &dev.ord_start, &dev.ord_key, &dev.ord_rank, &dev.ord_pos, &dev.ord_lst, &dev.ord_cnt,
                    &dev.w_ev, &dev.ev_h, &dev.ev_nb, &dev.ev_raw, &dev.spec_tab, &dev.spec_area, &dev.qclk_sh};
  if (dev.inited) {
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamSynchronize(ctx->copy_stream);
  }
  for (DevBuf* b : bufs) b->release();
  dev.h_ctrl.release();
  dev.h_ring.release();
  dev.h_trig.release();
  if (dev.wave_graph) (void)hipGraphExecDestroy(dev.wave_graph);
  if (dev.ev_ring) (void)hipEventDestroy(dev.ev_ring);
  if (dev.ev_wave) (void)hipEventDestroy(dev.ev_wave);
  if (dev.ev_wave2) (void)hipEventDestroy(dev.ev_wave2);
}

// node selection for every slot of the wave, src/forest.h:136-151 (non-priority mode)
void Forest::begin_wave() {
  slots.clear();
  if (use_priority() && !empty_frontier) {     // src/forest.h:126-147 priority frontier
    int pool = 0;
    for (auto& hs : heaps) if (!hs.empty()) pool += (int)hs[0].v.size();
    const int n_slots = std::max(1, std::min(cfg.wave, pool));
    for (int s = 0; s < n_slots; ++s) {
      if (all_frontiers_empty()) break;        // every frontier node is already held by a slot
      int t = -1;
      while (t < 0 || tree_frontiers_empty(t)) t = rng.uniform_int(0, (int)trees.size() - 1);
      int hp = -1;
      while (hp < 0 || heaps[t][hp].v.empty()) hp = rng.uniform_int(0, (int)heaps[t].size() - 1);
      PHeap& prior = heaps[t][hp];
      Slot sl;
      if (sffg::uniform_real(rng.next(), 0.0, 1.0) <= cfg.priority_bias) sl.node = prior.pop();     // :143-144
      else sl.node = prior.pop_at(rng.uniform_int(0, (int)prior.v.size() - 1));                      // :145-147
      sl.from_closed = false;
      sl.failing = true;
      sl.tree = t;
      sl.heap = hp;
      slots.push_back(sl);
    }
  } else {
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
  }
  round = 0;
  in_wave = true;
  ++st.waves;
}

// src/forest.h:160-201
void Forest::end_wave() {
  // the reference erases each exhausted node from the frontier deque one by one (:160-163); all
  // removals of a wave are applied in one order-preserving compaction, which leaves the same deque
  bool removed = false;
  for (Slot& sl : slots) {
    if (use_priority() && sl.tree >= 0) {      // src/forest.h:164-181
      if (sl.failing) {                        // exhausted: drop it from the tree's other heaps too
        for (int i = (int)heaps[sl.tree].size() - 1; i > -1; --i) {
          if (i == sl.heap) continue;
          PHeap& p = heaps[sl.tree][i];
          for (int j = (int)p.v.size() - 1; j > -1; --j)
            if (j < (int)p.v.size() && p.v[j] == sl.node) p.pop_at(j);
        }
        if (!(nflag[sl.node] & 1)) {
          nflag[sl.node] |= 1;
          closed.push_back(sl.node);
        }
      } else {
        heaps[sl.tree][sl.heap].push(sl.node); // back onto the heap it was taken from
      }
      continue;
    }
    if (sl.failing && !sl.from_closed && (nflag[sl.node] & 2)) {
      nflag[sl.node] = (uint8_t)((nflag[sl.node] & ~2) | 1);
      closed.push_back(sl.node);
      removed = true;
    }
  }
  if (removed) {
    size_t w = 0;
    for (size_t r = 0; r < frontier.size(); ++r)
      if (nflag[frontier[r]] & 2) frontier[w++] = frontier[r];
    frontier.resize(w);
  }
  ctx->grid_check();
  if (!solved && use_priority()) empty_frontier = all_frontiers_empty();   // src/forest.h:184-191
  else empty_frontier = frontier.empty();
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

// RandGen::randomPointInDistance (src/randGen.h:70-109) with the C library's trig: the same expressions, in the same
// order, as sffg::sample_point (csrc/sff_geom.h) - which the kernels evaluate with the portable trig - so that the two
// differ exactly where glibc and sff_pmath.h differ (<= 1 ulp).  Parity mode only (cfg.libm_sampling).
static void sample_point_libm(const uint64_t* w, const double* center, double dist, int dim, double* out) {
  sffg::SampleTrig t{};
  const double phi = sffg::sample_angle(w[0]);
  t.c_phi = std::cos(phi);
  t.s_phi = std::sin(phi);
  if (dim != 2) {
    const double theta = sffg::sample_angle(w[1]);
    t.c_theta = std::cos(theta);
    t.s_theta = std::sin(theta);
    t.acos_u = std::acos(sffg::sample_acos_arg(w[3]));
  }
  const double nolim[6] = {-1e300, 1e300, -1e300, 1e300, -1e300, 1e300};   // (the kernel applies the limits test)
  sffg::sample_point_with(w, center, dist, dim, nolim, out, t);
}

void Forest::round_begin() {
  Ctx& c = *ctx;
  HIPCHK(hipSetDevice(c.device));
  if (pending_round) throw HipError{"forest: round_begin called twice without round_commit"};
  if (dev.active) dev_to_host();   // the round protocol runs on the host path
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
  auto _t0 = Clock::now();
  if (cands.size() < slots.size()) cands.resize(slots.size());
  n_cands = 0;
  for (int s = 0; s < (int)slots.size(); ++s) {
    if (!slots[s].failing) continue;
    if (iter + n_cands >= cfg.max_iterations) break;
    cands[n_cands++].reset(s, slots[s].node);
  }
  const int n = n_cands;
  ++round;
  pending_round = true;
  records.clear();
  records.push_back(REC_MAGIC);
  records.push_back(n);
  for (int k = 0; k < 8; ++k) records.push_back(0);   // bulk counters, filled at the end of round_begin
  bulk_counts[0] = bulk_counts[1] = bulk_counts[2] = bulk_counts[3] = 0;
  if (n == 0) return;
  iter0 = iter;
  iter += n;
  N0 = (int)nodes.size();
  Tb = (N0 + 3) & ~3;
  c.store_reserve(Tb + n + 4);

  g_sec[0] += ms_since(_t0);
  auto _t1 = Clock::now();
  // ---- one GPU pipeline per round, a single host sync at its end:
  //   H2D {engine words, expanded ids, ForceChildren flags}
  //   k_sample_steer -> k_store_write (temporaries) -> k_sweep -> k_classify -> k_collide_poses
  //   -> k_collide_segments_dyn -> D2H {samples, flags, neighbour records, pose / edge answers}
  const int words_per = cfg.dim == 2 ? 1 : 6;
  const int CAP = hit_cap, NBCAP = nb_cap, STRIDE = 1 + NBCAP;
  // packed host input: words (n*6 u64) | parent (n i32) | force (n u8)
  const size_t in_words = 0, in_parent = (size_t)n * 48, in_force = in_parent + (size_t)n * 4;
  const size_t in_preset = ((in_force + (size_t)n + 15) / 16) * 16;   // (libm parity mode: n x 6 sample positions)
  const size_t in_bytes = in_preset + (cfg.libm_sampling ? (size_t)n * 48 : 0);
  c.p_in.ensure(in_bytes);
  c.r_in.ensure(in_bytes);
  {
    uint64_t* hw = reinterpret_cast<uint64_t*>(c.p_in.as<char>() + in_words);
    int32_t* hp = reinterpret_cast<int32_t*>(c.p_in.as<char>() + in_parent);
    uint8_t* hf = reinterpret_cast<uint8_t*>(c.p_in.as<char>() + in_force);
    if (words_per == 6) rng.fill(hw, (size_t)n * 6);
    for (int i = 0; i < n; ++i) {
      if (words_per != 6)
        for (int k = 0; k < 6; ++k) hw[6 * (size_t)i + k] = k < words_per ? rng.next() : 0;
      hp[i] = cands[i].expanded;
      hf[i] = nflag[cands[i].expanded] & 1;
    }
    if (cfg.libm_sampling) {
      double* hs = reinterpret_cast<double*>(c.p_in.as<char>() + in_preset);
      for (int i = 0; i < n; ++i)
        sample_point_libm(hw + 6 * (size_t)i, nodes[cands[i].expanded].pos, cfg.sampling_dist, cfg.dim, hs + 6 * (size_t)i);
    }
  }
  c.timing_on = c.timer_stride <= 1 || st.sweeps % (uint64_t)c.timer_stride == 0;
  c.round_scope = true;
  const uint64_t* d_words = reinterpret_cast<const uint64_t*>(c.r_in.as<char>() + in_words);
  const int32_t* d_parent = reinterpret_cast<const int32_t*>(c.r_in.as<char>() + in_parent);
  const uint8_t* d_force = reinterpret_cast<const uint8_t*>(c.r_in.as<char>() + in_force);
  // device output block in two D2H copies.  Early part (complete after k_classify, copied on a second stream
  // while the collision kernels run): pos | pdist | in_lim | records | edge sample counts.  Late part: edge
  // first hits (0 = redo on the host path) | ctrl (32 ints incl. the u64 settle counters) | pose answers | settle codes.
  // records: flags (n) | nnb (n) | nb ids (n*NBCAP) | nb meta (n*NBCAP)
  const size_t rec_ints = (size_t)n * (2 + 2 * NBCAP);
  const size_t o_pos = 0, o_pd = o_pos + (size_t)n * 48, o_lim = o_pd + (size_t)n * 8,
               o_rec = o_lim + ((size_t)n + 15) / 16 * 16, o_ns = o_rec + rec_ints * 4,
               o_fh = o_ns + (size_t)n * STRIDE * 4, o_ctrl = o_fh + (size_t)n * STRIDE * 4, o_pose = o_ctrl + 128,
               o_code = o_pose + (size_t)n, o_bytes = (o_code + (size_t)n + 15) / 16 * 16,
               o_ovf = o_bytes;   // (device-only scratch behind the copied block)
  const size_t early_bytes = o_fh;
  c.r_out.ensure(o_ovf + (size_t)n * STRIDE * 4);
  char* dout = c.r_out.as<char>();
  double* d_pos = reinterpret_cast<double*>(dout + o_pos);
  double* d_pd = reinterpret_cast<double*>(dout + o_pd);
  int32_t* d_rec = reinterpret_cast<int32_t*>(dout + o_rec);
  int32_t* d_ctrl = reinterpret_cast<int32_t*>(dout + o_ctrl);
  uint8_t* d_lim = reinterpret_cast<uint8_t*>(dout + o_lim);
  uint8_t* d_pose = reinterpret_cast<uint8_t*>(dout + o_pose);
  c.r_q.ensure((size_t)n * sizeof(sffk::SweepQuery));
  c.r_cnt.ensure((size_t)n * 4);
  c.r_hidx.ensure((size_t)n * CAP * 4);
  c.r_hdist.ensure((size_t)n * CAP * 8);
  c.r_sega.ensure((size_t)n * STRIDE * 48);
  c.r_segb.ensure((size_t)n * STRIDE * 48);
  HIPCHK(hipMemcpyAsync(c.r_in.p, c.p_in.p, in_bytes, hipMemcpyHostToDevice, c.stream));
  sffk::SampleParams prm{};
  memcpy(prm.limits, cfg.limits, sizeof prm.limits);
  prm.dist_tree = cfg.dist_tree;
  prm.sweep_abs_eps = c.sweep_eps();
  prm.rank = cfg.rank;
  prm.world = cfg.world;
  // one launch: sample + steer + limits, the round's temporary store entries [Tb, Tb+n) (the same query pass then
  // also finds, for every sample, the EARLIER samples of this round: query i sees ids < Tb + i), NaN placeholders
  // for [N0, Tb), and the per-round counter resets
  sffk::RoundTemps tmp{};
  tmp.st = sffk::NodeStoreMut{c.sx.as<float>(), c.sy.as<float>(), c.sz.as<float>(), c.syaw.as<float>(),
                              c.spitch.as<float>(), c.sroll.as<float>(), c.stree.as<int32_t>(), c.spos.as<double>()};
  tmp.cnt = c.r_cnt.as<int32_t>();
  tmp.tg = c.tgridv;
  tmp.ctrl = d_ctrl;
  c.r_sub.ensure((size_t)SFFK_SUBLISTS * SFFK_SUB_STRIDE * 4);
  tmp.sub = c.r_sub.as<int32_t>();
  tmp.n_perm = N0;
  tmp.base = Tb;
  tmp.preset = cfg.libm_sampling ? reinterpret_cast<const double*>(c.r_in.as<char>() + in_preset) : nullptr;
  c.r_qrec.ensure((size_t)n * sizeof(sffk::QRec));
  tmp.qrec = c.r_qrec.as<sffk::QRec>();
  memcpy(tmp.clear_org, c.envv.clear_org, sizeof tmp.clear_org);
  tmp.clear_inv = c.envv.clear_inv;
  c.time_begin(T_SAMPLE);
  sffk::launch_sample_steer(c.stream, d_words, d_parent, c.spos.as<double>(), nullptr, n, cfg.sampling_dist, cfg.dim,
                            prm, d_pos, d_lim, d_pd, c.r_q.as<sffk::SweepQuery>(), Tb, tmp);
  c.time_end();
  // the sweep only serves the queries of this rank's shard (the others are marked inactive)
  st.sweeps += 1;
  st.sweep_nodes += (uint64_t)(N0 + n);
  st.sweep_queries += (uint64_t)((n - cfg.rank + cfg.world - 1) / cfg.world);
  sffk::ClassifyArgs ca{};
  ca.n = n; ca.N0 = Tb; ca.cap = CAP; ca.nbcap = NBCAP; ca.rank = cfg.rank; ca.world = cfg.world;
  ca.goal_id = goal_node;
  ca.wide = query_wide ? 1 : 0;
  ca.qrec = c.r_qrec.as<sffk::QRec>();
  ca.dist_tree = cfg.dist_tree;
  ca.newpos = d_pos;
  ca.in_lim = d_lim;
  ca.pdist = d_pd;
  ca.parent = d_parent;
  ca.force = d_force;
  ca.cnt = c.r_cnt.as<int32_t>();
  ca.hit_idx = c.r_hidx.as<int32_t>();
  ca.hit_dist = c.r_hdist.as<double>();
  ca.tree = c.stree.as<int32_t>();
  ca.pos = c.spos.as<double>();
  ca.rec_flags = d_rec;
  ca.rec_nnb = ca.rec_flags + n;
  ca.rec_nb = ca.rec_nnb + n;
  ca.rec_meta = ca.rec_nb + (size_t)n * NBCAP;
  ca.seg_a = c.r_sega.as<double>();
  ca.seg_b = c.r_segb.as<double>();
  ca.seg_ns = reinterpret_cast<int32_t*>(dout + o_ns);
  ca.first_hit = reinterpret_cast<int32_t*>(dout + o_fh);
  ca.seg_ovf = reinterpret_cast<int32_t*>(dout + o_ovf);
  ca.ctrl = d_ctrl;
  // neighbour query (27 grid cells per sample in the node grid, plus the round's own grid for the EARLIER samples of
  // this round) and classification of the hits, one wavefront per sample
  c.time_begin(T_SWEEP);
  // (the clearance cull of the sample's pose and edge chunks is fused in: what needs the exact test is on r_items)
  const int list_cap = 4 * n * STRIDE + 65536;
  c.r_items.ensure((size_t)list_cap * SFFK_ITEM_BYTES);
  ca.items = c.r_items.p;
  ca.items_cap = list_cap;
  ca.sub = c.r_sub.as<int32_t>();
  ca.pose_hit = d_pose;
  const bool blocked = sffk::launch_query_classify(c.stream, c.gridv, &c.tgridv, c.store_view(), c.r_q.as<sffk::SweepQuery>(), ca, &c.envv);
  c.time_end();
  c.time_begin(T_COLLIDE);
  c.p_out.ensure(o_bytes);
  char* ho = c.p_out.as<char>();
  HIPCHK(hipEventRecord(c.ev_mid, c.stream));
  HIPCHK(hipStreamWaitEvent(c.copy_stream, c.ev_mid, 0));
  HIPCHK(hipMemcpyAsync(ho, dout, early_bytes, hipMemcpyDeviceToHost, c.copy_stream));
  HIPCHK(hipEventRecord(c.ev_early, c.copy_stream));
  // poses and edges together: the exact kernel on the survivors of the fused cull
  sffk::TempGridRef tref{c.tgridv, c.sx.as<float>() + Tb, c.sy.as<float>() + Tb, c.sz.as<float>() + Tb, n};
  // (SFF*: the k-nearest stage below still reads the round's own grid; it is emptied after that)
  sffk::TempGridRef tref_keep = tref;
  tref_keep.tg = sffk::GridView{};
  tref_keep.n = 0;
  sffk::launch_collide_items(c.stream, c.envv, c.robv, d_pos, n, ca.rec_flags, d_pose, ca.seg_a, ca.seg_b, ca.seg_ns,
                             STRIDE, ca.ctrl, c.r_items.p, ca.items_cap, ca.sub, ca.first_hit, ca.seg_ovf, cfg.optimize ? &tref_keep : &tref,
                             nullptr, blocked ? &ca : nullptr);
  c.time_end();
  // samples this rank can settle alone need no replay (with a goal the replay may stop in the middle of the
  // round, so there every sample stays in it)
  const bool settle_on_device = !cfg.has_goal;
  if (settle_on_device) {
    sffk::SettleArgs sa{};
    sa.n = n; sa.Tb = Tb; sa.nbcap = NBCAP; sa.stride = STRIDE; sa.n_trees = (int)trees.size();
    sa.in_lim = d_lim; sa.rec_flags = ca.rec_flags; sa.rec_nnb = ca.rec_nnb; sa.rec_nb = ca.rec_nb;
    sa.rec_meta = ca.rec_meta; sa.seg_ns = ca.seg_ns; sa.first_hit = ca.first_hit;
    sa.pose_hit = d_pose;
    sa.code = reinterpret_cast<uint8_t*>(dout + o_code);
    sa.bulk = reinterpret_cast<unsigned long long*>(dout + o_ctrl + 16);
    sffk::launch_settle(c.stream, sa);
  }
  HIPCHK(hipMemcpyAsync(ho + early_bytes, dout + early_bytes, o_bytes - early_bytes, hipMemcpyDeviceToHost, c.stream));
  c.timing_on = true;
  c.round_scope = false;
  g_sec[1] += ms_since(_t1);
  // the GPU is busy for a while: generate the engine words of the next draws now
  if (!rng_ahead.empty()) rng.prefetch(rng_ahead.data(), rng_ahead.size());

  const double* hpos = reinterpret_cast<const double*>(ho + o_pos);
  const double* hpd = reinterpret_cast<const double*>(ho + o_pd);
  const int32_t* hflags = reinterpret_cast<const int32_t*>(ho + o_rec);
  const int32_t* hnnb = hflags + n;
  const int32_t* hnb = hnnb + n;
  const int32_t* hmeta = hnb + (size_t)n * NBCAP;
  const int32_t* hns = reinterpret_cast<const int32_t*>(ho + o_ns);
  const int32_t* hfh = reinterpret_cast<const int32_t*>(ho + o_fh);
  const uint8_t* hlim = reinterpret_cast<const uint8_t*>(ho + o_lim);
  const uint8_t* hpose = reinterpret_cast<const uint8_t*>(ho + o_pose);
  const uint8_t* hcode = reinterpret_cast<const uint8_t*>(ho + o_code);

  // ---- read the answers in two passes: samples, neighbour records and edge sample counts as soon as the early
  // copy has landed (the collision kernels are still running), the pose / edge answers after the final sync.
  // Samples whose lists overflowed (and edges whose candidate lists did) take the host path below.
  std::vector<int> slow;        // samples to redo entirely on the host path
  std::vector<double> fix_a, fix_b;
  struct Fix { int cand; int slot; };
  std::vector<Fix> fixes;       // single edges to redo (triangle candidate list overflow)
  { auto tw = Clock::now(); HIPCHK(hipEventSynchronize(c.ev_early)); g_wait[0] += ms_since(tw); }
  auto _t2 = Clock::now();
  round_hpos = hpos;
  round_hpd = hpd;
  round_hlim = hlim;
  // (another rank's sample is filled in from its record, if it sends one)
  for (int i = cfg.rank; i < n; i += cfg.world) {
    Cand& cd = cands[i];
    memcpy(cd.pos, hpos + 6 * (size_t)i, sizeof cd.pos);
    cd.in_lim = hlim[i] != 0;
    cd.pdist = hpd[i];
    if (!cd.in_lim) continue;
    if (hflags[i] & 2) { slow.push_back(i); continue; }
    cd.answered = true;
    st.poses_executed += 1;
    const size_t s0 = (size_t)i * STRIDE;
    const int nnb = hnnb[i];
    cd.par_ns = hns[s0];
    uint64_t samples = (uint64_t)cd.par_ns;
    cd.nbs.resize(nnb);
    for (int k = 0; k < nnb; ++k) {
      Nb& nb = cd.nbs[k];
      const int id = hnb[(size_t)i * NBCAP + k];
      const int meta = hmeta[(size_t)i * NBCAP + k];
      nb.id = id < Tb ? id : -1 - (id - Tb);
      nb.tree = meta >> 1;
      nb.same_tree = meta & 1;
      nb.seg = -1;
      nb.ns = hns[s0 + 1 + k];
      samples += (uint64_t)nb.ns;
    }
    st.segments_executed += 1 + (uint64_t)nnb;
    st.samples_executed += samples;
  }
  g_sec[2] += ms_since(_t2);
  { auto tw = Clock::now(); timed_sync(); g_wait[1] += ms_since(tw); }
  _t2 = Clock::now();
  {
    const int32_t* hc = reinterpret_cast<const int32_t*>(ho + o_ctrl);
    g_items[0] += 1; g_items[1] += (uint64_t)hc[2];
  }
  if (settle_on_device) {
    const uint64_t* hb = reinterpret_cast<const uint64_t*>(ho + o_ctrl + 16);
    for (int k = 0; k < 4; ++k) bulk_counts[k] += hb[k];
  }
  // the replay's work list: samples inside the limits that were not settled on the device
  round_skip.assign((size_t)n, cfg.world > 1 ? 1 : 0);   // (another rank's sample joins the list in round_commit
  round_todo.clear();                                     //  if its owner reports it)
  g_cnt[0] += (uint64_t)n;
  for (int i = cfg.rank; i < n; i += cfg.world) {
    round_skip[i] = 0;
    if (settle_on_device ? hcode[i] != 0 : hlim[i] == 0) {   // settled (1) / outside the limits (2)
      round_skip[i] = 1;
      g_cnt[1] += 1;
      if (settle_on_device && hcode[i] == 1) g_cnt[2] += 1;
    } else {
      round_todo.push_back(i);
    }
  }
  const size_t n_todo = round_todo.size();
  for (size_t j = 0; j < n_todo; ++j) {
    if (j + 8 < n_todo) {
      __builtin_prefetch(hfh + (size_t)round_todo[j + 8] * STRIDE);
      __builtin_prefetch(&cands[round_todo[j + 8]].answered);
    }
    const int i = round_todo[j];
    Cand& cd = cands[i];
    if (!cd.answered) continue;
    cd.pose_hit = hpose[i] != 0;
    const size_t s0 = (size_t)i * STRIDE;
    cd.par_fh = hfh[s0] == 0x7fffffff ? -1 : hfh[s0];
    cd.par_free = cd.par_fh < 0;
    if (hfh[s0] == 0) fixes.push_back({i, 0});
    const int nnb = (int)cd.nbs.size();
    for (int k = 0; k < nnb; ++k) {
      Nb& nb = cd.nbs[k];
      nb.fh = hfh[s0 + 1 + k] == 0x7fffffff ? -1 : hfh[s0 + 1 + k];
      nb.free = nb.fh < 0;
      if (hfh[s0 + 1 + k] == 0) fixes.push_back({i, 1 + k});
    }
  }
  g_sec[2] += ms_since(_t2);
  g_wait[2] += ms_since(_t2);
  if (!fixes.empty()) {
    auto t0 = Clock::now();
    for (const Fix& f : fixes) {
      const Cand& cd = cands[f.cand];
      const double* ex = nodes[cd.expanded].pos;
      if (f.slot == 0) { fix_a.insert(fix_a.end(), ex, ex + 6); fix_b.insert(fix_b.end(), cd.pos, cd.pos + 6); }
      else {
        const Nb& nb = cd.nbs[f.slot - 1];
        const double* npos = nb.id >= 0 ? nodes[nb.id].pos : round_hpos + 6 * (size_t)(-1 - nb.id);
        if (nb.same_tree) { fix_a.insert(fix_a.end(), npos, npos + 6); fix_b.insert(fix_b.end(), cd.pos, cd.pos + 6); }
        else if (nb.id == goal_node && goal_node >= 0) { fix_a.insert(fix_a.end(), cd.pos, cd.pos + 6); fix_b.insert(fix_b.end(), npos, npos + 6); }
        else { fix_a.insert(fix_a.end(), ex, ex + 6); fix_b.insert(fix_b.end(), npos, npos + 6); }
      }
    }
    const int m = (int)fixes.size();
    std::vector<uint8_t> fr(m);
    std::vector<int32_t> fh(m), nsv(m);
    c.collide_segments(fix_a.data(), fix_b.data(), m, fr.data(), fh.data(), nsv.data());
    for (int k = 0; k < m; ++k) {
      Cand& cd = cands[fixes[k].cand];
      if (fixes[k].slot == 0) { cd.par_free = fr[k] != 0; cd.par_fh = fh[k]; cd.par_ns = nsv[k]; }
      else { Nb& nb = cd.nbs[fixes[k].slot - 1]; nb.free = fr[k] != 0; nb.fh = fh[k]; nb.ns = nsv[k]; }
    }
    wait_ms += ms_since(t0);
  }

  // ---- host path for the (rare) samples whose hit or neighbour lists overflowed on the device:
  // unbounded lists through the generic radius query, classification and batched checks here
  if (!slow.empty()) {
    auto t0 = Clock::now();
    const int BIG = 4096;
    const int m = (int)slow.size();
    std::vector<double> q6((size_t)m * 6), rr(m);
    std::vector<int32_t> mx(m), cn(m), ix((size_t)m * BIG);
    std::vector<double> dd((size_t)m * BIG);
    for (int k = 0; k < m; ++k) {
      int i = slow[k];
      memcpy(&q6[6 * (size_t)k], cands[i].pos, 6 * sizeof(double));
      rr[k] = std::max(cands[i].pdist, cfg.dist_tree);
      mx[k] = Tb + i;
    }
    int keep = c.store_n;
    c.store_n = Tb + n;  // include the temporaries
    c.radius(q6.data(), m, rr.data(), nullptr, mx.data(), ix.data(), dd.data(), cn.data(), BIG);
    c.store_n = keep;
    std::vector<double> pose_tasks, seg_a, seg_b;
    auto add_seg = [&](const double* a, const double* b) {
      int id = (int)(seg_a.size() / 6);
      seg_a.insert(seg_a.end(), a, a + 6);
      seg_b.insert(seg_b.end(), b, b + 6);
      return id;
    };
    for (int k = 0; k < m; ++k) {
      if (cn[k] > BIG) throw HipError{"forest: neighbour list overflow (> 4096 hits)"};
      Cand& cd = cands[slow[k]];
      cd.nbs.clear();   // (reset() leaves the list of an earlier round in place)
      const FNode& ex = nodes[cd.expanded];
      cd.pose_task = (int)(pose_tasks.size() / 6);
      pose_tasks.insert(pose_tasks.end(), cd.pos, cd.pos + 6);
      cd.seg_parent = add_seg(ex.pos, cd.pos);
      const int mine = ex.tree;
      std::vector<Nb> all;
      for (int j = 0; j < cn[k]; ++j) {
        const double d = dd[(size_t)k * BIG + j];
        const int id = ix[(size_t)k * BIG + j];
        Nb nb;
        nb.d = d;
        if (id < N0) {
          nb.id = id;
          nb.tree = nodes[id].tree;
          nb.order = nodes[id].idx_in_tree;
        } else {
          int cc = id - Tb;
          if (!round_hlim[cc]) continue;
          nb.id = -1 - cc;
          nb.tree = nodes[cands[cc].expanded].tree;
          nb.order = 0x40000000 + cc;
        }
        nb.same_tree = nb.tree == mine;
        nb.seg = -1;
        if (nb.same_tree) {
          if ((nflag[cd.expanded] & 1) || !(d < cd.pdist - SFFG_TOL)) continue;   // src/forest.h:276
        } else {
          if (!(d < cfg.dist_tree - SFFG_TOL)) continue;                    // src/forest.h:283
        }
        all.push_back(nb);
      }
      std::sort(all.begin(), all.end(), [](const Nb& a, const Nb& b) {
        if (a.tree != b.tree) return a.tree < b.tree;
        if (a.d != b.d) return a.d < b.d;
        return a.order < b.order;
      });
      // everything after the first STORE neighbour of another tree is unreachable (:296-299)
      for (Nb& nb : all) {
        const double* npos = nb.id >= 0 ? nodes[nb.id].pos : round_hpos + 6 * (size_t)(-1 - nb.id);
        if (nb.same_tree) nb.seg = add_seg(npos, cd.pos);     // isPathFree(neighbour, newPoint)  :276
        else if (nb.id == goal_node && goal_node >= 0) nb.seg = add_seg(cd.pos, npos);  // isPathFree(newPoint, goal) :287
        else nb.seg = add_seg(ex.pos, npos);                  // isPathFree(expanded, neighbour)  :288
        cd.nbs.push_back(nb);
        if (!nb.same_tree && nb.id >= 0) break;
      }
    }
    const int n_pose = (int)(pose_tasks.size() / 6), n_seg = (int)(seg_a.size() / 6);
    std::vector<uint8_t> pose_hit(n_pose), seg_free(n_seg);
    std::vector<int32_t> seg_fh(n_seg), seg_ns(n_seg);
    c.collide_poses(pose_tasks.data(), n_pose, pose_hit.data());
    c.collide_segments(seg_a.data(), seg_b.data(), n_seg, seg_free.data(), seg_fh.data(), seg_ns.data());
    st.poses_executed += n_pose;
    st.segments_executed += n_seg;
    for (int k = 0; k < n_seg; ++k) st.samples_executed += (uint64_t)seg_ns[k];
    for (int k = 0; k < m; ++k) {
      Cand& cd = cands[slow[k]];
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
    st.slow_path_samples += (uint64_t)m;
    wait_ms += ms_since(t0);
  }

  // ---- SFF* (src/forest.h:307-351): candidates that no STORE neighbour rejects may be accepted;
  // for them fetch the potential k-nearest set of their tree and answer both edge directions
  if (cfg.optimize) {
    std::vector<int> maybe;
    for (int i = 0; i < n; ++i) {
      Cand& cd = cands[i];
      if (round_skip[i] || !cd.answered || cd.pose_hit || !cd.par_free) continue;
      bool rejected = false;
      for (const Nb& nb : cd.nbs) {
        if (nb.id < 0) continue;
        if (nb.same_tree) { if (nb.free) { rejected = true; break; } }
        else if (cfg.has_goal && nb.id == goal_node) { if (!nb.free) { rejected = true; break; } }
        else { rejected = true; break; }
      }
      if (!rejected) maybe.push_back(i);
    }
    const int m = (int)maybe.size();
    if (m) {
      auto t0 = Clock::now();
      // k-nearest sets on the device (k_knn_grid): per sample one wavefront grows a cube of grid cells until its k-th
      // nearest STORE node of the tree lies inside the covered ball, keeping the k best in its lanes; the round's own
      // samples (ids >= Tb, earlier in the round) that are not farther than that k-th node come from the round grid.
      auto ts0 = Clock::now();
      const int KCAP = 64;
      c.h_f.ensure((size_t)m * sizeof(sffk::KnnQuery));
      sffk::KnnQuery* hq = c.h_f.as<sffk::KnnQuery>();
      std::vector<int32_t> kmax(m);
      for (int k = 0; k < m; ++k) {
        const Cand& cd = cands[maybe[k]];
        memcpy(hq[k].pos, cd.pos, sizeof hq[k].pos);
        hq[k].tree = nodes[cd.expanded].tree;
        hq[k].max_id = Tb + maybe[k];
        // k = 2e log10(#nodes) (:309) can only grow with the nodes accepted earlier in this round
        kmax[k] = (int32_t)(size_t)(2 * M_E * std::log10((double)(N0 + maybe[k])));
        if (kmax[k] > KCAP) throw HipError{"forest: k of the k-nearest set exceeds 64"};
        // (a tree with fewer nodes than that is wanted whole: asking for exactly its size lets the search stop as
        // soon as the last of them is found instead of growing the cube over the whole grid)
        hq[k].k = std::max(0, std::min<int>(kmax[k], (int)trees[hq[k].tree].size()));
        hq[k].mate_base = Tb;
        hq[k].whole_tree = (int)trees[hq[k].tree].size() <= kmax[k] ? 1 : 0;
      }
      const size_t q_b = (size_t)m * sizeof(sffk::KnnQuery);
      const size_t o_idx = 0, o_cnt = o_idx + (size_t)m * KCAP * 4, o_mate = o_cnt + (size_t)m * 4,
                   o_mcnt = o_mate + (size_t)m * SFFK_KNN_MATES * 4, o_end = o_mcnt + (size_t)m * 4;
      c.d_f.ensure(q_b);
      c.d_g.ensure(o_end);
      c.d_h.ensure((size_t)m * KCAP * 8);
      c.h_g.ensure(o_end);
      HIPCHK(hipMemcpyAsync(c.d_f.p, c.h_f.p, q_b, hipMemcpyHostToDevice, c.stream));
      char* dres = c.d_g.as<char>();
      c.time_begin(T_SWEEP);
      sffk::launch_knn_grid(c.stream, c.gridv, &c.tgridv, c.store_view(), c.d_f.as<sffk::KnnQuery>(), m, KCAP,
                            reinterpret_cast<int32_t*>(dres + o_idx), c.d_h.as<double>(), reinterpret_cast<int32_t*>(dres + o_cnt),
                            reinterpret_cast<int32_t*>(dres + o_mate), reinterpret_cast<int32_t*>(dres + o_mcnt), c.grid_cell,
                            8 * c.sweep_eps());
      c.time_end();
      HIPCHK(hipMemcpyAsync(c.h_g.p, c.d_g.p, o_end, hipMemcpyDeviceToHost, c.stream));
      timed_sync();
      g_star[3] += 1;
      st.sweeps += 1;
      st.sweep_nodes += (uint64_t)(N0 + n);
      st.sweep_queries += (uint64_t)m;
      g_star[0] += ms_since(ts0);
      auto ts1 = Clock::now();
      {
        const char* hres = c.h_g.as<char>();
        const int32_t* r_idx = reinterpret_cast<const int32_t*>(hres + o_idx);
        const int32_t* r_cnt = reinterpret_cast<const int32_t*>(hres + o_cnt);
        const int32_t* r_mate = reinterpret_cast<const int32_t*>(hres + o_mate);
        const int32_t* r_mcnt = reinterpret_cast<const int32_t*>(hres + o_mcnt);
        // (a dense young frontier under a large wave can put more than SFFK_KNN_MATES earlier samples of the round inside
        // one k-nearest ball, and a tree that is wanted whole has no ball at all: those queries are asked again with a
        // mate list that holds the whole round)
        std::vector<int> big;
        for (int k = 0; k < m; ++k)
          if (kmax[k] > 0 && r_mcnt[k] > SFFK_KNN_MATES) big.push_back(k);
        const int32_t* big_mates = nullptr;
        int big_mcap = 0;                  // (the first pass reports every query's true count: the largest one is enough)
        for (int k : big) big_mcap = std::max(big_mcap, (int)r_mcnt[k]);
        big_mcap = std::min(big_mcap, n);
        if (!big.empty()) {
          const int mb = (int)big.size(), mcap = big_mcap;
          c.h_e.ensure((size_t)mb * sizeof(sffk::KnnQuery));
          sffk::KnnQuery* bq = c.h_e.as<sffk::KnnQuery>();
          for (int j = 0; j < mb; ++j) bq[j] = hq[big[j]];
          const size_t b_idx = 0, b_cnt = b_idx + (size_t)mb * KCAP * 4, b_mate = b_cnt + (size_t)mb * 4,
                       b_mcnt = b_mate + (size_t)mb * mcap * 4, b_end = b_mcnt + (size_t)mb * 4;
          c.d_e.ensure((size_t)mb * sizeof(sffk::KnnQuery));
          c.d_c.ensure(b_end);
          c.d_d.ensure((size_t)mb * KCAP * 8);
          HIPCHK(hipMemcpyAsync(c.d_e.p, c.h_e.p, (size_t)mb * sizeof(sffk::KnnQuery), hipMemcpyHostToDevice, c.stream));
          char* db = c.d_c.as<char>();
          sffk::launch_knn_grid(c.stream, c.gridv, &c.tgridv, c.store_view(), c.d_e.as<sffk::KnnQuery>(), mb, KCAP,
                                reinterpret_cast<int32_t*>(db + b_idx), c.d_d.as<double>(), reinterpret_cast<int32_t*>(db + b_cnt),
                                reinterpret_cast<int32_t*>(db + b_mate), reinterpret_cast<int32_t*>(db + b_mcnt), c.grid_cell,
                                8 * c.sweep_eps(), mcap);
          HIPCHK(hipStreamSynchronize(c.stream));                 // (h_e holds the queries of the launch above)
          c.h_e.ensure(std::max((size_t)mb * sizeof(sffk::KnnQuery), ((size_t)mb * mcap + mb) * 4));   // (pinned: no staged copy)
          HIPCHK(hipMemcpyAsync(c.h_e.p, db + b_mate, ((size_t)mb * mcap + mb) * 4, hipMemcpyDeviceToHost, c.stream));
          timed_sync();
          big_mates = c.h_e.as<int32_t>();
          st.mate_overflow_requeries += (uint64_t)mb;
        }
        size_t big_at = 0;
        for (int k = 0; k < m; ++k) {
          Cand& cd = cands[maybe[k]];
          if (kmax[k] <= 0) continue;
          const int32_t* mate_list = r_mate + (size_t)k * SFFK_KNN_MATES;
          int mate_n = r_mcnt[k];
          if (mate_n > SFFK_KNN_MATES) {     // the second pass's list (same store neighbours, every mate)
            const int mb = (int)big.size(), mcap = big_mcap;
            mate_list = big_mates + big_at * mcap;
            mate_n = big_mates[(size_t)mb * mcap + big_at];
            if (mate_n != r_mcnt[k] || mate_n > mcap) throw HipError{"forest: k-nearest mate list of the second pass is inconsistent (internal error)"};
            ++big_at;
          }
          for (int q = 0; q < r_cnt[k]; ++q) {
            Member mb;
            mb.id = r_idx[(size_t)k * KCAP + q];
            cd.members.push_back(mb);
            cd.has_members = true;
          }
          for (int q = 0; q < mate_n; ++q) {
            Member mb;
            mb.id = -1 - (mate_list[q] - Tb);
            cd.members.push_back(mb);
            cd.has_members = true;
          }
        }
      }
      // both directions of every member edge, as pairs of store ids (the sample itself is the temporary store
      // entry Tb + i, a round-mate Tb + its index): the device gathers the positions
      std::vector<int32_t>& ia = edge_ia;
      std::vector<int32_t>& ib = edge_ib;
      ia.clear();
      ib.clear();
      for (int k = 0; k < m; ++k) {
        Cand& cd = cands[maybe[k]];
        const int32_t self = Tb + maybe[k];
        for (Member& mb : cd.members) {
          const int32_t other = mb.id >= 0 ? mb.id : Tb + (-1 - mb.id);
          mb.seg_f = (int)ia.size();
          ia.push_back(self);  ib.push_back(other);   // isPathFree(newPoint, neighbor)   :323
          mb.seg_b = (int)ia.size();
          ia.push_back(other); ib.push_back(self);    // isPathFree(neighbor, newPoint)   :336
        }
      }
      g_star[1] += ms_since(ts1);
      auto ts2 = Clock::now();
      const int ns2 = (int)ia.size();
      if (ns2) {
        std::vector<uint8_t> fr(ns2);
        std::vector<int32_t> fh(ns2), nsv(ns2);
        c.collide_segments_ids(ia.data(), ib.data(), ns2, fr.data(), fh.data(), nsv.data());
        st.segments_executed += ns2;
        for (int k = 0; k < ns2; ++k) st.samples_executed += (uint64_t)nsv[k];
        for (int k = 0; k < m; ++k)
          for (Member& mb : cands[maybe[k]].members) {
            mb.fwd_free = fr[mb.seg_f] != 0; mb.fwd_fh = fh[mb.seg_f]; mb.fwd_ns = nsv[mb.seg_f];
            mb.bwd_free = fr[mb.seg_b] != 0; mb.bwd_fh = fh[mb.seg_b]; mb.bwd_ns = nsv[mb.seg_b];
          }
      }
      g_star[2] += ms_since(ts2);
      wait_ms += ms_since(t0);
    }
  }

  if (cfg.optimize) sffk::launch_tgrid_clear(c.stream, tref);
  auto _t3 = Clock::now();
  // ---- samples whose fate this rank settled alone (k_settle): only their reference-equivalent counters travel;
  // the in-order replay skips them
  {
    for (int k = 0; k < 4; ++k) {
      records[2 + 2 * k] = (int32_t)(uint32_t)(bulk_counts[k] & 0xffffffffu);
      records[3 + 2 * k] = (int32_t)(uint32_t)(bulk_counts[k] >> 32);
    }
  }

  // ---- the int32 record stream of the owned candidates (only needed when there are other ranks)
  for (size_t j = 0; j < round_todo.size() && cfg.world > 1; ++j) {   // (still only this rank's samples)
    const int i = round_todo[j];
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
  g_sec[3] += ms_since(_t3);
  st.host_ms += ms_since(t_host) - wait_ms;
}

// all = concatenation of every rank's record stream (rank order), counts in int32 words
void Forest::round_commit(const int32_t* all, int total_words, const int32_t* counts, int world) {
  Ctx& c = *ctx;
  HIPCHK(hipSetDevice(c.device));
  if (!pending_round) throw HipError{"forest: round_commit without round_begin"};
  if (world != cfg.world) throw HipError{"forest: round_commit world size mismatch"};
  {   // the per-rank lengths must tile the supplied buffer exactly
    long long sum = 0;
    for (int r = 0; r < world; ++r) {
      if (counts[r] < 0) throw HipError{"forest: negative record stream length"};
      sum += counts[r];
    }
    if (sum != (long long)total_words) throw HipError{"forest: record stream lengths do not add up to the buffer size"};
  }
  auto t_host = Clock::now();
  double wait_ms = 0;
  const int n = n_cands;
  auto _t4 = Clock::now();
  // ---- absorb the other ranks' answers
  uint64_t settled_elsewhere = 0, recorded_elsewhere = 0;
  const size_t own_todo = round_todo.size();
  std::vector<size_t> run_bounds;   // end of every other rank's run in round_todo
  st.collide_calls += bulk_counts[0];   // this rank's own bulk-settled samples
  st.path_free_calls += bulk_counts[1];
  st.nn_queries += bulk_counts[2];
  size_t off = 0;
  for (int r = 0; r < world && world > 1; ++r) {
    const int32_t* p = all + off;
    const int32_t* end = p + counts[r];
    off += (size_t)counts[r];
    if (counts[r] < 10 || p[0] != REC_MAGIC || p[1] != n) throw HipError{"forest: ranks disagree on the round (diverged state)"};
    if (r != cfg.rank) {   // the other ranks' bulk-settled samples: counters only
      uint64_t v[4];
      for (int k = 0; k < 4; ++k) v[k] = (uint64_t)(uint32_t)p[2 + 2 * k] | ((uint64_t)(uint32_t)p[3 + 2 * k] << 32);
      st.collide_calls += v[0];
      st.path_free_calls += v[1];
      st.nn_queries += v[2];
      settled_elsewhere += v[3];
    }
    p += 10;
    while (p < end) {
      // a truncated or mismatched gather must end in a clean error, not in reads past the buffer
      if (end - p < 6) throw HipError{"forest: truncated record stream"};
      int i = p[0];
      if (i < 0 || i >= n || i % world != r) throw HipError{"forest: malformed record stream"};
      Cand& cd = cands[i];
      int flags = p[1], nn = p[4], nm = p[5];
      if (nn < 0 || nm < 0 || nn > (1 << 20) || nm > (1 << 20) || (size_t)(end - p) < 6 + 5 * (size_t)nn + 6 * (size_t)nm)
        throw HipError{"forest: malformed record (neighbour / member counts run past the stream)"};
      if (r != cfg.rank) {
        memcpy(cd.pos, round_hpos + 6 * (size_t)i, sizeof cd.pos);   // (sampling is replicated: every rank has them)
        cd.pdist = round_hpd[i];
        cd.in_lim = true;
        round_skip[i] = 0;
        round_todo.push_back(i);
        ++recorded_elsewhere;
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
        cd.has_members = nm > 0;
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
    if (r != cfg.rank) run_bounds.push_back(round_todo.size());
  }
  if (world > 1) {
    // the work list of the replay: own unsettled samples + every sample another rank sent a record for
    // (every rank's records come in ascending sample order: merge the runs)
    size_t sorted_end = own_todo;
    for (size_t rb : run_bounds) {
      std::inplace_merge(round_todo.begin(), round_todo.begin() + sorted_end, round_todo.begin() + rb);
      sorted_end = rb;
    }
    // every in-limit sample of another rank is either recorded or counted as settled there
    uint64_t inlim_all = 0, inlim_own = 0;
    for (int i = 0; i < n; ++i) inlim_all += round_hlim[i];
    for (int i = cfg.rank; i < n; i += world) inlim_own += round_hlim[i];
    const uint64_t inlim_elsewhere = inlim_all - inlim_own;
    if (inlim_elsewhere != recorded_elsewhere + settled_elsewhere)
      throw HipError{"forest: answer records missing (a rank did not report all of its samples)"};
  }
  g_sec[4] += ms_since(_t4);
  auto _t5 = Clock::now();
  if (getenv("SFFGPU_DIGEST")) {
    for (int i = 0; i < n; ++i) {
      const Cand& cd = cands[i];
      if (!cd.in_lim) continue;
      fprintf(stderr, "D %d %d | %d %d %d %d |", iter0, i, (int)cd.pose_hit, (int)cd.par_free, cd.par_fh, cd.par_ns);
      for (const Nb& nb : cd.nbs) fprintf(stderr, " nb(%d %d %d %d %d)", nb.id, (int)nb.same_tree, (int)nb.free, nb.fh, nb.ns);
      for (const Member& mb : cd.members) fprintf(stderr, " mb(%d %d %d %d %d %d %d)", mb.id, (int)mb.fwd_free, mb.fwd_fh, mb.fwd_ns, (int)mb.bwd_free, mb.bwd_fh, mb.bwd_ns);
      fprintf(stderr, "\n");
    }
  }
  // ---- replay expandNode in slot order (src/forest.h:240-376)
  auto calls = [](int fh, int ns) -> uint64_t {  // Collide calls isPathFree makes (early exit at the first hit)
    return fh > 0 ? (uint64_t)fh : (uint64_t)ns;
  };
  std::vector<double> app_pos;
  std::vector<int32_t> app_tree;
  // (samples outside the limits (:246 !result) and the ones settled by their owner are not on the work list)
  const size_t n_todo = round_todo.size();
  int last_i = -1;
  for (size_t j = 0; j < n_todo; ++j) {
    if (solved) break;     // goal reached by an earlier slot of this round: the remaining slots are not run
    if (j + 12 < n_todo) __builtin_prefetch(&cands[round_todo[j + 12]]);
    if (j + 6 < n_todo) __builtin_prefetch(&nodes[cands[round_todo[j + 6]].expanded].tree);
    const int i = round_todo[j];
    last_i = i;
    Cand& cd = cands[i];
    Slot& sl = slots[cd.slot];
    const unsigned iteration = (unsigned)(iter0 + i + 1);
    if (!cd.answered) throw HipError{"forest: a candidate has no answer record"};
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
      if (nb.same_tree) {
        st.path_free_calls += 1;
        st.collide_calls += calls(nb.fh, nb.ns);
        if (nb.free) { reject = true; break; }                 // :276-280 overcrowded
      } else if (cfg.has_goal) {
        if (nb_node == goal_node) {                            // :286-287 goal reached?
          st.path_free_calls += 1;
          st.collide_calls += calls(nb.fh, nb.ns);
          solved = nb.free;
        }
        if (!solved) { reject = true; break; }                 // :296-299
      } else {
        st.path_free_calls += 1;
        st.collide_calls += calls(nb.fh, nb.ns);
        if (nb.free) {                                         // :288-294
          int a = std::min(nb_node, expanded), b = std::max(nb_node, expanded);
          // (two nodes belong to one tree pair only, so the pair itself identifies the list entry)
          if (border_keys.insert(((uint64_t)(uint32_t)a << 32) | ((uint64_t)(uint32_t)b + 1))) {   // (+1: never 0)
            double d = nodes[nb_node].d_root + nodes[expanded].d_root + sffg::dist6(nodes[nb_node].pos, nodes[expanded].pos);
            border(nb.tree, mine).push_back({a, b, d});
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
      static thread_local std::vector<KN> knn;   // (scratch: one allocation for the whole run)
      knn.clear();
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
        double nd = kn.d + nodes[kn.node].d_root;               // kn.d = dist6(new, node), computed above
        if (nd < best - SFFG_TOL) {
          st.path_free_calls += 1;
          st.collide_calls += calls(kn.mb->fwd_fh, kn.mb->fwd_ns);
          if (kn.mb->fwd_free) { best = nd; parent = kn.node; }
        }
      }
      id = add_node(cd.pos, mine, parent, sffg::dist6(cd.pos, nodes[parent].pos), best, iteration);  // :329
      if (cfg.record_parents) hist.push_back({id, parent, iteration});
      for (const KN& kn : knn) {                               // :332-350
        double npd = kn.d;   // dist6(node, new) == dist6(new, node) bit for bit (squares of exactly negated terms)
        double proposed = best + npd;
        if (proposed < nodes[kn.node].d_root - SFFG_TOL) {
          st.path_free_calls += 1;
          st.collide_calls += calls(kn.mb->bwd_fh, kn.mb->bwd_ns);
          if (kn.mb->bwd_free) {
            nodes[kn.node].parent = id;
            if (cfg.record_parents) hist.push_back({kn.node, id, iteration});
            nodes[kn.node].d_closest = npd;
            nodes[kn.node].d_root = proposed;                  // descendants keep their old cost (Appendix A.6)
          }
        }
      }
      st.nn_queries += 1;                                      // :317 knnSearch
    } else {
      id = add_node(cd.pos, mine, expanded, cd.pdist, cd.pdist + nodes[expanded].d_root, iteration);  // :353
      if (cfg.record_parents) hist.push_back({id, expanded, iteration});
    }
    cd.accepted_id = id;
    if (use_priority()) {                                      // :360-363
      for (PHeap& h : heaps[mine]) h.push(id);
    } else {
      frontier.push_back(id);                                  // :365
      nflag[id] |= 2;
    }
    if (solved) {                                              // :369-372
      double gd = sffg::dist6(cd.pos, cfg.goal);
      border_keys.insert(((uint64_t)(uint32_t)std::min(id, goal_node) << 32) | ((uint64_t)(uint32_t)std::max(id, goal_node) + 1));
      border(num_roots - 1, mine).push_back({std::min(id, goal_node), std::max(id, goal_node), nodes[id].d_root + gd});
    }
    sl.failing = false;
    app_pos.insert(app_pos.end(), cd.pos, cd.pos + 6);
    app_tree.push_back(mine);
  }
  if (solved) iter = iter0 + last_i + 1;   // the iterations after the solving one were never run
  g_sec[5] += ms_since(_t5);
  auto _t6 = Clock::now();
  // ---- commit the accepted nodes to the device store (replaces flannIndex->addPoints, :367)
  if (n > 0) c.store_n = N0;
  if (!app_tree.empty()) {
    auto t0 = Clock::now();
    c.store_append(app_pos.data(), app_tree.data(), (int)app_tree.size(), /*wait=*/false);
    c.grid_insert_new();
    wait_ms += ms_since(t0);
  }
  pending_round = false;
  g_sec[6] += ms_since(_t6);
  auto _t7 = Clock::now();
  // ---- wave bookkeeping (src/forest.h:155: at most ThresholdMisses attempts per slot)
  bool any_failing = false;
  for (const Slot& s : slots) any_failing |= s.failing;
  if (round >= cfg.threshold_misses || !any_failing || solved || iter >= cfg.max_iterations) end_wave();
  g_sec[7] += ms_since(_t7);
  st.host_ms += ms_since(t_host) - wait_ms;
}

void Forest::run(int max_waves) {
  if (cfg.world != 1) {
    // the library's own RCCL exchange (sffgpu_ctx_rccl_init) drives a sharded device-engine forest by itself
    if (!(dev.on && ctx->can_exchange(cfg.rank, cfg.world)))
      throw HipError{"forest: run() drives a single-GPU forest, or a device-engine forest on a context with an RCCL "
                     "communicator (or a caller's all-gather) of the same rank / world; otherwise use round_begin/round_commit"};
  }
  if (dev.on) { run_device(max_waves); return; }
  auto t0 = Clock::now();
  const uint64_t w0 = st.waves;
  while (true) {
    if (!in_wave) {
      if (terminated()) break;
      if (max_waves > 0 && (int)(st.waves - w0) >= max_waves) break;
    }
    round_begin();
    int32_t cnt = (int32_t)records.size();
    round_commit(records.data(), cnt, &cnt, 1);
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
