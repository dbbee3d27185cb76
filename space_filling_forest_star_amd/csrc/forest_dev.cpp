// forest_dev.cpp — host driver of the device-resident SFF engine (devforest.hip).
//
// The forest's authoritative state (node records, frontier, closed list, slots, borders, counters, RNG cursor)
// lives in HBM while this engine runs.  Per wave the host enqueues  k_wave_begin, ThresholdMisses x {sample, query,
// classify, compact, cull, exact, settle, resolve}, k_wave_end  and one 256-byte status copy, then waits once.  While
// the GPU works on wave w the host generates and uploads the engine words wave w+1 may need.  The host mirror
// (Forest::nodes, frontier, borders ...) is refreshed lazily: only when a getter, the fault path or a caller of the
// round protocol needs it.
//
// Faults: a round in which a bounded device list overflowed is not committed by k_commit; the host downloads the
// state, finishes that wave on the host path of forest.cpp (unbounded lists) and uploads the result.  A full border
// table or node / border arrays that need to grow only cost a reallocation on the host and a resumed wave.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "engine.h"
#include "sff_geom.h"

namespace sff {

#define HIPCHK(x) hip_check((x), #x)
using Clock = std::chrono::steady_clock;
static double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }

static uint64_t next_pow2(uint64_t v) {
  uint64_t p = 1;
  while (p < v) p <<= 1;
  return p;
}

bool Forest::device_eligible() const {
  // plain SFF and SFF* (choose-parent + rewire on the device: devstar.hip), the single-goal mode, and the priority-frontier
  // mode without a goal (heaps in HBM, one workgroup per heap: devprio.hip); priority + goal runs on the host-replay engine
  if (!use_priority()) return true;
  // (measured on dense_3D, 100 k nodes: 73 k nodes/s at waves of 64 slots against the host engine's 29 k, 0.83 M at 1 024
  // against 98 k, 1.6 M at 8 192; waves of one slot - the reference's own loop - stay on the host engine;
  // SFFGPU_PRIO_DEVICE=0 keeps the whole mode there)
  const char* const knob = getenv("SFFGPU_PRIO_DEVICE");
  const int min_wave = (knob && !atoi(knob)) ? 0x7fffffff : 2;
  if (cfg.has_goal || cfg.wave < min_wave || cfg.world > 1) return false;
  const long long heaps_n = (long long)num_roots * (num_roots - 1);
  const long long cap = (long long)std::max(cfg.node_budget, 4096) + 2LL * cfg.wave + 128;
  return heaps_n >= 1 && heaps_n <= SFFK_PRIO_MAX_HEAPS && heaps_n * cap * 16 <= (32LL << 30);   // (ids + keys + position map)
}

// ---- priority-frontier mode: the trees' heaps between the host mirror (PHeap) and the device arrays
void Forest::dev_prio_upload(int gen) {
  DevEngine& d = dev;
  Ctx& c = *ctx;
  int H = 0;
  std::vector<int32_t> base(heaps.size() + 1, 0);
  for (size_t t = 0; t < heaps.size(); ++t) { base[t] = H; H += (int)heaps[t].size(); }
  base[heaps.size()] = H;
  // entries per heap (and the range of its position map): what the forest can hold - its node budget plus the wave that
  // may run past it - not the store's capacity, which an earlier user of the context may have left very large
  const int cap = cfg.node_budget > 0 ? std::min(d.node_cap, cfg.node_budget + 2 * cfg.wave + 256) : d.node_cap;
  d.prio_heaps = H;
  d.prio_cap = cap;
  d.hp_base.ensure(base.size() * 4);
  d.hp_size.ensure((size_t)H * 4);
  d.hp_gen.ensure((size_t)H * 4);
  d.hp_cnt.ensure(16);
  d.hp_ref.ensure((size_t)H * 48);
  d.hp_v.ensure((size_t)H * cap * 4);
  d.hp_key.ensure((size_t)H * cap * 8);
  d.hp_pos.ensure((size_t)H * cap * 4);
  d.slot_tree.ensure((size_t)cfg.wave * 4);
  d.slot_heap.ensure((size_t)cfg.wave * 4);
  d.slot_idx.ensure((size_t)cfg.wave * 4);
  d.slot_word.ensure((size_t)cfg.wave * 8);
  d.hp_plan.ensure(64);     // (k_prio_plan works in LDS; the pointer says "plan in parallel")
  std::vector<int32_t> sizes(H);
  std::vector<double> refs((size_t)H * 6);
  HIPCHK(hipMemsetAsync(d.hp_pos.p, 0xFF, (size_t)H * cap * 4, c.stream));
  HIPCHK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(d.hp_gen.p), gen, (size_t)H, c.stream));
  HIPCHK(hipMemsetAsync(d.hp_cnt.p, 0, 16, c.stream));
  std::vector<double> keys;
  int h = 0;
  for (size_t t = 0; t < heaps.size(); ++t)
    for (PHeap& hp : heaps[t]) {
      sizes[h] = (int32_t)hp.v.size();
      memcpy(&refs[6 * (size_t)h], hp.ref, 48);
      if (!hp.v.empty()) {
        keys.resize(hp.v.size());
        for (size_t i = 0; i < hp.v.size(); ++i) keys[i] = hp.cost((int)i);
        HIPCHK(hipMemcpy(d.hp_v.as<int32_t>() + (size_t)h * cap, hp.v.data(), hp.v.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d.hp_key.as<double>() + (size_t)h * cap, keys.data(), keys.size() * 8, hipMemcpyHostToDevice));
      }
      ++h;
    }
  HIPCHK(hipMemcpy(d.hp_base.p, base.data(), base.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d.hp_size.p, sizes.data(), (size_t)H * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d.hp_ref.p, refs.data(), refs.size() * 8, hipMemcpyHostToDevice));
  sffk::launch_prio_index(c.stream, dev_view().prio);
  HIPCHK(hipStreamSynchronize(c.stream));
}
// (keys and entries stay on the device: the host mirror may not even know the newest nodes)
void Forest::dev_prio_regrow() {
  DevEngine& d = dev;
  Ctx& c = *ctx;
  const int H = d.prio_heaps, old_cap = d.prio_cap;
  const int cap = cfg.node_budget > 0 ? std::min(d.node_cap, cfg.node_budget + 2 * cfg.wave + 256) : d.node_cap;
  if (!H || cap <= old_cap) return;
  DevBuf nv, nk;
  nv.ensure((size_t)H * cap * 4);
  nk.ensure((size_t)H * cap * 8);
  HIPCHK(hipMemcpy2DAsync(nv.p, (size_t)cap * 4, d.hp_v.p, (size_t)old_cap * 4, (size_t)old_cap * 4, H, hipMemcpyDeviceToDevice, c.stream));
  HIPCHK(hipMemcpy2DAsync(nk.p, (size_t)cap * 8, d.hp_key.p, (size_t)old_cap * 8, (size_t)old_cap * 8, H, hipMemcpyDeviceToDevice, c.stream));
  HIPCHK(hipStreamSynchronize(c.stream));
  d.hp_v.release(); d.hp_key.release(); d.hp_pos.release();
  d.hp_v = nv; d.hp_key = nk;
  nv.p = nullptr; nv.cap = 0; nk.p = nullptr; nk.cap = 0;
  d.hp_pos.ensure((size_t)H * cap * 4);
  d.prio_cap = cap;
  HIPCHK(hipMemsetAsync(d.hp_pos.p, 0xFF, (size_t)H * cap * 4, c.stream));
  sffk::launch_prio_index(c.stream, dev_view().prio);
  HIPCHK(hipStreamSynchronize(c.stream));
}
void Forest::dev_prio_download() {
  DevEngine& d = dev;
  if (!d.prio_heaps) return;
  const int H = d.prio_heaps, cap = d.prio_cap;
  std::vector<int32_t> sizes(H);
  HIPCHK(hipMemcpy(sizes.data(), d.hp_size.p, (size_t)H * 4, hipMemcpyDeviceToHost));
  int h = 0;
  for (size_t t = 0; t < heaps.size(); ++t)
    for (PHeap& hp : heaps[t]) {
      hp.v.resize((size_t)sizes[h]);
      if (sizes[h]) HIPCHK(hipMemcpy(hp.v.data(), d.hp_v.as<int32_t>() + (size_t)h * cap, (size_t)sizes[h] * 4, hipMemcpyDeviceToHost));
      ++h;
    }
}

sffk::StarView Forest::star_view() const {
  const DevEngine& d = dev;
  sffk::StarView v{};
  v.ktab = d.s_ktab.as<int32_t>();
  v.tree_cnt = d.s_tree_cnt.as<int32_t>();
  v.head = d.s_head.as<unsigned long long>();
  v.m_cnt = d.s_mcnt.as<int32_t>();
  v.m_id = d.s_mid.as<int32_t>();
  v.m_d = d.s_md.as<double>();
  v.next = d.s_next.as<int32_t>();
  v.prop = d.s_prop.as<double>();
  v.best = d.s_best.as<double>();
  v.psel = d.s_psel.as<int32_t>();
  v.dcl = d.s_dcl.as<double>();
  v.cnt = d.s_cnt.as<unsigned long long>();
  v.acc_sample = d.s_accs.as<int32_t>();
  v.hdr = d.s_hdr.as<int32_t>();
  v.changed = d.s_changed.as<int32_t>();
  v.ew = d.s_ew.as<int32_t>();
  v.ens = d.s_segns.as<int32_t>();
  v.first_hit = d.s_fh.as<int32_t>();
  v.seg_ovf = d.s_sovf.as<int32_t>();
  v.ida = d.s_ida.as<int32_t>();
  v.idb = d.s_idb.as<int32_t>();
  v.sub = d.s_sub.as<int32_t>();
  v.items = d.s_items.p;
  v.items_cap = d.s_items_cap;
  v.ev_sample = d.s_evs.as<int32_t>();
  v.ev_nb = d.s_evn.as<int32_t>();
  v.ev_ex = d.s_eve.as<int32_t>();
  v.ev_dist = d.s_evd.as<double>();
  v.acc = d.s_acc.as<unsigned long long>();
  v.backup = d.s_backup.as<sffk::DevCtrl>();
  v.dbg = d.s_dbg.as<unsigned long long>();
  if (cfg.record_parents) {
    v.hist = d.s_hist.as<int32_t>() + 4;
    v.hist_ctl = d.s_hist.as<int32_t>();
    v.hist_cap = d.hist_cap;
  }
  return v;
}

void Forest::dev_star_setup() {
  DevEngine& d = dev;
  const size_t W = (size_t)cfg.wave, KC = SFFK_STAR_KC;
  if (!d.star_inited) {
    // k = (size_t)(2e log10(#nodes)) (src/forest.h:309): the node counts at which it steps, found with the C library's
    // log10 in the reference's own expression (the kernels only compare integers)
    std::vector<int32_t> ktab(64, 0x7fffffff);
    auto k_of = [](long long nn) { return (long long)(size_t)(2 * M_E * std::log10((double)nn)); };
    ktab[0] = 0;
    for (int m = 1; m < 64; ++m) {
      long long lo = 1, hi = 0x7fffffffLL;
      if (k_of(hi) < m) { ktab[m] = 0x7fffffff; continue; }
      while (lo < hi) {
        const long long mid = (lo + hi) / 2;
        if (k_of(mid) >= m) hi = mid; else lo = mid + 1;
      }
      ktab[m] = (int32_t)lo;
    }
    d.s_ktab.ensure(64 * 4);
    HIPCHK(hipMemcpy(d.s_ktab.p, ktab.data(), 64 * 4, hipMemcpyHostToDevice));
    d.s_tree_cnt.ensure((size_t)num_roots * 16 * 4);
    d.s_mcnt.ensure(W * 4);
    d.s_mid.ensure(W * KC * 4);
    d.s_md.ensure(W * KC * 8);
    d.s_next.ensure(W * KC * 4);
    d.s_prop.ensure(W * KC * 8);
    d.s_best.ensure(W * 8);
    d.s_psel.ensure(W * 4);
    d.s_dcl.ensure(W * 8);
    d.s_cnt.ensure(W * 16);
    d.s_accs.ensure(W * 4);
    d.s_hdr.ensure(64);
    d.s_changed.ensure(256);
    d.s_ew.ensure(W * KC * 2 * 4);
    d.s_ida.ensure(W * KC * 2 * 4);
    d.s_idb.ensure(W * KC * 2 * 4);
    d.s_sub.ensure((size_t)SFFK_STAR_PASSES * SFFK_SUBLISTS * SFFK_STAR_SUB * 4);
    d.s_segns.ensure(W * KC * 2 * 4);
    d.s_fh.ensure(W * KC * 2 * 4);
    d.s_sovf.ensure(W * KC * 2 * 4);
    d.s_evs.ensure(W * 4);
    d.s_evn.ensure(W * 4);
    d.s_eve.ensure(W * 4);
    d.s_evd.ensure(W * 8);
    d.s_acc.ensure(64 * SFFK_STAR_ACC * 8);
    d.s_backup.ensure(sizeof(sffk::DevCtrl));
    // (survivor items of ONE pass: the chunks of the reachable member edges the clearance bits leave open)
    d.s_items_cap = (int)std::min<size_t>(16 * W + 65536, (size_t)1 << 26) / SFFK_SUBLISTS * SFFK_SUBLISTS;
    if (const char* e = getenv("SFFGPU_TEST_STAR_ITEMS")) d.s_items_cap = std::max(SFFK_SUBLISTS, atoi(e)) / SFFK_SUBLISTS * SFFK_SUBLISTS;
    d.s_items.ensure((size_t)d.s_items_cap * sizeof(sffk::SurvivorItem));
    HIPCHK(hipMemset(d.s_sub.p, 0, d.s_sub.cap));
    if (getenv("SFFGPU_PROFILE")) {
      d.s_dbg.ensure(32 * 8);
      HIPCHK(hipMemset(d.s_dbg.p, 0, 32 * 8));
    }
    HIPCHK(hipMemset(d.s_hdr.p, 0, 64));
    HIPCHK(hipMemset(d.s_changed.p, 0, 256));
    HIPCHK(hipMemset(d.s_acc.p, 0, 64 * SFFK_STAR_ACC * 8));
    d.star_inited = true;
  }
  if (cfg.record_parents) {   // (device entries start at 0 with every upload: the host's list holds everything before)
    d.hist_cap = std::max(d.hist_cap, 4 * ctx->store_cap + 4 * cfg.wave);
    d.s_hist.ensure(((size_t)3 * d.hist_cap + 4) * 4);
    HIPCHK(hipMemset(d.s_hist.p, 0, 16));
  }
  // (the control block's epoch restarts with every upload: no list head of an earlier stay on the device may match it)
  if (d.s_head.p) HIPCHK(hipMemset(d.s_head.p, 0, d.s_head.cap));
  std::vector<int32_t> tc((size_t)num_roots * 16, 0);
  for (int t = 0; t < num_roots; ++t) tc[(size_t)t * 16] = (int32_t)trees[t].size();
  HIPCHK(hipMemcpy(d.s_tree_cnt.p, tc.data(), tc.size() * 4, hipMemcpyHostToDevice));
}

sffk::DevForestView Forest::dev_view() const {
  const DevEngine& d = dev;
  sffk::DevForestView v{};
  v.ctrl = d.ctrl.as<sffk::DevCtrl>();
  v.parent = d.parent.as<int32_t>();
  v.d_root = d.d_root.as<double>();
  v.d_closest = d.d_closest.as<double>();
  v.iter = d.iter.as<uint32_t>();
  v.nflag = d.nflag.as<uint8_t>();
  v.frontier = d.frontier.as<int32_t>();
  v.frontier2 = d.frontier2.as<int32_t>();
  v.rm_words = d.rm_words.as<unsigned long long>();
  v.rm_pref = d.rm_pref.as<int32_t>();
  v.slot_pos = d.slot_pos.as<int32_t>();
  v.closed = d.closed.as<int32_t>();
  v.claim = d.claim.as<int32_t>();
  v.slot_node = d.slot_node.as<int32_t>();
  v.act_slot = d.act_slot.as<int32_t>();
  v.act_slot2 = d.act_slot2.as<int32_t>();
  v.w_acc = d.w_acc.as<unsigned long long>();
  v.acc_pref = d.acc_pref.as<int32_t>();
  v.w_ev = d.w_ev.as<unsigned long long>();
  v.ev_h = d.ev_h.as<unsigned long long>();
  v.ev_nb = d.ev_nb.as<int32_t>();
  v.ev_raw = d.ev_raw.as<int32_t>();
  v.ustate32 = d.ustate32.as<int32_t>();
  v.wg_pub = d.wg_pub.as<unsigned long long>();
  v.commit_seq = d.commit_seq.as<int32_t>();
  v.goal_id = cfg.has_goal ? goal_node : -1;
  {
    // spatial order of the wave's slots (sffk::OrderView): plain frontier picks only - in priority mode the slots' nodes
    // come from k_prio_begin - and only with the node grid in place; SFFGPU_NO_ORDER=1 switches it off
    const sffk::GridView& g = ctx->gridv;
    // (waves below 4 096 slots: a few hundred samples per round share one XCD's L2 anyway, the bookkeeping only costs)
    if (d.ord_enabled && !use_priority() && d.ord_hist.p && g.cnt && g.nx > 0 && cfg.wave >= d.ord_min_wave) {
      sffk::OrderView& o = v.ord;
      o.hist = d.ord_hist.as<int32_t>(); o.start = d.ord_start.as<int32_t>();
      o.slot_key = d.ord_key.as<int32_t>(); o.slot_rank = d.ord_rank.as<int32_t>();
      o.slot_pos = d.ord_pos.as<int32_t>();
      o.n_sub = (cfg.wave + 63) / 64;
      o.lst[0] = d.ord_lst.as<int32_t>(); o.lst[1] = o.lst[0] + (size_t)o.n_sub * 64;
      o.cnt[0] = d.ord_cnt.as<int32_t>(); o.cnt[1] = o.cnt[0] + (size_t)o.n_sub * SFFK_ORD_CNT_STRIDE;
      o.pos = ctx->spos.as<double>();
      o.ox = g.ox; o.oy = g.oy; o.oz = g.oz; o.inv_cell = g.inv_cell;
      o.nx = g.nx; o.ny = g.ny; o.nz = g.nz;
      int sh = 0;
      auto coarse = [&](int n) { return ((n - 1) >> sh) + 1; };
      while ((long long)coarse(g.nx) * coarse(g.ny) * coarse(g.nz) > SFFK_ORD_BUCKETS) ++sh;
      o.shift = sh; o.cnx = coarse(g.nx); o.cny = coarse(g.ny);
      o.n_buckets = coarse(g.nx) * coarse(g.ny) * coarse(g.nz);
    }
  }
  if (use_priority() && d.prio_heaps) {
    v.prio.n_heaps = d.prio_heaps; v.prio.cap = d.prio_cap;
    v.prio.base = d.hp_base.as<int32_t>(); v.prio.size = d.hp_size.as<int32_t>(); v.prio.v = d.hp_v.as<int32_t>();
    v.prio.key = d.hp_key.as<double>(); v.prio.pos = d.hp_pos.as<int32_t>(); v.prio.ref = d.hp_ref.as<double>();
    v.prio.slot_tree = d.slot_tree.as<int32_t>(); v.prio.slot_heap = d.slot_heap.as<int32_t>(); v.prio.slot_idx = d.slot_idx.as<int32_t>();
    v.prio.counters = d.hp_cnt.as<int32_t>(); v.prio.gen = d.hp_gen.as<int32_t>();
    v.prio.slot_word = d.slot_word.as<unsigned long long>();
    static const bool seq_only = getenv("SFFGPU_PRIO_SEQ") && atoi(getenv("SFFGPU_PRIO_SEQ")) != 0;   // (tests: the sequential picks)
    v.prio.plan = seq_only ? nullptr : d.hp_plan.as<int32_t>();
    v.prio.bias = cfg.priority_bias;
  }
  static const int profile = getenv("SFFGPU_PROFILE") ? 1 : 0;
  v.profile = profile;
  v.qclk_sh = d.qclk_sh.as<unsigned long long>();
  v.host_status = (d.zc_status && d.h_ctrl.p) ? d.h_ctrl.as<sffk::DevCtrl>() : nullptr;   // (hipHostMalloc: one address on both sides)
  v.kc_trace = d.kc_trace.as<unsigned long long>();
  v.kc_trace_round = getenv("SFFGPU_KC_TRACE") ? atoi(getenv("SFFGPU_KC_TRACE")) : -1;
  v.b_n1 = d.b_n1.as<int32_t>();
  v.b_n2 = d.b_n2.as<int32_t>();
  v.b_ta = d.b_ta.as<int32_t>();
  v.b_tb = d.b_tb.as<int32_t>();
  v.b_dist = d.b_dist.as<double>();
  v.bt_key = d.bt_key.as<unsigned long long>();
  v.bt_val = d.bt_val.as<unsigned long long>();
  v.bt_mask = d.bt_size - 1;
  v.pair = d.pair.as<uint8_t>();
  v.ring = d.ring.as<uint64_t>();
  v.ring_mask = d.ring_words - 1;
  v.node_cap = d.node_cap;
  v.border_cap = d.border_cap;
  v.wave = cfg.wave;
  v.n_trees = num_roots;
  v.words_per = cfg.dim == 2 ? 1 : 6;
  v.threshold_misses = cfg.threshold_misses;
  v.max_iterations = cfg.max_iterations;
  v.node_budget = cfg.node_budget;
  v.temp_base = d.temp_base;
  v.ulist = d.ulist.as<int32_t>();
  return v;
}

// (re)allocates the per-node arrays for the store's current capacity and places the temporaries of a round behind
// every node the forest can hold
void Forest::dev_size_node_arrays() {
  Ctx& c = *ctx;
  DevEngine& d = dev;
  const int cap = c.store_cap;
  d.node_cap = cap - cfg.wave - 16;                    // nodes live in [0, node_cap), temporaries behind them
  d.temp_base = (cap - cfg.wave - 8) & ~3;
  if (d.node_cap < 64) throw HipError{"forest: node store too small for the device engine"};
  d.parent.ensure((size_t)cap * 4);
  d.d_root.ensure((size_t)cap * 8);
  d.d_closest.ensure((size_t)cap * 8);
  d.iter.ensure((size_t)cap * 4);
  d.nflag.ensure((size_t)cap);
  d.frontier.ensure((size_t)cap * 4);
  d.frontier2.ensure((size_t)cap * 4);
  const size_t old_rm = d.rm_words.cap;
  d.rm_words.ensure(((size_t)cap / 64 + 2) * 8);
  if (d.rm_words.cap != old_rm)   // (all zero between waves: k_frontier_compact clears what k_wave_end marked)
    HIPCHK(hipMemsetAsync(d.rm_words.p, 0, d.rm_words.cap, c.stream));
  d.rm_pref.ensure(((size_t)cap / 64 + 2) * 4);
  d.closed.ensure((size_t)cap * 4);
  if (cfg.optimize) {   // SFF*: heads of the per-node toucher lists (stamped with the round's epoch; 0 = never used)
    const size_t old_head = d.s_head.cap;
    d.s_head.ensure((size_t)cap * 8);
    if (d.s_head.cap != old_head) HIPCHK(hipMemsetAsync(d.s_head.p, 0, d.s_head.cap, c.stream));
  }
  if (cfg.record_parents && d.s_hist.p && 4 * cap + 4 * cfg.wave > d.hist_cap) {   // (grows with the store; entries are kept)
    d.hist_cap = 4 * cap + 4 * cfg.wave;
    d.s_hist.ensure(((size_t)3 * d.hist_cap + 4) * 4);
  }
  const size_t old_claim = d.claim.cap;
  d.claim.ensure((size_t)cap * 4);
  if (d.claim.cap != old_claim)   // (fresh part must read "unclaimed"; the whole array is unclaimed between waves)
    HIPCHK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(d.claim.p), 0x7fffffff, d.claim.cap / 4, c.stream));
}

void Forest::dev_size_border_arrays(int want_cap) {
  Ctx& c = *ctx;
  DevEngine& d = dev;
  if (want_cap > d.border_cap) {
    d.border_cap = want_cap;
    d.b_n1.ensure((size_t)want_cap * 4);
    d.b_n2.ensure((size_t)want_cap * 4);
    d.b_ta.ensure((size_t)want_cap * 4);
    d.b_tb.ensure((size_t)want_cap * 4);
    d.b_dist.ensure((size_t)want_cap * 8);
  }
  const uint64_t want_tab = next_pow2((uint64_t)std::max(getenv("SFFGPU_TEST_BORDER_CAP") ? 64 : 1 << 16, 4 * d.border_cap));
  if (want_tab > d.bt_size) {
    d.bt_size = want_tab;
    d.bt_key.release();
    d.bt_val.release();
    d.bt_key.ensure((size_t)want_tab * 8);
    d.bt_val.ensure((size_t)want_tab * 8);
    HIPCHK(hipMemsetAsync(d.bt_key.p, 0, (size_t)want_tab * 8, c.stream));
    HIPCHK(hipMemsetAsync(d.bt_val.p, 0xFF, (size_t)want_tab * 8, c.stream));
    d.table_dirty = true;   // the list entries have to be re-inserted
  }
}

// ---- engine words: rng is the generator; in device mode it holds no queue and rng.draws == dev.produced
void Forest::dev_ring_append(const uint64_t* words, size_t n) {   // words for absolute positions [produced, produced + n)
  DevEngine& d = dev;
  Ctx& c = *ctx;
  uint64_t* hr = d.h_ring.as<uint64_t>();
  size_t done = 0;
  while (done < n) {
    const uint64_t at = (d.produced + done) & (d.ring_words - 1);
    const size_t run = std::min<size_t>(n - done, (size_t)(d.ring_words - at));
    if (words) memcpy(hr + at, words + done, run * 8);
    else rng.fill(hr + at, run);
    HIPCHK(hipMemcpyAsync(d.ring.as<uint64_t>() + at, hr + at, run * 8, hipMemcpyHostToDevice, c.copy_stream));
    if (cfg.libm_sampling) {
      // parity mode: RandGen::randomPointInDistance's transcendental functions (src/randGen.h:78-100) are evaluated
      // HERE, by the C library the reference itself calls.  Which word of the stream becomes which angle is only known on
      // the device (frontier picks and samples share the stream), so every word gets all three values it could be asked
      // for: cos / sin of the word as phi or theta, acos of the word as the pitch draw.
      double* ht = d.h_trig.as<double>() + 3 * at;
      const uint64_t* hw = hr + at;
      const unsigned hw_threads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
      const size_t per = (run + hw_threads - 1) / hw_threads;
      auto work = [&](size_t b, size_t e) {
        for (size_t j = b; j < e; ++j) {
          const double ang = sffg::sample_angle(hw[j]);
          ht[3 * j] = std::cos(ang);
          ht[3 * j + 1] = std::sin(ang);
          ht[3 * j + 2] = std::acos(sffg::sample_acos_arg(hw[j]));
        }
      };
      if (run < 4096 || hw_threads == 1) work(0, run);
      else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < hw_threads; ++t) {
          const size_t b = std::min(run, t * per), e = std::min(run, b + per);
          if (b < e) th.emplace_back(work, b, e);
        }
        for (auto& x : th) x.join();
      }
      HIPCHK(hipMemcpyAsync(d.trig.as<double>() + 3 * at, ht, run * 24, hipMemcpyHostToDevice, c.copy_stream));
    } else if (d.dev_trig) {
      sffk::launch_ring_trig(c.copy_stream, d.ring.as<uint64_t>() + at, d.trig.as<double>() + 3 * at, (int)run);
    }
    done += run;
  }
  d.produced += n;
  HIPCHK(hipEventRecord(d.ev_ring, c.copy_stream));
  d.ring_pending = true;
}

void Forest::dev_ring_top_up(uint64_t cursor, uint64_t ahead) {   // make [cursor, cursor + ahead) resident
  DevEngine& d = dev;
  if (d.produced >= cursor + ahead) return;
  const uint64_t need = cursor + ahead - d.produced;
  if (d.produced + need - cursor > d.ring_words) throw HipError{"forest: engine-word ring too small (internal error)"};
  dev_ring_append(nullptr, (size_t)need);
}

// host state -> device (first use, and after a wave finished on the host path)
void Forest::dev_upload_state() {
  Ctx& c = *ctx;
  DevEngine& d = dev;
  HIPCHK(hipSetDevice(c.device));
  const int wave = cfg.wave;
  const int words_per = cfg.dim == 2 ? 1 : 6;
  if (!d.inited) {
    HIPCHK(hipEventCreateWithFlags(&d.ev_ring, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&d.ev_wave, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&d.ev_wave2, hipEventDisableTiming));
    // (a slot's pick takes one word - four or so in the priority-frontier mode, whose plan also looks a few words ahead)
    d.max_wave_words = (uint64_t)wave * ((use_priority() ? 5 : 1) + (uint64_t)std::max(1, cfg.threshold_misses) * words_per) + 64;
    // (room for four waves' worth - and for everything the host engine may have generated ahead when the state moves
    // to the device in the middle of a run)
    d.ring_words = next_pow2(std::max<uint64_t>(4 * d.max_wave_words, (uint64_t)rng_ahead.size() + 2 * d.max_wave_words + 64));
    if (cfg.wave == 1) d.ring_words = std::max<uint64_t>(d.ring_words, 1 << 18);   // (k_seq_waves: thousands of waves per launch)
    d.ring.ensure((size_t)d.ring_words * 8);
    d.h_ring.ensure((size_t)d.ring_words * 8);
    if (cfg.libm_sampling) {
      d.trig.ensure((size_t)d.ring_words * 24);
      d.h_trig.ensure((size_t)d.ring_words * 24);
    } else if (cfg.wave == 1 && !d.dev_trig_off) {
      // waves of one slot: the one wavefront that is waited for looks its sample's cos / sin / acos up (k_ring_trig fills the table)
      d.trig.ensure((size_t)d.ring_words * 24);
      d.dev_trig = true;
    }
    d.ctrl.ensure(sizeof(sffk::DevCtrl));
    d.h_ctrl.ensure((size_t)SFFK_STATUS_RING * sizeof(sffk::DevCtrl));   // (>= 2: the copy path's two slots)
    d.zc_status = !(getenv("SFFGPU_NO_ZC_STATUS") && atoi(getenv("SFFGPU_NO_ZC_STATUS")) != 0);
    d.slot_node.ensure((size_t)wave * 4);
    d.slot_pos.ensure((size_t)wave * 4);
    d.act_slot.ensure((size_t)wave * 4);
    d.act_slot2.ensure((size_t)wave * 4);
    d.w_acc.ensure(((size_t)wave / 64 + 2) * 8);
    d.acc_pref.ensure(((size_t)wave / 64 + 2) * 4);
    d.w_ev.ensure(((size_t)wave / 64 + 2) * 8);
    d.ev_h.ensure(((size_t)wave + 64) * 8);
    d.ev_nb.ensure(((size_t)wave + 64) * 4);
    d.ev_raw.ensure(((size_t)wave + 64) * 4);
    // k_commit's sequence-stamped words start at zero once and are never cleared again
    d.ustate32.ensure(((size_t)wave + 64) * 4);
    d.wg_pub.ensure(((size_t)wave / 64 + 2) * SFFK_PUB_WORDS * 8);
    d.commit_seq.ensure(64);
    d.qclk_sh.ensure(64 * 16 * 8);
    HIPCHK(hipMemsetAsync(d.qclk_sh.p, 0, 64 * 16 * 8, c.stream));
    d.ord_hist.ensure((size_t)SFFK_ORD_BUCKETS * 4);
    d.ord_start.ensure((size_t)SFFK_ORD_BUCKETS * 4);
    d.ord_key.ensure((size_t)wave * 4);
    d.ord_rank.ensure((size_t)wave * 4);
    d.ord_pos.ensure((size_t)wave * 4);
    {
      const size_t n_sub = ((size_t)wave + 63) / 64;
      d.ord_lst.ensure(2 * n_sub * 64 * 4);
      d.ord_cnt.ensure(2 * n_sub * SFFK_ORD_CNT_STRIDE * 4);
      HIPCHK(hipMemsetAsync(d.ord_cnt.p, 0, 2 * n_sub * SFFK_ORD_CNT_STRIDE * 4, c.stream));
    }
    HIPCHK(hipMemsetAsync(d.ord_hist.p, 0, (size_t)SFFK_ORD_BUCKETS * 4, c.stream));
    if (getenv("SFFGPU_KC_TRACE")) {
      d.kc_trace.ensure(((size_t)wave / 64 + 2) * 64);
      HIPCHK(hipMemsetAsync(d.kc_trace.p, 0, ((size_t)wave / 64 + 2) * 64, c.stream));
    }
    HIPCHK(hipMemsetAsync(d.ustate32.p, 0, ((size_t)wave + 64) * 4, c.stream));
    HIPCHK(hipMemsetAsync(d.wg_pub.p, 0, ((size_t)wave / 64 + 2) * SFFK_PUB_WORDS * 8, c.stream));
    HIPCHK(hipMemsetAsync(d.commit_seq.p, 0, 64, c.stream));
    d.ulist.ensure((size_t)wave * 4);
    d.d_parent.ensure((size_t)wave * 4);
    d.d_parent2.ensure((size_t)wave * 4);
    d.d_force.ensure((size_t)wave);
    d.fault_pending.ensure(16);
    HIPCHK(hipMemsetAsync(d.fault_pending.p, 0, 16, c.stream));
    d.pair.ensure((size_t)num_roots * num_roots);
    d.inited = true;
  }
  // a store that leaves room for a wave of new nodes plus the round's temporaries behind them
  c.store_reserve((int)nodes.size() + 2 * wave + 64);
  dev_size_node_arrays();
  int nb = 0;
  for (auto& kv : borders) nb += (int)kv.second.size();
  {
    // (borders end up at 6-7 % of the nodes on the maps of BASELINE.json: room for a tenth of the node budget, so that a
    // job sized by its budget does not stop in the middle to grow these arrays - 4 ms at 850 k nodes on the bench job)
    int first_cap = std::max(std::max(1 << 16, 2 * (nb + wave)), cfg.node_budget > 0 ? cfg.node_budget / 10 + 2 * wave : 0);
    if (const char* e = getenv("SFFGPU_TEST_BORDER_CAP")) first_cap = std::max(nb + 1, atoi(e));   // tests: force growth
    dev_size_border_arrays(first_cap);
  }
  if (cfg.optimize) dev_star_setup();
  if (use_priority()) dev_prio_upload();
  // all nodes lie inside the limits: a bound for the fp32 filter slack that does not depend on the nodes to come
  for (int a = 0; a < 6; ++a) c.store_maxabs = std::max(c.store_maxabs, std::fabs(cfg.limits[a]));

  const int n = (int)nodes.size();
  {
    std::vector<int32_t> par(n);
    std::vector<double> dr(n), dc(n);
    std::vector<uint32_t> it(n);
    for (int i = 0; i < n; ++i) { par[i] = nodes[i].parent; dr[i] = nodes[i].d_root; dc[i] = nodes[i].d_closest; it[i] = nodes[i].iter; }
    HIPCHK(hipMemcpyAsync(d.parent.p, par.data(), (size_t)n * 4, hipMemcpyHostToDevice, c.stream));
    HIPCHK(hipMemcpyAsync(d.d_root.p, dr.data(), (size_t)n * 8, hipMemcpyHostToDevice, c.stream));
    HIPCHK(hipMemcpyAsync(d.d_closest.p, dc.data(), (size_t)n * 8, hipMemcpyHostToDevice, c.stream));
    HIPCHK(hipMemcpyAsync(d.iter.p, it.data(), (size_t)n * 4, hipMemcpyHostToDevice, c.stream));
    HIPCHK(hipMemcpyAsync(d.nflag.p, nflag.data(), (size_t)n, hipMemcpyHostToDevice, c.stream));
    if (!frontier.empty()) HIPCHK(hipMemcpyAsync(d.frontier.p, frontier.data(), frontier.size() * 4, hipMemcpyHostToDevice, c.stream));
    if (!closed.empty()) HIPCHK(hipMemcpyAsync(d.closed.p, closed.data(), closed.size() * 4, hipMemcpyHostToDevice, c.stream));
    HIPCHK(hipStreamSynchronize(c.stream));   // (the staging vectors go out of scope)
  }
  {   // borders: per-pair order is what matters (SpaceForest::borders is a matrix of lists)
    std::vector<int32_t> n1, n2, ta, tb;
    std::vector<double> ds;
    std::vector<uint8_t> pair((size_t)num_roots * num_roots, 0);
    for (auto& kv : borders)
      for (const Border& b : kv.second) {
        n1.push_back(b.n1); n2.push_back(b.n2); ta.push_back(kv.first.first); tb.push_back(kv.first.second); ds.push_back(b.dist);
        pair[(size_t)kv.first.first * num_roots + kv.first.second] = 1;
        pair[(size_t)kv.first.second * num_roots + kv.first.first] = 1;
      }
    const size_t m = n1.size();
    if (m) {
      HIPCHK(hipMemcpyAsync(d.b_n1.p, n1.data(), m * 4, hipMemcpyHostToDevice, c.stream));
      HIPCHK(hipMemcpyAsync(d.b_n2.p, n2.data(), m * 4, hipMemcpyHostToDevice, c.stream));
      HIPCHK(hipMemcpyAsync(d.b_ta.p, ta.data(), m * 4, hipMemcpyHostToDevice, c.stream));
      HIPCHK(hipMemcpyAsync(d.b_tb.p, tb.data(), m * 4, hipMemcpyHostToDevice, c.stream));
      HIPCHK(hipMemcpyAsync(d.b_dist.p, ds.data(), m * 8, hipMemcpyHostToDevice, c.stream));
    }
    HIPCHK(hipMemcpyAsync(d.pair.p, pair.data(), pair.size(), hipMemcpyHostToDevice, c.stream));
    // fresh table: every list entry is re-inserted
    HIPCHK(hipMemsetAsync(d.bt_key.p, 0, (size_t)d.bt_size * 8, c.stream));
    HIPCHK(hipMemsetAsync(d.bt_val.p, 0xFF, (size_t)d.bt_size * 8, c.stream));
    d.table_dirty = false;
    sffk::launch_border_rehash(c.stream, dev_view(), (int)m);
    HIPCHK(hipStreamSynchronize(c.stream));
    d.host_borders = (int)m;
  }
  // engine words: what the host generated ahead moves into the ring, the generator continues behind it
  {
    const uint64_t cursor = rng.draws;
    d.produced = cursor;
    const size_t left = rng.qn - rng.qh;
    std::vector<uint64_t> ahead(rng.q ? rng.q + rng.qh : nullptr, rng.q ? rng.q + rng.qn : nullptr);
    rng.q = nullptr;
    rng.qh = rng.qn = 0;
    if (left) dev_ring_append(ahead.data(), std::min<size_t>(left, (size_t)d.ring_words));
    if (left > d.ring_words) throw HipError{"forest: look-ahead queue larger than the engine-word ring"};
    rng.draws = d.produced;
    sffk::DevCtrl k{};
    k.status_seq = (int32_t)d.status_next;   // (the ring's numbering goes on)
    k.n_nodes = n;
    k.iter = iter;
    k.round = round;
    k.in_wave = in_wave ? 1 : 0;
    k.frontier_n = (int)frontier.size();
    k.closed_n = (int)closed.size();
    k.solved = solved ? 1 : 0;
    k.empty_frontier = empty_frontier ? 1 : 0;
    k.terminated = (!in_wave && terminated()) ? 1 : 0;
    k.halt = k.terminated;
    k.n_borders = d.host_borders;
    k.cursor = cursor;
    k.collide_calls = st.collide_calls;
    k.path_free_calls = st.path_free_calls;
    k.nn_queries = st.nn_queries;
    k.poses_executed = st.poses_executed;
    k.segments_executed = st.segments_executed;
    k.samples_executed = st.samples_executed;
    k.waves = st.waves;
    k.rounds = st.sweeps;
    k.round_nodes = st.sweep_nodes;
    k.round_queries = st.sweep_queries;
    k.star_rounds = st.star_rounds;
    k.star_passes = st.star_passes;
    k.star_members = st.star_members;
    k.star_rewires = st.star_rewires;
    k.spec_steps = st.spec_steps;
    k.spec_evaluated = st.spec_evaluated;
    k.spec_committed = st.spec_committed;
    k.epoch = 1;
    k.prio_n0 = n;               // (the host engine has pushed the nodes it created itself)
    k.prio_all_empty = (use_priority() && all_frontiers_empty()) ? 1 : 0;
    k.prio_gen = -2;
    if (in_wave && use_priority() && !slots.empty() && slots[0].tree >= 0) {
      k.prio_wave = 1;
      std::vector<int32_t> stt(slots.size()), sh(slots.size());
      for (size_t s = 0; s < slots.size(); ++s) { stt[s] = slots[s].tree; sh[s] = slots[s].heap; }
      HIPCHK(hipMemcpy(d.slot_tree.p, stt.data(), stt.size() * 4, hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(d.slot_heap.p, sh.data(), sh.size() * 4, hipMemcpyHostToDevice));
    }
    if (in_wave) {
      k.n_slots = (int)slots.size();
      k.use_closed = (!slots.empty() && slots[0].from_closed) ? 1 : 0;
      std::vector<int32_t> sn(slots.size()), sp(slots.size(), 0), act;
      std::vector<int32_t> where(nodes.size(), 0);   // frontier position of every node (k_wave_end marks positions)
      for (size_t r = 0; r < frontier.size(); ++r) where[frontier[r]] = (int32_t)r;
      for (size_t s = 0; s < slots.size(); ++s) {
        sn[s] = slots[s].node;
        sp[s] = where[slots[s].node];
        if (slots[s].failing) act.push_back((int32_t)s);   // the active list: failing slots in slot order
      }
      k.act_cnt = (int)act.size();
      k.act_sel = 0;
      HIPCHK(hipMemcpy(d.slot_pos.p, sp.data(), sp.size() * 4, hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(d.slot_node.p, sn.data(), sn.size() * 4, hipMemcpyHostToDevice));
      if (!act.empty()) HIPCHK(hipMemcpy(d.act_slot.p, act.data(), act.size() * 4, hipMemcpyHostToDevice));
    }
    HIPCHK(hipMemcpy(d.ctrl.p, &k, sizeof k, hipMemcpyHostToDevice));
    d.last = k;
  }
  d.host_nodes = n;
  d.host_stale = false;
  d.active = true;
}

// device state -> host mirror
void Forest::sync_host() {
  DevEngine& d = dev;
  if (!d.active || !d.host_stale) return;
  Ctx& c = *ctx;
  HIPCHK(hipSetDevice(c.device));
  HIPCHK(hipStreamSynchronize(c.stream));
  sffk::DevCtrl k;
  HIPCHK(hipMemcpy(&k, d.ctrl.p, sizeof k, hipMemcpyDeviceToHost));
  if (cfg.optimize && d.star_inited) {
    // the star stage's sub-counters are folded into the control block at every wave end; what a wave that stopped in
    // the middle (a fault) left in them is folded here
    unsigned long long acc[64 * SFFK_STAR_ACC], sum[SFFK_STAR_ACC] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIPCHK(hipMemcpy(acc, d.s_acc.p, sizeof acc, hipMemcpyDeviceToHost));
    bool any = false;
    for (int j = 0; j < 64 * SFFK_STAR_ACC; ++j) { sum[j % SFFK_STAR_ACC] += acc[j]; any |= acc[j] != 0; }
    if (any) {
      k.collide_calls += sum[0]; k.path_free_calls += sum[1];
      k.star_rounds += sum[2]; k.star_passes += sum[3]; k.star_members += sum[4]; k.star_rewires += sum[5];
      HIPCHK(hipMemset(d.s_acc.p, 0, sizeof acc));
      HIPCHK(hipMemcpy(d.ctrl.p, &k, sizeof k, hipMemcpyHostToDevice));
    }
  }
  d.last = k;
  const int n = k.n_nodes, n0 = d.host_nodes;
  if (n > n0) {
    const int m = n - n0;
    std::vector<double> pos((size_t)m * 6), dr(m), dc(m);
    std::vector<int32_t> par(m), tr(m);
    std::vector<uint32_t> it(m);
    HIPCHK(hipMemcpy(pos.data(), c.spos.as<double>() + 6 * (size_t)n0, (size_t)m * 48, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(tr.data(), c.stree.as<int32_t>() + n0, (size_t)m * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(par.data(), d.parent.as<int32_t>() + n0, (size_t)m * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(dr.data(), d.d_root.as<double>() + n0, (size_t)m * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(dc.data(), d.d_closest.as<double>() + n0, (size_t)m * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(it.data(), d.iter.as<uint32_t>() + n0, (size_t)m * 4, hipMemcpyDeviceToHost));
    for (int j = 0; j < m; ++j) add_node(&pos[6 * (size_t)j], tr[j], par[j], dc[j], dr[j], it[j]);
  }
  if (cfg.optimize && cfg.record_parents && d.s_hist.p) {   // the device's new history entries join the host's list
    int32_t ctl[4];
    HIPCHK(hipMemcpy(ctl, d.s_hist.p, 16, hipMemcpyDeviceToHost));
    if (ctl[1]) hist_overflow = true;
    const int m = std::min(ctl[0], d.hist_cap);
    if (m > 0) {
      std::vector<int32_t> raw((size_t)3 * m);
      HIPCHK(hipMemcpy(raw.data(), d.s_hist.as<int32_t>() + 4, raw.size() * 4, hipMemcpyDeviceToHost));
      for (int j = 0; j < m; ++j) hist.push_back({raw[3 * (size_t)j], raw[3 * (size_t)j + 1], (uint32_t)raw[3 * (size_t)j + 2]});
      HIPCHK(hipMemset(d.s_hist.p, 0, 16));
    }
  }
  if (cfg.optimize && n0 > 0) {   // SFF*: rewires change parent / costs of nodes the mirror already holds
    std::vector<double> dr(n0), dc(n0);
    std::vector<int32_t> par(n0);
    HIPCHK(hipMemcpy(par.data(), d.parent.p, (size_t)n0 * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(dr.data(), d.d_root.p, (size_t)n0 * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(dc.data(), d.d_closest.p, (size_t)n0 * 8, hipMemcpyDeviceToHost));
    for (int j = 0; j < n0; ++j) { nodes[j].parent = par[j]; nodes[j].d_root = dr[j]; nodes[j].d_closest = dc[j]; }
  }
  if (n > 0) HIPCHK(hipMemcpy(nflag.data(), d.nflag.p, (size_t)n, hipMemcpyDeviceToHost));
  frontier.resize((size_t)k.frontier_n);
  closed.resize((size_t)k.closed_n);
  if (k.frontier_n) HIPCHK(hipMemcpy(frontier.data(), k.front_sel ? d.frontier2.p : d.frontier.p, (size_t)k.frontier_n * 4, hipMemcpyDeviceToHost));
  if (k.closed_n) HIPCHK(hipMemcpy(closed.data(), d.closed.p, (size_t)k.closed_n * 4, hipMemcpyDeviceToHost));
  if (k.n_borders > d.host_borders) {
    const int b0 = d.host_borders, m = k.n_borders - b0;
    std::vector<int32_t> n1(m), n2(m), ta(m), tb(m);
    std::vector<double> ds(m);
    HIPCHK(hipMemcpy(n1.data(), d.b_n1.as<int32_t>() + b0, (size_t)m * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(n2.data(), d.b_n2.as<int32_t>() + b0, (size_t)m * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ta.data(), d.b_ta.as<int32_t>() + b0, (size_t)m * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(tb.data(), d.b_tb.as<int32_t>() + b0, (size_t)m * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ds.data(), d.b_dist.as<double>() + b0, (size_t)m * 8, hipMemcpyDeviceToHost));
    for (int j = 0; j < m; ++j) {
      border(ta[j], tb[j]).push_back({n1[j], n2[j], ds[j]});
      border_keys.insert(((uint64_t)(uint32_t)n1[j] << 32) | ((uint64_t)(uint32_t)n2[j] + 1));
    }
    d.host_borders = k.n_borders;
  }
  iter = k.iter;
  solved = k.solved != 0;
  empty_frontier = k.empty_frontier != 0;
  in_wave = k.in_wave != 0;
  round = k.round;
  st.collide_calls = k.collide_calls;
  st.path_free_calls = k.path_free_calls;
  st.nn_queries = k.nn_queries;
  st.poses_executed = k.poses_executed;
  st.segments_executed = k.segments_executed;
  st.samples_executed = k.samples_executed;
  st.waves = k.waves;
  st.sweeps = k.rounds;
  st.sweep_nodes = k.round_nodes;
  st.sweep_queries = k.round_queries;
  st.star_rounds = k.star_rounds;
  st.star_passes = k.star_passes;
  st.star_members = k.star_members;
  st.star_rewires = k.star_rewires;
  st.spec_steps = k.spec_steps;
  st.spec_evaluated = k.spec_evaluated;
  st.spec_committed = k.spec_committed;
  if (use_priority()) {
    dev_prio_download();
    // the device pushes a wave's new nodes at the wave's END (k_prio_end); a wave that stopped in the middle has not
    // pushed them yet - the mirror does it here, in creation order, like the host engine does when it accepts them
    if (in_wave)
      for (int id = std::max(k.prio_n0, 0); id < n; ++id)
        for (PHeap& hp : heaps[nodes[id].tree]) hp.push(id);
  }
  if (in_wave) {
    std::vector<int32_t> sn(k.n_slots), act((size_t)k.act_cnt);
    std::vector<uint8_t> sf(k.n_slots, 0);
    HIPCHK(hipMemcpy(sn.data(), d.slot_node.p, (size_t)k.n_slots * 4, hipMemcpyDeviceToHost));
    if (k.act_cnt) HIPCHK(hipMemcpy(act.data(), k.act_sel ? d.act_slot2.p : d.act_slot.p, (size_t)k.act_cnt * 4, hipMemcpyDeviceToHost));
    for (int32_t s : act) sf[s] = 1;     // still failing = still on the active list
    slots.resize((size_t)k.n_slots);
    for (int s = 0; s < k.n_slots; ++s) { slots[s].node = sn[s]; slots[s].failing = sf[s] != 0; slots[s].from_closed = k.use_closed != 0; slots[s].tree = -1; slots[s].heap = -1; }
    if (use_priority() && k.prio_wave) {   // the slots hold heap nodes: where each came from
      std::vector<int32_t> stt(k.n_slots), sh(k.n_slots);
      HIPCHK(hipMemcpy(stt.data(), d.slot_tree.p, (size_t)k.n_slots * 4, hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(sh.data(), d.slot_heap.p, (size_t)k.n_slots * 4, hipMemcpyDeviceToHost));
      for (int s = 0; s < k.n_slots; ++s) { slots[s].tree = stt[s]; slots[s].heap = sh[s]; slots[s].from_closed = false; }
    }
  }
  c.store_n = n;
  c.grid_inserted = n;
  d.host_nodes = n;
  d.host_stale = false;
}

// leave device mode: the host path owns the state again (its rng continues at the device's cursor)
void Forest::dev_to_host() {
  DevEngine& d = dev;
  exchange_timer_drop();
  if (!d.active) return;
  d.host_stale = true;
  sync_host();
  Ctx& c = *ctx;
  HIPCHK(hipStreamSynchronize(c.copy_stream));
  const uint64_t cursor = d.last.cursor;
  const size_t left = (size_t)(d.produced - cursor);
  const size_t keep = std::max(rng_ahead.size(), left + 16);
  std::vector<uint64_t> buf(keep, 0);
  const uint64_t* hr = d.h_ring.as<uint64_t>();
  for (size_t j = 0; j < left; ++j) buf[j] = hr[(cursor + j) & (d.ring_words - 1)];
  rng_ahead.swap(buf);
  rng.q = rng_ahead.data();
  rng.qh = 0;
  rng.qn = left;
  rng.draws = cursor;
  d.active = false;
}

// buffers of one round (same layout as the host path's output block, sized for a full wave)
struct DevRoundBufs {
  int n, CAP, NBCAP, STRIDE, list_cap;
  double *d_pos, *d_pd;
  int32_t *d_rec, *d_rctrl, *seg_ns, *first_hit, *seg_ovf;
  uint8_t *d_lim, *d_pose, *code;
  unsigned long long* bulk;
};
static DevRoundBufs dev_round_bufs(Forest& F) {
  Ctx& c = *F.ctx;
  DevRoundBufs B{};
  const int n = F.cfg.wave;   // launch bound; the kernels read the real count from DevCtrl
  B.n = n; B.CAP = F.hit_cap; B.NBCAP = F.nb_cap; B.STRIDE = 1 + B.NBCAP;
  const size_t rec_ints = (size_t)n * (2 + 2 * B.NBCAP);
  const size_t o_pos = 0, o_pd = o_pos + (size_t)n * 48, o_lim = o_pd + (size_t)n * 8,
               o_rec = o_lim + ((size_t)n + 15) / 16 * 16, o_ns = o_rec + rec_ints * 4,
               o_fh = o_ns + (size_t)n * B.STRIDE * 4, o_ctrl = o_fh + (size_t)n * B.STRIDE * 4, o_pose = o_ctrl + 128,
               o_code = o_pose + (size_t)n, o_bytes = (o_code + (size_t)n + 15) / 16 * 16, o_ovf = o_bytes;
  c.r_out.ensure(o_ovf + (size_t)n * B.STRIDE * 4);
  char* dout = c.r_out.as<char>();
  B.d_pos = reinterpret_cast<double*>(dout + o_pos);
  B.d_pd = reinterpret_cast<double*>(dout + o_pd);
  if (F.dev.round_parity) {   // (the odd rounds of a wave: the append of round r reads what the sampling of round r + 1 writes)
    c.r_out2.ensure((size_t)n * 56);
    B.d_pos = c.r_out2.as<double>();
    B.d_pd = B.d_pos + (size_t)n * 6;
  }
  B.d_rec = reinterpret_cast<int32_t*>(dout + o_rec);
  B.d_rctrl = reinterpret_cast<int32_t*>(dout + o_ctrl);
  B.d_lim = reinterpret_cast<uint8_t*>(dout + o_lim);
  B.d_pose = reinterpret_cast<uint8_t*>(dout + o_pose);
  B.seg_ns = reinterpret_cast<int32_t*>(dout + o_ns);
  B.first_hit = reinterpret_cast<int32_t*>(dout + o_fh);
  B.seg_ovf = reinterpret_cast<int32_t*>(dout + o_ovf);
  B.code = reinterpret_cast<uint8_t*>(dout + o_code);
  B.bulk = reinterpret_cast<unsigned long long*>(dout + o_ctrl + 16);
  c.r_q.ensure((size_t)n * sizeof(sffk::SweepQuery));
  c.r_cnt.ensure((size_t)n * 4);
  c.r_sega.ensure((size_t)n * B.STRIDE * 48);
  c.r_segb.ensure((size_t)n * B.STRIDE * 48);
  c.r_center.ensure((size_t)n * 48);
  c.r_qrec.ensure((size_t)n * sizeof(sffk::QRec));
  B.list_cap = 4 * n * B.STRIDE + 65536;
  c.r_items.ensure((size_t)B.list_cap * SFFK_ITEM_BYTES);
  c.r_sub.ensure((size_t)SFFK_SUBLISTS * SFFK_SUB_STRIDE * 4);
  return B;
}
static sffk::ResolveArgs dev_resolve_args(Forest& F, const DevRoundBufs& B) {
  Ctx& c = *F.ctx;
  DevEngine& d = F.dev;
  sffk::ResolveArgs ra{};
  ra.f = F.dev_view();
  ra.st = sffk::NodeStoreMut{c.sx.as<float>(), c.sy.as<float>(), c.sz.as<float>(), c.syaw.as<float>(),
                             c.spitch.as<float>(), c.sroll.as<float>(), c.stree.as<int32_t>(), c.spos.as<double>()};
  ra.g = c.gridv;
  ra.nbcap = B.NBCAP;
  ra.stride = B.STRIDE;
  ra.rank = F.cfg.rank;
  ra.world = F.cfg.world;
  ra.newpos = B.d_pos;
  ra.pdist = B.d_pd;
  ra.parent = (d.round_parity ? d.d_parent2 : d.d_parent).as<int32_t>();
  ra.code = B.code;
  ra.in_lim = B.d_lim;
  ra.rec_flags = B.d_rec;
  ra.pose_hit = B.d_pose;
  ra.rec_nnb = B.d_rec + B.n;
  ra.rec_nb = B.d_rec + 2 * (size_t)B.n;
  ra.rec_meta = B.d_rec + 2 * (size_t)B.n + (size_t)B.n * B.NBCAP;
  ra.seg_ns = B.seg_ns;
  ra.first_hit = B.first_hit;
  ra.bulk = B.bulk;
  ra.round_ctrl = B.d_rctrl;
  ra.fault_pending = d.fault_pending.as<int32_t>();
  ra.star = F.cfg.optimize ? 1 : 0;
  if (ra.star) ra.S = F.star_view();
  return ra;
}

size_t Forest::dev_exchange_bytes() const {   // what one rank contributes to the all-gather of a round
  const int per_rank = (cfg.wave + cfg.world - 1) / cfg.world;
  return (size_t)per_rank * sffk::record_words(nb_cap) * 4;
}

void Forest::dev_enqueue_begin() {
  Ctx& c = *ctx;
  DevEngine& d = dev;
  if (d.ring_pending) {   // the words this wave may read have to be resident
    HIPCHK(hipStreamWaitEvent(c.stream, d.ev_ring, 0));
    d.ring_pending = false;
  }
  sffk::launch_wave_begin(c.stream, dev_view());
}

// sample -> neighbour query + classification -> collision, for the samples this rank owns; with send_dev the owned
// samples' answer records are packed for the all-gather
// k_sample_steer's arguments for the round whose buffers are B (DevEngine::round_parity)
static sffk::SampleLaunch dev_sample_launch(Forest& F, const DevRoundBufs& B) {
  Ctx& c = *F.ctx;
  DevEngine& d = F.dev;
  const sffk::DevForestView V = F.dev_view();
  sffk::NodeStoreMut stm{c.sx.as<float>(), c.sy.as<float>(), c.sz.as<float>(), c.syaw.as<float>(),
                         c.spitch.as<float>(), c.sroll.as<float>(), c.stree.as<int32_t>(), c.spos.as<double>()};
  sffk::SampleLaunch P{};
  memcpy(P.prm.limits, F.cfg.limits, sizeof P.prm.limits);
  P.prm.dist_tree = F.cfg.dist_tree;
  P.prm.sweep_abs_eps = c.sweep_eps();
  P.prm.rank = F.cfg.rank;
  P.prm.world = F.cfg.world;
  P.tmp.st = stm;
  P.tmp.cnt = c.r_cnt.as<int32_t>();
  P.tmp.tg = c.tgridv;
  P.tmp.ctrl = B.d_rctrl;
  P.tmp.sub = c.r_sub.as<int32_t>();
  P.tmp.n_perm = d.temp_base;
  P.tmp.base = d.temp_base;
  P.tmp.center_out = c.r_center.as<double>();
  P.tmp.qrec = c.r_qrec.as<sffk::QRec>();
  memcpy(P.tmp.clear_org, c.envv.clear_org, sizeof P.tmp.clear_org);
  P.tmp.clear_inv = c.envv.clear_inv;
  P.dv.ctrl = V.ctrl;
  P.dv.act_slot = V.act_slot;
  P.dv.act_slot2 = V.act_slot2;
  P.dv.slot_node = V.slot_node;
  P.dv.nflag = V.nflag;
  P.dv.ring = V.ring;
  P.dv.ring_mask = V.ring_mask;
  P.dv.words_per = V.words_per;
  P.dv.trig = F.cfg.libm_sampling ? d.trig.as<double>() : nullptr;
  P.dv.parent_out = (d.round_parity ? d.d_parent2 : d.d_parent).as<int32_t>();
  P.dv.force_out = d.d_force.as<uint8_t>();
  P.dv.qclk = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(d.ctrl.p) + offsetof(sffk::DevCtrl, q_t0));
  P.dv.qclk_sh = d.qclk_sh.as<unsigned long long>();
  P.dv.ord = V.ord;
  P.node_pos = c.spos.as<double>();
  P.n = B.n;
  P.dist = F.cfg.sampling_dist;
  P.dim = F.cfg.dim;
  P.out6 = B.d_pos;
  P.in_lim = B.d_lim;
  P.parent_dist = B.d_pd;
  P.queries = c.r_q.as<sffk::SweepQuery>();
  P.q_max_base = d.temp_base;
  return P;
}

void Forest::dev_enqueue_round_eval(void* send_dev, bool sample) {
  Ctx& c = *ctx;
  DevEngine& d = dev;
  const sffk::DevForestView V = dev_view();
  const DevRoundBufs B = dev_round_bufs(*this);
  const int n = B.n;
  const int32_t* dev_n = reinterpret_cast<const int32_t*>(d.ctrl.p);   // {n_act, halt}
  c.timing_on = d.force_timing >= 0 ? d.force_timing != 0
                                   : (c.timer_stride <= 1 || d.rounds_enqueued % (uint64_t)c.timer_stride == 0);
  d.round_timing = c.timing_on;   // (the commit of this round is timed like its evaluation)
  c.round_scope = true;
  ++d.rounds_enqueued;
  if (sample) {
    const sffk::SampleLaunch P = dev_sample_launch(*this, B);
    c.time_begin(T_SAMPLE);
    sffk::launch_sample_steer(c.stream, P);
    c.time_end();
  }
  int32_t* const round_parent = (d.round_parity ? d.d_parent2 : d.d_parent).as<int32_t>();
  unsigned long long* const qclk = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(d.ctrl.p) + offsetof(sffk::DevCtrl, q_t0));
  sffk::ClassifyArgs ca{};
  ca.n = n; ca.N0 = d.temp_base; ca.cap = B.CAP; ca.nbcap = B.NBCAP; ca.rank = cfg.rank; ca.world = cfg.world;
  ca.goal_id = cfg.has_goal ? goal_node : -1;
  ca.wide = query_wide ? 1 : 0;
  ca.lazy_nb = (cfg.world <= 1 && !cfg.optimize) ? 1 : 0;   // (SFF*'s choose-parent reads whole records; sharded rounds exchange them)
  ca.qrec = c.r_qrec.as<sffk::QRec>();
  ca.dist_tree = cfg.dist_tree;
  ca.newpos = B.d_pos;
  ca.in_lim = B.d_lim;
  ca.pdist = B.d_pd;
  ca.parent = round_parent;
  ca.center = c.r_center.as<double>();
  ca.force = d.d_force.as<uint8_t>();
  ca.cnt = c.r_cnt.as<int32_t>();
  ca.hit_idx = nullptr;
  ca.hit_dist = nullptr;
  ca.tree = c.stree.as<int32_t>();
  ca.pos = c.spos.as<double>();
  ca.rec_flags = B.d_rec;
  ca.rec_nnb = ca.rec_flags + n;
  ca.rec_nb = ca.rec_nnb + n;
  ca.rec_meta = ca.rec_nb + (size_t)n * B.NBCAP;
  ca.seg_a = c.r_sega.as<double>();
  ca.seg_b = c.r_segb.as<double>();
  ca.seg_ns = B.seg_ns;
  ca.first_hit = B.first_hit;
  ca.seg_ovf = B.seg_ovf;
  ca.ctrl = B.d_rctrl;
  ca.dev_n = dev_n;
  ca.qclk = qclk;
  ca.qclk_sh = d.qclk_sh.as<unsigned long long>();
  if (V.ord.hist) {
    ca.ord_valid = reinterpret_cast<const int32_t*>(reinterpret_cast<char*>(d.ctrl.p) + offsetof(sffk::DevCtrl, ord_valid));
    ca.ord_nslots = reinterpret_cast<const int32_t*>(reinterpret_cast<char*>(d.ctrl.p) + offsetof(sffk::DevCtrl, n_slots));
    ca.ord_sel = reinterpret_cast<const int32_t*>(reinterpret_cast<char*>(d.ctrl.p) + offsetof(sffk::DevCtrl, act_sel));
    for (int q = 0; q < 2; ++q) { ca.ord_lst[q] = V.ord.lst[q]; ca.ord_cnt[q] = V.ord.cnt[q]; }
  }
  c.time_begin(T_SWEEP);
  ca.items = c.r_items.p;
  ca.items_cap = B.list_cap;
  ca.sub = c.r_sub.as<int32_t>();
  ca.pose_hit = B.d_pose;
  const bool blocked = sffk::launch_query_classify(c.stream, c.gridv, &c.tgridv, c.store_view(), c.r_q.as<sffk::SweepQuery>(), ca, &c.envv);
  c.time_end();
  c.time_begin(T_COLLIDE);
  sffk::TempGridRef tref{c.tgridv, c.sx.as<float>() + d.temp_base, c.sy.as<float>() + d.temp_base,
                         c.sz.as<float>() + d.temp_base, n};
  // (SFF*: the star stage of the commit still reads the round's own grid - k_star_apply empties it)
  sffk::TempGridRef tref_keep = tref;
  tref_keep.tg = sffk::GridView{};
  tref_keep.n = 0;
  sffk::launch_collide_items(c.stream, c.envv, c.robv, B.d_pos, n, ca.rec_flags, B.d_pose, ca.seg_a, ca.seg_b, ca.seg_ns,
                             B.STRIDE, ca.ctrl, c.r_items.p, ca.items_cap, ca.sub, ca.first_hit, ca.seg_ovf,
                             cfg.optimize ? &tref_keep : &tref, dev_n, blocked ? &ca : nullptr);
  c.time_end();
  if (send_dev) {
    c.time_begin(T_EXCHANGE);     // (closed by the commit: pack, the caller's / the library's all-gather, unpack)
    sffk::launch_pack_records(c.stream, dev_resolve_args(*this, B), cfg.rank, cfg.world, n, static_cast<int32_t*>(send_dev));
    exchange_timer_drop();
    exchange_open = c.timed_now;
    if (exchange_open) { exchange_t = c.pending.back(); c.pending.pop_back(); }
  }
  c.timing_on = true;
  c.round_scope = false;
}

// the in-order commit, replicated on every rank; with recv_dev the other ranks' answer records are unpacked first
void Forest::dev_enqueue_round_commit(const void* recv_dev, bool sample_next) {
  Ctx& c = *ctx;
  const DevRoundBufs B = dev_round_bufs(*this);
  const sffk::ResolveArgs ra = dev_resolve_args(*this, B);
  if (recv_dev) {
    sffk::launch_unpack_records(c.stream, ra, cfg.rank, cfg.world, B.n, static_cast<const int32_t*>(recv_dev));
    if (exchange_open) {
      HIPCHK(hipEventRecord(exchange_t.b, c.stream));
      c.pending.push_back(exchange_t);
      exchange_open = false;
    }
  }
  // (the commit's last kernel also draws the next round's samples - into the other set of sample arrays)
  sffk::SampleLaunch next{};
  if (sample_next) {
    dev.round_parity ^= 1;
    next = dev_sample_launch(*this, dev_round_bufs(*this));
    dev.round_parity ^= 1;
  }
  c.round_scope = true;
  c.timing_on = dev.round_timing;
  c.time_begin(T_COMMIT);
  if (cfg.optimize) {
    // SFF*: k nearest + member edges + the rewire fixed point for the accepted samples, between k_commit and k_append
    sffk::StarLaunch sl{};
    sl.g = c.gridv;
    sl.tg = c.tgridv;
    sl.st = c.store_view();
    sl.env = c.envv;
    sl.rob = c.robv;
    sl.cell_edge = c.grid_cell;
    sl.slack = 8 * c.sweep_eps();
    sl.cube_reach = 2.0 * cfg.sampling_dist;
    sl.passes = star_pass_limit;
    sl.tail = star_tail ? 1 : 0;
    sl.tail_wgs = star_tail_wgs;
    sl.tail_stall = star_tail_stall;
    sffk::launch_commit(c.stream, ra, B.n, &sl, sample_next ? &next : nullptr);
  } else {
    sffk::launch_commit(c.stream, ra, B.n, nullptr, sample_next ? &next : nullptr);
  }
  c.time_end();
  c.timing_on = true;
  c.round_scope = false;
}

void Forest::dev_enqueue_end(int slot) {
  Ctx& c = *ctx;
  DevEngine& d = dev;
  if (use_priority()) sffk::launch_prio_end(c.stream, dev_view(), c.store_view());
  sffk::launch_wave_end(c.stream, dev_view(), c.gridv.ovf_cnt, c.tgridv.ovf_cnt,
                        cfg.optimize ? d.s_acc.as<unsigned long long>() : nullptr);
  d.status_copied[slot] = !d.zc_status;
  if (d.zc_status) d.status_of[slot] = d.status_next++;   // (k_wave_end_wide published the block itself)
  else HIPCHK(hipMemcpyAsync(d.h_ctrl.as<sffk::DevCtrl>() + slot, d.ctrl.p, sizeof(sffk::DevCtrl), hipMemcpyDeviceToHost, c.stream));
  HIPCHK(hipEventRecord(slot ? d.ev_wave2 : d.ev_wave, c.stream));
}

// FNV-1a over everything a wave's launches bake in: a captured graph is only valid while this stays the same
uint64_t Forest::dev_launch_signature() {
  Ctx& c = *ctx;
  uint64_t x = 1469598103934665603ULL;
  auto mix = [&](const void* p, size_t n) {
    const unsigned char* q = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < n; ++i) { x ^= q[i]; x *= 1099511628211ULL; }
  };
  for (int parity = 0; parity < 2; ++parity) {
    dev.round_parity = parity;
    const DevRoundBufs B = dev_round_bufs(*this);
    const sffk::ResolveArgs ra = dev_resolve_args(*this, B);
    mix(&B, sizeof B);
    mix(&ra, sizeof ra);
  }
  dev.round_parity = 0;
  mix(&c.gridv, sizeof c.gridv);
  mix(&c.tgridv, sizeof c.tgridv);
  mix(&c.envv, sizeof c.envv);
  mix(&c.robv, sizeof c.robv);
  const void* ptrs[] = {c.r_q.p, c.r_cnt.p, c.r_sega.p, c.r_segb.p, c.r_items.p, c.r_sub.p, c.r_out.p, c.r_center.p, c.r_qrec.p, dev.d_parent.p, dev.d_force.p,
                        dev.ctrl.p, dev.ring.p, dev.trig.p, c.sx.p, c.spos.p, c.stree.p};
  mix(ptrs, sizeof ptrs);
  const double scal[] = {c.sweep_eps(), c.grid_cell, cfg.sampling_dist, cfg.dist_tree};
  mix(scal, sizeof scal);
  const int ints[] = {cfg.threshold_misses, star_pass_limit, cfg.wave, c.store_cap, dev.temp_base, hit_cap, nb_cap, query_wide ? 1 : 0};
  mix(ints, sizeof ints);
  return x ? x : 1;
}

void Forest::dev_enqueue_wave_kernels(bool sharded, size_t words) {
  sffk::launch_wave_begin(ctx->stream, dev_view());
  // round r's append and round r + 1's sampling are one launch (k_append_sample): the rounds alternate between two sets
  // of sample arrays, only the first round of the wave samples on its own
  const int R = std::max(1, cfg.threshold_misses);
  static const bool fuse = !(getenv("SFFGPU_NO_FUSED_SAMPLE") && atoi(getenv("SFFGPU_NO_FUSED_SAMPLE")) != 0);
  for (int r = 0; r < R; ++r) {
    dev.round_parity = fuse ? (r & 1) : 0;
    dev_enqueue_round_eval(sharded ? x_send.p : nullptr, !fuse || r == 0);
    if (sharded) ctx->rccl_all_gather_i32(x_send.p, x_recv.p, words);
    dev_enqueue_round_commit(sharded ? x_recv.p : nullptr, fuse && r + 1 < R);
  }
  dev.round_parity = 0;
  if (use_priority()) sffk::launch_prio_end(ctx->stream, dev_view(), ctx->store_view());
  sffk::launch_wave_end(ctx->stream, dev_view(), ctx->gridv.ovf_cnt, ctx->tgridv.ovf_cnt,
                        cfg.optimize ? dev.s_acc.as<unsigned long long>() : nullptr);
}

// one whole wave: begin, ThresholdMisses rounds (the device skips what it does not need), end + status copy
void Forest::dev_enqueue_wave(int slot) {
  Ctx& c = *ctx;
  DevEngine& d = dev;
  if (d.ring_pending) {   // the words this wave may read have to be resident
    HIPCHK(hipStreamWaitEvent(c.stream, d.ev_ring, 0));
    d.ring_pending = false;
  }
  // (SFFGPU_TEST_EXCHANGE_SELF: a one-rank forest packs, all-gathers and unpacks too - the collective on one GPU)
  const bool self_exchange = test_exchange_self;
  const bool sharded = cfg.world > 1 || (self_exchange && (ctx->rccl_comm != nullptr || ctx->xchg_fn != nullptr));
  size_t words = 0;
  if (sharded) {   // this rank's answer records of a round -> all ranks' (ncclAllGather between device buffers)
    words = dev_exchange_bytes() / 4;
    x_send.ensure(words * 4);
    x_recv.ensure(words * 4 * (size_t)cfg.world);
  }
  // every timer_stride-th wave is launched kernel by kernel with HIP events around the timed kernels (the figures are
  // scaled to all waves); the others replay the captured graph
  const bool timed_wave = c.timer_stride <= 1 || d.waves_enqueued % (uint64_t)c.timer_stride == 0;
  ++d.waves_enqueued;
  const bool use_graph = d.graph_enabled && !sharded && !timed_wave;
  if (use_graph) {
    const uint64_t sig = dev_launch_signature();   // (also makes sure every buffer of a round exists: no allocation inside the capture)
    if (!d.wave_graph || sig != d.wave_graph_sig) {
      if (d.wave_graph) { (void)hipGraphExecDestroy(d.wave_graph); d.wave_graph = nullptr; }
      uint64_t before[T_KINDS];
      for (int k = 0; k < T_KINDS; ++k) before[k] = c.round_calls[k];
      d.force_timing = 0;
      HIPCHK(hipStreamBeginCapture(c.stream, hipStreamCaptureModeRelaxed));
      hipGraph_t g = nullptr;
      try {
        dev_enqueue_wave_kernels(false, 0);
      } catch (...) {
        (void)hipStreamEndCapture(c.stream, &g);
        if (g) (void)hipGraphDestroy(g);
        d.force_timing = -1;
        throw;
      }
      d.force_timing = -1;
      HIPCHK(hipStreamEndCapture(c.stream, &g));
      HIPCHK(hipGraphInstantiate(&d.wave_graph, g, nullptr, nullptr, 0));
      (void)hipGraphDestroy(g);
      for (int k = 0; k < T_KINDS; ++k) {   // (the capture itself executed nothing: its counts are taken back)
        d.graph_calls[k] = c.round_calls[k] - before[k];
        c.round_calls[k] = before[k];
        c.kernel_calls[k] -= d.graph_calls[k];
      }
      d.rounds_enqueued -= (uint64_t)std::max(1, cfg.threshold_misses);
      d.wave_graph_sig = sig;
      ++d.graph_captures;
    }
    HIPCHK(hipGraphLaunch(d.wave_graph, c.stream));
    for (int k = 0; k < T_KINDS; ++k) { c.round_calls[k] += d.graph_calls[k]; c.kernel_calls[k] += d.graph_calls[k]; }
    d.rounds_enqueued += (uint64_t)std::max(1, cfg.threshold_misses);
    ++d.graph_launches;
  } else {
    d.force_timing = d.graph_enabled && !sharded ? 1 : -1;   // (graph mode: the eager waves are the timed ones, all their rounds)
    dev_enqueue_wave_kernels(sharded, words);
    d.force_timing = -1;
  }
  d.status_copied[slot] = !d.zc_status;
  if (d.zc_status) d.status_of[slot] = d.status_next++;   // (k_wave_end_wide published the block itself)
  else HIPCHK(hipMemcpyAsync(d.h_ctrl.as<sffk::DevCtrl>() + slot, d.ctrl.p, sizeof(sffk::DevCtrl), hipMemcpyDeviceToHost, c.stream));
  HIPCHK(hipEventRecord(slot ? d.ev_wave2 : d.ev_wave, c.stream));
  dev.host_stale = true;
}

// waits for the wave, reads its status block and deals with what the host has to do between waves.  Returns the
// fault the caller has to handle (SFFK_FAULT_LISTS: finish the wave on the host path) or 0; growth faults are
// resolved here (the wave is then resumed by simply enqueuing it again).  stream_idle = false: another wave is
// enqueued behind this one - only the status is read, nothing that needs the stream to be idle is done.
int Forest::dev_finish_wave(double* wait_ms, int slot, bool stream_idle) {
  Ctx& c = *ctx;
  DevEngine& d = dev;
  {
    auto tw = Clock::now();
    HIPCHK(hipEventSynchronize(slot ? d.ev_wave2 : d.ev_wave));
    if (stream_idle) c.sync();   // (harvests the timing events; the stream is idle)
    if (wait_ms) *wait_ms += ms_since(tw);
  }
  if (!d.status_copied[slot]) {
    d.last = d.h_ctrl.as<sffk::DevCtrl>()[d.status_of[slot] & (SFFK_STATUS_RING - 1)];
    if ((uint32_t)d.last.status_seq != d.status_of[slot])
      throw HipError{"device engine: the wave's status block is not the one expected (sequence " + std::to_string(d.last.status_seq) +
                     ", expected " + std::to_string(d.status_of[slot]) + ")"};
  } else {
    d.last = d.h_ctrl.as<sffk::DevCtrl>()[slot];
  }
  if (!stream_idle) return d.last.fault;
  d.host_stale = true;
  const sffk::DevCtrl& s = d.last;
  static const bool prof = getenv("SFFGPU_PROFILE") != nullptr;
  const auto t_ev = Clock::now();
  if (s.fault) {
    const int fault = s.fault;
    if (fault == SFFK_FAULT_INTERNAL) throw HipError{"device engine: a workgroup of the commit waited in vain for a lower one's word"};
    if (fault == SFFK_FAULT_LISTS) return fault;
    if (fault == SFFK_FAULT_CAPACITY) {
      // (nodes and temporaries share the store: grow it, re-place the temporaries) - only the side that is short
      if (s.n_nodes + cfg.wave > d.node_cap - 8) {
        c.store_reserve(std::max(c.store_cap * 2, s.n_nodes + 4 * cfg.wave + 64));
        dev_size_node_arrays();
        if (use_priority()) dev_prio_regrow();   // (entries per heap = node capacity)
      }
      if (s.n_borders + cfg.wave > d.border_cap) dev_size_border_arrays(2 * (s.n_borders + cfg.wave));
    } else if (fault == SFFK_FAULT_BORDER_TABLE) {
      dev_size_border_arrays(std::max(4 * d.border_cap, 2 * (s.n_borders + cfg.wave)));
    } else if (fault == SFFK_FAULT_PRIO_REDRAW) {
      throw HipError{"forest: a priority-frontier draw fell into the rejection zone of the uniform-int algorithm (probability ~ heap size / 2^64); run this forest with SFFGPU_PRIO_SEQ=1"};
    } else {
      throw HipError{"forest: unknown device fault"};
    }
    if (d.table_dirty) {
      sffk::launch_border_rehash(c.stream, dev_view(), s.n_borders);
      d.table_dirty = false;
    }
    int32_t clear[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(reinterpret_cast<char*>(d.ctrl.p) + offsetof(sffk::DevCtrl, fault), &clear[0], 4, hipMemcpyHostToDevice, c.stream));
    HIPCHK(hipMemcpyAsync(reinterpret_cast<char*>(d.ctrl.p) + offsetof(sffk::DevCtrl, halt), &clear[1], 4, hipMemcpyHostToDevice, c.stream));
    HIPCHK(hipStreamSynchronize(c.stream));
    d.last.fault = 0;
    d.last.halt = 0;
    if (prof) fprintf(stderr, "[sffgpu host event] wave %llu: growth fault %d handled in %.2f ms (%d nodes, %d borders)\n",
                      (unsigned long long)s.waves, fault, ms_since(t_ev), s.n_nodes, s.n_borders);
    return 0;   // (in_wave is still set: the next k_wave_begin resumes the wave)
  }
  // the neighbour grid's shared overflow list (checked once per wave like the host path does)
  if (s.grid_ovf > c.gridv.ovf_cap || s.tgrid_ovf > c.tgridv.ovf_cap)
    throw HipError{"neighbour grid overflow list exhausted during a wave (nodes were dropped)"};
  if (s.grid_ovf > c.grid_rebuild_at()) {
    c.store_n = s.n_nodes;
    c.grid_inserted = s.n_nodes;
    c.grid_check();
    if (prof) fprintf(stderr, "[sffgpu host event] wave %llu: node grid re-celled in %.2f ms (%d nodes, overflow list %d, cell %.3f)\n",
                      (unsigned long long)s.waves, ms_since(t_ev), s.n_nodes, s.grid_ovf, c.grid_cell);
  }
  return 0;
}

void Forest::exchange_timer_drop() {
  if (!exchange_open) return;
  ctx->pool.push_back(exchange_t.a);
  ctx->pool.push_back(exchange_t.b);
  exchange_open = false;
}

// a bounded device list ran over and the wave was finished on the host path: the forest's queries go to the wide kernel
// from here on (64 hits per sample instead of 24)
void Forest::on_list_fault() { query_wide = true; }

// one wave of the device engine for a caller that owns the exchange (multi-GPU): begin -> done?
bool Forest::dev_wave_begin() {
  if (!dev.on) throw HipError{"forest: the device engine does not drive this forest"};
  HIPCHK(hipSetDevice(ctx->device));
  exchange_timer_drop();
  if (!dev.active) dev_upload_state();
  const sffk::DevCtrl& k = dev.last;
  if (!k.in_wave && k.terminated) return false;
  dev_ring_top_up(k.cursor, 2 * dev.max_wave_words);
  dev_enqueue_begin();
  dev.host_stale = true;
  return true;
}

bool Forest::seq_eligible() const {
  static const bool off = getenv("SFFGPU_NO_SEQ") != nullptr && atoi(getenv("SFFGPU_NO_SEQ")) != 0;
  return dev.on && cfg.wave == 1 && cfg.world == 1 && !off && !seq_suspended && num_roots <= 64 && !cfg.has_goal && !use_priority();
}

// The scenario tree of k_spec_waves (kernels.h: SpecArgs).  Plain SFF: the full tree of outcomes (accept at attempt
// 0 .. TM-1 | all fail) to SFFGPU_SPEC_DEPTH waves (default 3: 1 + 6 + 36 scenarios at ThresholdMisses = 5, one workgroup
// per scenario and attempt), cut where the grid would no longer be resident at once; SFF*: the chain of all-fail
// scenarios.  SFFGPU_SPEC_SETS sets of workers take the steps in turn (default 1).  SFFGPU_SPEC=0 keeps the single
// wavefront (k_seq_waves).  All read when the forest is created (forest.cpp).
static const size_t SPEC_HEAD = 4096;   // bytes in front of the records: 4 x 64 control-block granules, the step word
bool Forest::spec_setup() {
  DevEngine& d = dev;
  if (d.spec_off) return false;
  const int TM = std::max(1, cfg.threshold_misses);
  if (d.spec_n_sc > 0 && d.spec_tm == TM) return true;
  if (TM > 8) { d.spec_off = true; return false; }
  const int depth = std::max(1, std::min(d.spec_depth > 0 ? d.spec_depth : (cfg.optimize ? 4 : 3), SFFK_SPEC_DEPTH));
  const int sets = std::max(1, std::min(d.spec_sets_want, 4));
  struct Sc { int level; int out[SFFK_SPEC_DEPTH]; int anc[SFFK_SPEC_DEPTH]; int child[9]; };
  std::vector<Sc> tab;
  Sc root{};
  for (int& v : root.child) v = -1;
  tab.push_back(root);
  for (size_t i = 0; i < tab.size(); ++i) {
    if (tab[i].level + 1 >= depth) continue;
    for (int o = cfg.optimize ? TM : 0; o <= TM; ++o) {
      // (every workgroup has to be resident: one wavefront per SIMD at this kernel's register count, two - SFF*: three - per workgroup)
      if ((tab.size() + 1) * (size_t)TM * (size_t)sets * (cfg.optimize ? 3 : 2) + 3 > 960) break;
      Sc ch = tab[i];
      ch.level = tab[i].level + 1;
      ch.out[tab[i].level] = o;
      ch.anc[tab[i].level] = (int)i;
      for (int& v : ch.child) v = -1;
      tab[i].child[o] = (int)tab.size();
      tab.push_back(ch);
    }
  }
  std::vector<int32_t> flat(tab.size() * SFFK_SPEC_TAB, 0);
  for (size_t i = 0; i < tab.size(); ++i) {
    int32_t* t = flat.data() + i * SFFK_SPEC_TAB;
    t[0] = tab[i].level;
    for (int l = 0; l < SFFK_SPEC_DEPTH; ++l) { t[1 + l] = tab[i].out[l]; t[1 + SFFK_SPEC_DEPTH + l] = tab[i].anc[l]; }
    for (int o = 0; o < 9; ++o) t[1 + 2 * SFFK_SPEC_DEPTH + o] = tab[i].child[o];
  }
  d.spec_tab.ensure(flat.size() * 4);
  HIPCHK(hipMemcpy(d.spec_tab.p, flat.data(), flat.size() * 4, hipMemcpyHostToDevice));
  d.spec_n_sc = (int)tab.size();
  d.spec_sets = sets;
  d.spec_tm = TM;
  // control blocks (64 granules per set, 4 sets at most) | cur_step (a line of its own) | records
  d.spec_area.ensure(SPEC_HEAD + (size_t)sets * d.spec_n_sc * TM * (SFFK_SPEC_REC * 8 + 8) + 2048);   // (+ the debugging words, SpecArgs::hb)
  return true;
}

// waves of ONE slot (the reference's own order): k_seq_waves runs whole outer iterations back to back inside one launch,
// thousands of waves per launch; the host tops the engine-word ring up between launches and handles what the round engine's
// host side handles (growth, re-celling, a list overflow -> that wave is finished on the host-replay engine)
void Forest::run_device_seq(int max_waves) {
  Ctx& c = *ctx;
  DevEngine& d = dev;
  HIPCHK(hipSetDevice(c.device));
  auto t0 = Clock::now();
  double wait_ms = 0;
  if (!d.active) dev_upload_state();
  const uint64_t w0 = d.last.waves;
  const uint64_t per_wave = (uint64_t)(8 + std::max(1, cfg.threshold_misses) * (cfg.dim == 2 ? 1 : 6));
  while (true) {
    const sffk::DevCtrl& k = d.last;
    if (!k.in_wave && k.terminated) break;
    if (max_waves > 0 && !k.in_wave && (int)(k.waves - w0) >= max_waves) break;
    if (k.in_wave) {   // (a wave the host-replay engine left unfinished: through the round engine)
      seq_suspended = true;
      st.total_ms += ms_since(t0);
      st.host_ms += ms_since(t0) - wait_ms;
      run_device(1);
      seq_suspended = false;
      t0 = Clock::now();
      wait_ms = 0;
      continue;
    }
    int batch = (int)std::min<uint64_t>(4096, d.ring_words / (2 * per_wave));
    if (max_waves > 0) batch = std::min(batch, max_waves - (int)(k.waves - w0));
    dev_ring_top_up(k.cursor, (uint64_t)batch * per_wave + 16);
    if (d.ring_pending) {
      HIPCHK(hipStreamWaitEvent(c.stream, d.ev_ring, 0));
      d.ring_pending = false;
    }
    sffk::SeqArgs a{};
    a.f = dev_view();
    a.st = sffk::NodeStoreMut{c.sx.as<float>(), c.sy.as<float>(), c.sz.as<float>(), c.syaw.as<float>(),
                              c.spitch.as<float>(), c.sroll.as<float>(), c.stree.as<int32_t>(), c.spos.as<double>()};
    a.g = c.gridv;
    a.env = c.envv;
    a.rob = c.robv;
    memcpy(a.limits, cfg.limits, sizeof a.limits);
    a.dist_tree = cfg.dist_tree;
    a.sampling_dist = cfg.sampling_dist;
    a.sweep_abs_eps = c.sweep_eps();
    a.trig = (cfg.libm_sampling || d.dev_trig) ? d.trig.as<double>() : nullptr;
    a.words_end = d.produced;
    a.grid_ovf_src = c.gridv.ovf_cnt;
    a.dim = cfg.dim;
    a.max_waves = batch;
    a.hit_cap = hit_cap;
    a.grid_ovf_limit = c.grid_rebuild_at();
    a.optimize = cfg.optimize ? 1 : 0;
    if (cfg.optimize) {
      const sffk::StarView sv = star_view();
      a.ktab = sv.ktab;
      a.tree_cnt = sv.tree_cnt;
      a.hist = sv.hist;
      a.hist_ctl = sv.hist_ctl;
      a.hist_cap = sv.hist_cap;
      a.cell_edge = c.grid_cell;
      a.knn_slack = 8 * c.sweep_eps();
    }
    const char* trace_path = getenv("SFFGPU_SEQ_TRACE");
    DevBuf trace_buf;
    if (trace_path) {
      trace_buf.ensure((size_t)batch * 32);
      HIPCHK(hipMemsetAsync(trace_buf.p, 0xff, (size_t)batch * 32, c.stream));
      a.trace = trace_buf.as<int32_t>();
      a.trace_cap = batch;
    }
    const uint64_t waves_before = k.waves;
    const bool spec = spec_setup();
    if (spec) {
      sffk::SpecArgs sa{};
      sa.q = a;
      sa.sc_tab = d.spec_tab.as<int32_t>();
      sa.n_sc = d.spec_n_sc;
      sa.tm = d.spec_tm;
      sa.n_slots = d.spec_n_sc * d.spec_tm;
      sa.n_sets = d.spec_sets;
      sa.base = d.spec_area.as<unsigned long long>();
      sa.cur_step = reinterpret_cast<int32_t*>(d.spec_area.as<uint8_t>() + SPEC_HEAD - 256);
      sa.rec = reinterpret_cast<unsigned long long*>(d.spec_area.as<uint8_t>() + SPEC_HEAD);
      sa.timeout_ticks = 20000000ULL;   // 200 ms
      sa.pipeline = d.spec_pipe ? 1 : 0;
      if (d.spec_test_stall && st.spec_steps == 0 && d.last.spec_steps == 0) sa.test_stall = d.spec_test_stall;
      const size_t rec_bytes = (size_t)sa.n_sets * sa.n_slots * SFFK_SPEC_REC * 8;
      if (getenv("SFFGPU_PROFILE")) sa.hb = reinterpret_cast<unsigned long long*>(d.spec_area.as<uint8_t>() + SPEC_HEAD + rec_bytes);
      HIPCHK(hipMemsetAsync(d.spec_area.p, 0, d.spec_area.cap, c.stream));
      sffk::launch_spec_waves(c.stream, sa);
    } else {
      sffk::launch_seq_waves(c.stream, a);
    }
    HIPCHK(hipMemcpyAsync(d.h_ctrl.as<sffk::DevCtrl>(), d.ctrl.p, sizeof(sffk::DevCtrl), hipMemcpyDeviceToHost, c.stream));
    d.status_copied[0] = true;
    HIPCHK(hipEventRecord(d.ev_wave, c.stream));
    d.host_stale = true;
    const int fault = dev_finish_wave(&wait_ms, 0, true);
    if (trace_path) {
      const size_t nw = (size_t)(d.last.waves - waves_before);
      std::vector<int32_t> tr(nw * 8);
      if (nw) HIPCHK(hipMemcpy(tr.data(), trace_buf.p, nw * 32, hipMemcpyDeviceToHost));
      if (FILE* fp = fopen(trace_path, "ab")) { fwrite(tr.data(), 4, tr.size(), fp); fclose(fp); }
      trace_buf.release();
    }
    if (spec && getenv("SFFGPU_PROFILE")) {
      const int nw = d.spec_sets * d.spec_n_sc * d.spec_tm;
      unsigned long long wq[22];
      HIPCHK(hipMemcpy(wq, d.spec_area.as<uint8_t>() + SPEC_HEAD + (size_t)nw * SFFK_SPEC_REC * 8 + ((size_t)nw + 64) * 8, sizeof wq, hipMemcpyDeviceToHost));
      const unsigned long long* wp = wq + 1;
      const double n = (double)std::max<unsigned long long>(1, wq[0]) * 100.0;
      fprintf(stderr, "[sffgpu k_spec_waves worker us per ACCEPTED attempt, this launch] control block -> scenario + node %.2f | node data + words %.2f sample %.2f "
              "pose %.2f parent edge %.2f neighbour query %.2f neighbour loop %.2f | SFF*: k %.2f k nearest %.2f choose parent %.2f rewire + record %.2f (%llu attempts)\n",
              wp[0] / n, wp[5] / n, wp[6] / n, wp[1] / n, wp[2] / n, wp[3] / n, wp[4] / n, wp[8] / n, wp[9] / n, wp[10] / n, wp[7] / n, wq[0]);
      if (wq[20]) fprintf(stderr, "   k nearest searches: %llu, per search: shells %.2f count groups %.2f item batches %.2f insertions %.2f overflow-list entries %.1f\n", wq[20],
                          (double)wq[16] / wq[20], (double)wq[17] / wq[20], (double)wq[18] / wq[20], (double)wq[19] / wq[20], (double)wq[21] / wq[20]);
    }
    if (spec && d.last.spec_stalled) {   // (its workgroups were not resident together: the single wavefront from here on)
      d.spec_off = true;
      if (getenv("SFFGPU_PROFILE")) {
        fprintf(stderr, "[sffgpu k_spec_waves] a record did not arrive: back to k_seq_waves\n");
        const int nw = d.spec_sets * d.spec_n_sc * d.spec_tm;
        std::vector<unsigned long long> hb((size_t)nw + 8 + 34);
        HIPCHK(hipMemcpy(hb.data(), d.spec_area.as<uint8_t>() + SPEC_HEAD + (size_t)nw * SFFK_SPEC_REC * 8, hb.size() * 8, hipMemcpyDeviceToHost));
        fprintf(stderr, "  leader waited at step %llu scenario %llu attempt %llu set %llu; granules seen:", hb[nw], hb[nw + 1], hb[nw + 2], hb[nw + 3]);
        for (int q = 0; q < 34; ++q) fprintf(stderr, " %llx", hb[nw + 8 + q]);
        fprintf(stderr, "\n  workers (step:phase):");
        for (int w = 0; w < nw; ++w) fprintf(stderr, " %llu:%llu", hb[w] >> 8, hb[w] & 255);
        fprintf(stderr, "\n");
      }
    }
    if (fault == SFFK_FAULT_LISTS) {
      ++st.host_fallback_waves;
      dev_to_host();
      while (in_wave) {
        round_begin();
        int32_t cnt = (int32_t)records.size();
        round_commit(records.data(), cnt, &cnt, 1);
      }
      on_list_fault();
      dev_upload_state();
    }
  }
  st.total_ms += ms_since(t0);
  st.host_ms += ms_since(t0) - wait_ms;
  if (getenv("SFFGPU_PROFILE") && d.last.spec_steps > 0) {
    const sffk::DevCtrl& k = d.last;
    const double sp = (double)k.spec_steps;
    fprintf(stderr, "[sffgpu k_spec_waves leader us/step] publish %.2f first record of the first wave %.2f of the later waves %.2f other records %.2f accepted node %.2f "
            "closed list + termination %.2f erases %.2f other %.2f | %llu steps, %.2f iterations/step, %.2f evaluated/committed\n",
            k.wprof[0] / sp / 100.0, k.wprof[1] / sp / 100.0, k.wprof[7] / sp / 100.0, k.wprof[2] / sp / 100.0, k.wprof[3] / sp / 100.0, k.wprof[4] / sp / 100.0,
            k.wprof[5] / sp / 100.0, k.wprof[6] / sp / 100.0, (unsigned long long)k.spec_steps, (double)k.spec_committed / sp,
            (double)k.spec_evaluated / (double)std::max<uint64_t>(1, k.spec_committed));
  } else if (getenv("SFFGPU_PROFILE")) {
    const sffk::DevCtrl& k = d.last;
    const double it = (double)std::max(1, k.iter);
    fprintf(stderr, "[sffgpu k_seq_waves us/iteration] pick + node %.2f sample %.2f pose %.2f parent edge %.2f neighbour query %.2f "
            "neighbour loop %.2f append %.2f wave end %.2f | %d iterations\n", k.wprof[0] / it / 100.0, k.wprof[1] / it / 100.0,
            k.wprof[2] / it / 100.0, k.wprof[3] / it / 100.0, k.wprof[4] / it / 100.0, k.wprof[5] / it / 100.0, k.wprof[6] / it / 100.0,
            k.wprof[7] / it / 100.0, k.iter);
  }
}

void Forest::run_device(int max_waves) {
  if (seq_eligible()) { run_device_seq(max_waves); return; }
  Ctx& c = *ctx;
  DevEngine& d = dev;
  HIPCHK(hipSetDevice(c.device));
  auto t0 = Clock::now();
  double wait_ms = 0;
  if (!d.active) dev_upload_state();
  const uint64_t w0 = d.last.waves;
  need_host_exchange = false;
  // One wave is kept enqueued AHEAD of the one the host waits for: the status round trip (device -> pinned host ->
  // wake-up -> some thirty launches) otherwise leaves the GPU idle for ~40 us per wave.  Everything the device does is
  // self-guarding - after termination or a fault every kernel of the wave behind returns at once - so the wave ahead
  // is harmless when the wave in front ends the run or needs the host.
  const bool ahead_ok = !getenv("SFFGPU_NO_WAVE_AHEAD");
  int slot = 0;          // status slot of the wave the host waits for next
  bool have_next = false;   // a second wave is enqueued behind it (status slot 1 - slot)
  uint64_t started = 0;  // waves enqueued since the last known status (fresh waves the device may have begun)
  while (true) {
    const sffk::DevCtrl& k = d.last;
    if (!have_next) {      // nothing in flight
      if (!k.in_wave) {
        if (k.terminated) break;
        if (max_waves > 0 && (int)(k.waves - w0) >= max_waves) break;
      }
      dev_ring_top_up(k.cursor, d.max_wave_words);
      dev_enqueue_wave(slot);
      started = 1;
    } else {
      have_next = false;   // the wave enqueued ahead is now the one waited for
    }
    // the wave behind: only from a clean, known state (the in-flight wave is a fresh one, or the resume of one), and
    // only if the caller's wave budget has room for it whatever the in-flight wave turns out to be
    const bool room = max_waves <= 0 || (int)(k.waves - w0) + (int)started + 1 <= max_waves;
    if (ahead_ok && room && !k.fault) {
      dev_ring_top_up(k.cursor, 2 * d.max_wave_words);
      dev_enqueue_wave(1 - slot);
      have_next = true;
    } else {
      // while the GPU works: the words the NEXT wave may need, whatever this one consumes
      dev_ring_top_up(k.cursor, 2 * d.max_wave_words);
    }
    int fault = dev_finish_wave(&wait_ms, slot, !have_next);
    const sffk::DevCtrl& s1 = d.last;
    const bool needs_host = fault != 0 || s1.terminated || s1.grid_ovf > c.grid_rebuild_at() || s1.tgrid_ovf > c.tgridv.ovf_cap;
    if (have_next && needs_host) {
      // the wave behind did nothing (halted device) or - grid overflow list filling up - ran normally: wait for it,
      // then handle whatever the LAST status says with an idle stream
      fault = dev_finish_wave(&wait_ms, 1 - slot, true);
      have_next = false;
      started = 0;
    } else if (have_next) {
      slot = 1 - slot;
      started = 1;         // (the wave ahead may already have begun)
      continue;
    } else {
      started = 0;
    }
    if (fault == SFFK_FAULT_LISTS && cfg.world > 1) {
      // sharded forest: the host protocol of that wave needs the caller's variable-size record exchange
      ++st.host_fallback_waves;
      dev_to_host();
      on_list_fault();
      need_host_exchange = true;
      break;
    }
    if (fault == SFFK_FAULT_LISTS) {
      // a bounded device list overflowed: the round that did is redone on the host path (unbounded lists), the rest of
      // the wave comes back to the device - k_wave_begin resumes a wave in progress (until round 4 the host finished
      // the whole wave: 130 ms for a wave of 16 384 slots instead of ~30)
      ++st.host_fallback_waves;
      dev_to_host();
      const bool whole_wave = getenv("SFFGPU_FALLBACK_WHOLE_WAVE") != nullptr;
      do {
        round_begin();
        int32_t cnt = (int32_t)records.size();
        round_commit(records.data(), cnt, &cnt, 1);
      } while (in_wave && whole_wave);
      on_list_fault();
      dev_upload_state();
    }
  }
  st.total_ms += ms_since(t0);
  st.host_ms += ms_since(t0) - wait_ms;
  if (getenv("SFFGPU_PROFILE")) {
    int32_t why[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpy(why, dev.fault_pending.p, 16, hipMemcpyDeviceToHost));
    {
      unsigned long long pd[16];
      sffk::debug_counters_prio(pd);
      if (pd[0]) fprintf(stderr, "[sffgpu prio heap 0] pops %llu, us per pop %.2f (first loads %.2f), windows per pop %.2f, us per window load %.2f, walk %.2f\n",
                         pd[0], pd[1] * 0.01 / pd[0], pd[2] * 0.01 / pd[0], (double)pd[3] / pd[0], pd[4] * 0.01 / (pd[3] ? pd[3] : 1), pd[5] * 0.01 / (pd[3] ? pd[3] : 1));
      if (pd[0]) fprintf(stderr, "[sffgpu prio heap 0, wave ends] new nodes pushed %llu (gather %.1f ms, pushes %.1f ms) | slot operations %llu, %llu of them removals (gather %.1f ms, apply %.1f ms)\n",
                         pd[6], pd[7] * 1e-5, pd[8] * 1e-5, pd[9], pd[12], pd[10] * 1e-5, pd[11] * 1e-5);
    }
    fprintf(stderr, "[sffgpu samples that sent their round to the host path] hit / neighbour list overflow %d, triangle candidate list %d, walk past a cut neighbour record %d | host fallback waves %llu\n",
            why[1], why[2], why[3], (unsigned long long)st.host_fallback_waves);
  }
  if (getenv("SFFGPU_PROFILE")) {
    const sffk::DevCtrl& k = d.last;
    {
      const double w = (double)std::max<unsigned long long>(1ULL, k.wprof[7]);
      // (k_wave_end_wide is many workgroups: its duration is in the kernel trace; the one-workgroup kernel's phase clocks are gone)
      fprintf(stderr, "[sffgpu us/wave, last workgroup] k_wave_begin %.1f\n", k.wprof[6] / w / 100.0);
    }
    if (d.s_dbg.p) {
      unsigned long long g[32];
      HIPCHK(hipMemcpy(g, d.s_dbg.p, sizeof g, hipMemcpyDeviceToHost));
      const double w = (double)std::max<unsigned long long>(1ULL, g[0]);
      if (g[0] && !g[9])   // (k_star_knn_wg: the store search by its four wavefronts as one section)
        fprintf(stderr, "[sffgpu k_star_knn_wg per accepted sample] us: store search %.1f earlier samples of the round %.1f lists %.1f | longest sample %.1f | samples %llu\n",
                g[1] / w / 100.0, g[3] / w / 100.0, g[4] / w / 100.0, g[7] / 100.0, g[0]);
      if (g[0] && g[9]) {   // (the one-wavefront kernel, SFFGPU_STAR_KNN=lone)
        fprintf(stderr, "[sffgpu k_star_knn per accepted sample] us: cube %.1f shells %.1f mates %.1f lists %.1f | longest %.1f | shells walked %.2f "
                "(samples beyond the cube %.3f) cube candidates %.0f | samples %llu\n", g[1] / w / 100.0, g[2] / w / 100.0, g[3] / w / 100.0,
                g[4] / w / 100.0, g[7] / 100.0, g[5] / w, g[8] / w, g[6] / w, g[0]);
        fprintf(stderr, "[sffgpu k_star_knn cube phase] us: counts %.1f items+distances %.1f bisection %.1f sort+rest %.1f\n", g[9] / w / 100.0,
                g[10] / w / 100.0, g[11] / w / 100.0, g[12] / w / 100.0);
      }
      if (g[16]) {
        const double n = (double)g[16], ps = (double)std::max<unsigned long long>(1ULL, g[17]);
        fprintf(stderr, "[sffgpu k_star_tail, workgroup 0] launches that ran passes %llu, passes each %.2f | us per pass: pass phase %.1f exact phase %.1f "
                "barriers: counting in %.1f waiting for the others %.1f\n", g[16], g[17] / n, g[18] / ps / 100.0, g[19] / ps / 100.0, g[20] / ps / 100.0,
                g[21] / ps / 100.0);
        if (g[24]) fprintf(stderr, "[sffgpu k_star_tail, sections of a sample's pass (STAR_PASS_TRACE build)] us: first loads %.1f views %.1f evaluate %.1f (of it requests %.1f) "
                           "loops %.1f writes %.1f\n", g[24] / ps / 100.0, g[25] / ps / 100.0, (g[26] + g[27]) / ps / 100.0, g[27] / ps / 100.0, g[28] / ps / 100.0, g[29] / ps / 100.0);
      }
    }
    const double r = (double)std::max<unsigned long long>(1ULL, k.prof[6]);
    if (d.kc_trace.p) {
      const size_t nw = (size_t)cfg.wave / 64 + 2;
      std::vector<unsigned long long> tr(nw * 8);
      HIPCHK(hipMemcpy(tr.data(), d.kc_trace.p, nw * 64, hipMemcpyDeviceToHost));
      unsigned long long t0 = ~0ULL;
      size_t used = 0;
      for (size_t w = 0; w < nw; ++w) if (tr[8 * w]) { t0 = std::min(t0, tr[8 * w]); used = w + 1; }
      static const char* names[8] = {"start", "walk done", "count published", "stamps posted", "lower counts in (ids)", "lower stamps in", "lower owned counts in", "end"};
      for (int k = 0; k < 8 && used; ++k) {
        std::vector<double> v;
        for (size_t w = 0; w < used; ++w) if (tr[8 * w + k]) v.push_back((double)(tr[8 * w + k] - t0) / 100.0);
        if (v.empty()) continue;
        std::vector<double> sv = v;
        std::sort(sv.begin(), sv.end());
        fprintf(stderr, "[sffgpu k_commit trace, %zu workgroups] %-24s us after the first start: min %.1f median %.1f max %.1f | first wg %.1f last wg %.1f\n",
                used, names[k], sv.front(), sv[sv.size() / 2], sv.back(), v.front(), v.back());
      }
    }
    // (the LAST workgroup's clock: what the others do is over by then)
    fprintf(stderr, "[sffgpu k_commit us/commit, last workgroup] walks + waits for earlier samples %.1f (%.2f polls, max %llu) "
            "lower workgroups' counts %.1f borders %.1f control block %.1f | dependent/round %.0f\n", k.prof[0] / r / 100.0, k.prof[5] / r,
            (unsigned long long)k.prof[7], k.prof[1] / r / 100.0, k.prof[2] / r / 100.0, k.prof[3] / r / 100.0,
            (double)k.n_unsettled / r);
#ifdef SFFK_CI_TRACE
    {
      std::vector<unsigned long long> tr(4096 * 8);
      sffk::debug_ci_trace(tr.data());
      unsigned long long t0 = ~0ULL, tend = 0;
      for (int w = 0; w < 4096; ++w) if (tr[8 * w]) { t0 = std::min(t0, tr[8 * w]); tend = std::max(tend, tr[8 * w + 7]); }
      struct It { double start, broad, end; int nc, smp, items, wave; double entry, pre; };
      std::vector<It> its;
      int waves = 0, n_pose_items = 0;
      double pose_us = 0, pose_max = 0;
      for (int w = 0; w < 4096; ++w) {
        if (!tr[8 * w]) continue;
        ++waves;
        { const unsigned long long pp = tr[8 * w + 6] >> 8; n_pose_items += (int)(pp & 0xff); pose_us += (double)(pp >> 8) / 100.0; if ((pp & 0xff) && (double)(pp >> 8) / 100.0 > pose_max) pose_max = (double)(pp >> 8) / 100.0; tr[8 * w + 6] &= 0xff; }
        if (!tr[8 * w + 6]) continue;
        its.push_back({(tr[8 * w + 1] - t0) / 100.0, tr[8 * w + 2] >= tr[8 * w + 1] ? (tr[8 * w + 2] - t0) / 100.0 : -1.0, (tr[8 * w + 3] - t0) / 100.0,
                       (int)tr[8 * w + 4], (int)(tr[8 * w + 5] & 0xff), (int)tr[8 * w + 6], w, (tr[8 * w] - t0) / 100.0, ((tr[8 * w + 5] >> 8) - t0) / 100.0});
      }
      std::sort(its.begin(), its.end(), [](const It& a, const It& b) { return a.end > b.end; });
      fprintf(stderr, "[sffgpu k_collide_items trace] %d waves, %zu with items, kernel %.1f us (first start -> last end)\n", waves, its.size(), (tend - t0) / 100.0);
      for (size_t k = 0; k < its.size() && k < 12; ++k)
        fprintf(stderr, "  wave %4d: entry %.1f preamble done %.1f | items %d | first item: start %.1f broad done %.1f end %.1f us | candidates %d masked samples %d\n",
                its[k].wave, its[k].entry, its[k].pre, its[k].items, its[k].start, its[k].broad, its[k].end, its[k].nc, its[k].smp);
      {
        std::vector<double> e, p, st, en;
        for (const It& i : its) { e.push_back(i.entry); p.push_back(i.pre); st.push_back(i.start); en.push_back(i.end); }
        auto q = [](std::vector<double> v, double f) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[(size_t)(f * (v.size() - 1))]; };
        fprintf(stderr, "  entry min/med/max %.1f %.1f %.1f | preamble done %.1f %.1f %.1f | first item start %.1f %.1f %.1f | end %.1f %.1f %.1f\n",
                q(e, 0), q(e, .5), q(e, 1), q(p, 0), q(p, .5), q(p, 1), q(st, 0), q(st, .5), q(st, 1), q(en, 0), q(en, .5), q(en, 1));
      }
      double sum_b = 0, sum_n = 0; int nb = 0;
      for (const It& i : its) if (i.broad >= 0) { sum_b += i.broad - i.start; sum_n += i.end - i.broad; ++nb; }
      if (nb) fprintf(stderr, "  first items with a broad phase: %d, avg broad %.1f us, avg after-broad %.1f us\n", nb, sum_b / nb, sum_n / nb);
      fprintf(stderr, "  pose items: %d, avg %.1f us, longest per wave %.1f us\n", n_pose_items, n_pose_items ? pose_us / n_pose_items : 0.0, pose_max);
    }
#endif
#ifdef SFFK_DEBUG_COUNTERS
    unsigned long long g[16];
    sffk::debug_counters(g);
    const double items = (double)std::max<unsigned long long>(1ULL, g[1]);
    fprintf(stderr, "[sffgpu exact kernel, per masked chunk] chunks %llu (with work %llu, with candidates %llu) flushes %.2f "
            "candidates %.1f | us: setup %.2f hierarchy %.2f narrow %.2f | per wave total %.1f us over %llu waves\n",
            g[0], g[1], g[2], g[3] / items, g[7] / items, g[4] / items / 100.0, g[5] / items / 100.0, g[6] / items / 100.0,
            (double)g[8] / (double)std::max<unsigned long long>(1ULL, g[9]) / 100.0, g[9]);
    fprintf(stderr, "[sffgpu exact kernel, narrow phase per chunk with candidates] us: staging %.2f culls %.2f exact steps %.2f (the rest: pair formation) | "
            "touching (sample, candidate) pairs %.2f\n", g[10] / 100.0 / (double)std::max<unsigned long long>(1ULL, g[2]),
            g[11] / 100.0 / (double)std::max<unsigned long long>(1ULL, g[2]), g[13] / 100.0 / (double)std::max<unsigned long long>(1ULL, g[2]),
            (double)g[14] / (double)std::max<unsigned long long>(1ULL, g[2]));
    unsigned long long q[16];
    sffk::debug_counters_query(q);
    const double qw = (double)std::max<unsigned long long>(1ULL, q[0]);
    fprintf(stderr, "[sffgpu query kernel, per sampled wave] us: grid scan %.2f classify %.2f cull %.2f (flushes %.2f) | live %.2f "
            "pairs/live %.1f survivors/live %.2f\n", q[1] / qw / 100.0, q[2] / qw / 100.0, q[3] / qw / 100.0, q[4] / qw / 100.0,
            q[7] / qw, (double)q[5] / (double)std::max<unsigned long long>(1ULL, q[7]),
            (double)q[6] / (double)std::max<unsigned long long>(1ULL, q[7]));
    fprintf(stderr, "[sffgpu block query kernel] us per sampled workgroup: samples %.2f | cells+lists %.2f | 2nd list %.2f | exact %.2f | classify %.2f | cull %.2f | flush %.2f\n",
            q[1] / qw / 100.0, q[2] / qw / 100.0, q[3] / qw / 100.0, q[4] / qw / 100.0, q[5] / qw / 100.0, q[6] / qw / 100.0, q[7] / qw / 100.0);
    fprintf(stderr, "[sffgpu block query kernel] shader clock over the sampled workgroups' lifetimes: %.0f MHz (lifetime %.2f us)\n",
            100.0 * (double)q[10] / (double)std::max<unsigned long long>(1ULL, q[11]), q[11] / qw / 100.0);
#endif
  }
}

}  // namespace sff
