// engine.h — host side of libsffgpu: device context, batched primitive operations and the
// wave-parallel SpaceForest engine.  C++17, compiled with hipcc; exported through capi.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <map>
#include <unordered_set>
#include <string>
#include <utility>
#include <vector>

#include "../../include/sffgpu.h"
#include "kernels.h"

namespace sff {

struct HipError {
  std::string msg;
};
void hip_check(hipError_t e, const char* what);

// growable device buffer
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  void ensure(size_t bytes);
  void release();
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};
// growable pinned host buffer
struct PinBuf {
  void* p = nullptr;
  size_t cap = 0;
  void ensure(size_t bytes);
  void release();
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// std::mt19937_64 (ISO C++ [rand.predef]) + the libstdc++ distribution algorithms the reference's
// RandGen relies on (src/randGen.h:50-62,149-157)
struct Mt64 {
  uint64_t mt[312];
  int idx;
  uint64_t draws = 0;   // engine words handed out so far (lets a speculative wave rewind)
  // optional queue of words generated ahead of their use (the forest engine fills it while the GPU works);
  // the words come out in the same order either way
  uint64_t* q = nullptr;
  size_t qh = 0, qn = 0;
  explicit Mt64(uint64_t seed = 5489ULL) { reseed(seed); }
  void reseed(uint64_t seed);
  void twist();
  void prefetch(uint64_t* buf, size_t want);   // top the queue (in buf, >= want words large) up to `want` words
  uint64_t next();
  void fill(uint64_t* out, size_t n);          // n consecutive words (same as n calls of next())
  int uniform_int(int lo, int hi);  // uniform_int_distribution<int>(lo,hi) (Lemire multiply-shift)
};

struct HitRec {
  double d;
  int id;
  bool operator<(const HitRec& o) const { return d < o.d || (d == o.d && id < o.id); }
};

// T_COMMIT: the in-order commit of a round (k_commit [/ the SFF* stage] / k_append_sample) - the part every rank
// of a sharded forest repeats; T_EXCHANGE: pack + all-gather + unpack of the answer records
enum TimerKind { T_SWEEP = 0, T_COLLIDE = 1, T_SAMPLE = 2, T_COMMIT = 3, T_EXCHANGE = 4, T_KINDS = 5 };

struct Ctx {
  int device = 0;
  int wall_clock_khz = 100000;                // rate of wall_clock64() on this device
  hipStream_t stream = nullptr;
  hipStream_t own_stream = nullptr;           // the stream the context created (stream may be replaced by the caller's)
  void set_stream(hipStream_t s);             // nullptr = back to the own stream
  hipStream_t copy_stream = nullptr;          // early D2H of a forest round (runs beside the collision kernels)
  hipEvent_t ev_mid = nullptr, ev_early = nullptr;
  std::string err;

  // collision models
  DevBuf env_tri, env_box, env_plane, level_box[SFFK_MAX_LEVELS], rob_tri, env_clear, env_clear_edge, env_ext, env_cand;
  std::vector<double> h_plane, h_rob;   // host copies (robot extents along the triangle normals)
  void build_robot_extents();
  // RCCL communicator of the library's own (multi-GPU device engine; librccl is bound at run time)
  void* rccl_comm = nullptr;
  int rccl_rank = 0, rccl_world = 1;
  void rccl_init(const uint8_t* id128, int rank, int world);
  void rccl_all_gather_i32(const void* send, void* recv, size_t words);
  // the caller's collective instead (sffgpu_ctx_set_allgather): same call sites as the RCCL all-gather
  int (*xchg_fn)(void* user, const void* send_dev, void* recv_dev, size_t words_i32, void* hip_stream) = nullptr;
  void* xchg_user = nullptr;
  int xchg_rank = 0, xchg_world = 1;
  bool can_exchange(int rank, int world) const {
    return (xchg_fn && xchg_world == world && xchg_rank == rank) || (rccl_comm && rccl_world == world && rccl_rank == rank);
  }
  static void rccl_unique_id(uint8_t* id128);
  sffk::EnvView envv{};
  sffk::RobotView robv{};
  bool have_env = false, have_robot = false;
  double env_maxabs = 1.0;
  double env_lo[3] = {0, 0, 0}, env_hi[3] = {0, 0, 0};
  long long clear_cells = 0;   // cells of the clearance grid (0 = none)
  void build_clearance();
  void build_tri_grid();       // cell -> triangle lists for the collision kernels' broad phase
  DevBuf env_tg_start, env_tg_list;

  // node store
  int store_cap = 0, store_n = 0;
  DevBuf sx, sy, sz, syaw, spitch, sroll, stree, spos;
  double store_maxabs = 1.0;

  // uniform grid over the store (forest engine only)
  bool grid_on = false;
  sffk::GridView gridv{};
  DevBuf g_cnt, g_items, g_ovfcnt, g_ovf, g_lite, g_ovf_lite;
  DevBuf t_cnt, t_items, t_ovfcnt, t_ovf, t_occ, t_lite, t_ovf_lite;   // the round's own grid (same cells; filled and emptied every round)
  sffk::GridView tgridv{};
  int grid_inserted = 0;
  void grid_setup(const double limits[6], double cell);
  void grid_insert_new();   // store entries [grid_inserted, store_n)
  void grid_check(bool bulk = false);   // re-cells the grid when the shared overflow list fills up (bulk: an index
                                        // built over an existing store - rebuilt until everything fits)
  double grid_cell = 0, grid_limits[6] = {0, 0, 0, 0, 0, 0};
  int grid_rebuilds = 0;
  int gridv_ovf_cap_next = 65536;
  int grid_bk = 8;          // items per cell bucket (doubled when the cells cannot shrink any further)
  // the shared overflow list is scanned by EVERY query: the grid is re-celled as soon as it holds a few batches' worth
  // overflow entries at which the grid re-cells itself: 128 (every query scans the list) while smaller cells or deeper
  // buckets can still absorb them; once both are exhausted the list is all that is left to grow, and the trigger scales
  // with it again (a fixed 128 would re-insert every node and quadruple the list after every wave)
  bool grid_exhausted = false;
  int grid_rebuild_at() const { return grid_exhausted ? std::max(128, gridv.ovf_cap / 4) : std::min(gridv.ovf_cap / 4, 128); }
  void grid_grow_list();
  static int grid_bk_max() { const char* e = getenv("SFFGPU_TEST_GRID_BKMAX"); return e ? std::max(1, std::min(64, atoi(e))) : 64; }   // (tests: shallow buckets)
  double grid_cell0 = 0;    // the cell edge the forest asked for (re-celling never goes below half of it: a query's
                            // cell count grows with the cube of the ratio)
  int tgrid_ovf_min = 0;    // lower bound of the round grid's overflow list (one wave of samples)

  // scratch
  DevBuf d_a, d_b, d_c, d_d, d_e, d_f, d_g, d_h;
  PinBuf h_a, h_b, h_c, h_d, h_e, h_f, h_g, h_h;
  // round pipeline buffers of the forest engine (kept apart from the batch entry points' scratch)
  DevBuf r_in, r_out, r_q, r_cnt, r_hidx, r_hdist, r_sega, r_segb, r_items, r_items2, r_sub, r_center;
  DevBuf r_qrec;   // per sample: QRec (kernels.h), written by the sampling kernels for k_query_block
  DevBuf r_out2;   // device engine: positions + parent distances of the odd rounds of a wave (see Forest::dev_enqueue_wave_kernels)
  PinBuf p_in, p_out;

  // kernel timing (HIP events on the launch stream)
  struct Timed { hipEvent_t a, b; int kind; bool round; };
  std::vector<Timed> pending;
  std::vector<hipEvent_t> pool;
  double kernel_ms[T_KINDS] = {0, 0, 0, 0, 0};
  uint64_t kernel_launches[T_KINDS] = {0, 0, 0, 0, 0};   // timed launches
  uint64_t kernel_calls[T_KINDS] = {0, 0, 0, 0, 0};      // all launches
  // the forest engine's rounds are sampled (every timer_stride-th is bracketed): their sum is scaled to all rounds,
  // the always-timed batch calls are added as measured
  double round_ms[T_KINDS] = {0, 0, 0, 0, 0};
  uint64_t round_calls[T_KINDS] = {0, 0, 0, 0, 0}, round_timed[T_KINDS] = {0, 0, 0, 0, 0};
  bool round_scope = false;
  bool timing_on = true, timed_now = true;
  int timer_stride = 32;   // (SFFGPU_TIMER_STRIDE; an eagerly launched, event-bracketed wave costs ~0.25 ms more than its graph
                           // replay and its HIP events read ~4 us long per kernel: 8 -> 32 is + 2-3 % on the headline job)
  double kernel_ms_total(int kind) const;

  explicit Ctx(int dev);
  ~Ctx();
  void sync();  // stream sync + harvest timers
  hipEvent_t get_event();
  void time_begin(int kind);
  void time_end();

  void upload_mesh(int role, const double* tri9, int n);
  sffk::NodeStoreView store_view() const;
  void store_reset(int capacity);
  void store_reserve(int capacity);
  void store_append(const double* pos6, const int32_t* tree, int n, bool wait = true);
  void store_set_tree(const int32_t* ids, int n, int32_t tree);  // relabel nodes (tree merging, src/rrt.h:240-250)

  void collide_poses(const double* pos6, int n, uint8_t* hit, bool explicit_rt = false);   // explicit_rt: n x 12 (R, T)
  void collide_segments(const double* a6, const double* b6, int n, uint8_t* is_free, int32_t* first_hit,
                        int32_t* n_samples);
  // the same with the end points given as store ids (fp64 positions of the device store: permanent nodes and the
  // round's temporary entries): 8 bytes per edge over PCIe instead of 96
  void collide_segments_ids(const int32_t* ida, const int32_t* idb, int n, uint8_t* is_free, int32_t* first_hit,
                            int32_t* n_samples);
  // ids < 0 name row -1 - id of the new points the last rrt_chain left on the device (edges of an RRT wave: 8 bytes per edge up)
  void collide_segments_refs(const int32_t* ida, const int32_t* idb, int n, uint8_t* is_free, int32_t* first_hit, int32_t* n_samples);
  const double* rr_np_dev = nullptr;   // the new points of the last rrt_chain (+ rrt_chain_alt): rr_np
  void collide_segments_core(const double* a6, const double* b6, const int32_t* ida, const int32_t* idb, int n,
                             uint8_t* is_free, int32_t* first_hit, int32_t* n_samples, const double* extra_dev = nullptr);
  void sample_steer(const uint64_t* words, const double* center6, int n, double dist, int dim, const double* limits,
                    double* out6, uint8_t* in_limits);
  // exact radius query; results sorted by (dist, id).  Returns per-query totals in cnt.
  void radius(const double* q6, int nq, const double* r, const int32_t* tree, const int32_t* max_id, int32_t* idx,
              double* dist, int32_t* cnt, int cap);
  // tree_by_grid: the caller vouches that every queried tree holds (nearly) all nodes, so per-tree queries may use the
  // index too (sffgpu_nodes_index / the RRT session's grid)
  void knn(const double* q6, int nq, int k, const int32_t* tree, const int32_t* max_id, int32_t* idx, double* dist,
           int32_t* cnt, bool tree_by_grid = false);
  // RRT session, the GPU half of a speculative wave as ONE enqueued chain with one wait (was: three calls, three waits; DESIGN.md
  // 3, "The RRT session's wave").
  // rrt_chain: the two nearest nodes of every steering target in its tree (near_* n x 2; the first is the nearest - two, so that
  // the caller sees a tie) -> the steered new point -> per row (RrtRows): its pose check (hit), the edge nearest -> new point
  // (seg: n_samples | first_hit or INT_MAX | candidate list overflowed, n each), kmax > 0: the kmax nearest nodes of the new
  // point in the tree (mem_* n x kmax), conn_r > 0 (several live trees): every node of the OTHER trees within conn_r of the
  // new point (:228-231; conn_cnt n - > conn_cap: the list ran over, ask Ctx::radius -, conn_idx / conn_d n x conn_cap in no
  // order) -> mate != null: per slot the earlier new point of the wave that would be its nearest node (-1: none).
  // rrt_chain_alt (SFFGPU_RRT_ONE_CHAIN=0; by default the repaired rows ride rrt_chain itself, below): the same rows for the slots
  // repaired by hand of that answer - slot[i] steered from new point mate[i] (a row of the last rrt_chain) -; their new points become
  // rows n.. of the table collide_segments_refs / seg_refs_begin read.
  // Replaces src/rrt.h:143-151,166,228-231.
  struct RrtRows {
    double* np6; uint8_t* hit; int32_t* seg; int32_t* mem_idx; double* mem_d; int32_t* mem_cnt;   // host, caller-owned
    int32_t* conn_idx; double* conn_d; int32_t* conn_cnt;
  };
  // alt_cap > 0 (with mate and R2): the repaired rows ride the same chain - alt_slot / alt_mate / n_alt: the slots that have a mate,
  // in slot order, at most alt_cap of them (the others are the caller's to cut at); R2: their rows (arrays of alt_cap rows)
  void rrt_chain(const double* rnd6, const int32_t* tree, int n, double dist, bool by_grid1, int kmax, bool by_gridk,
                 int32_t* near_idx, double* near_d, int32_t* near_cnt, int32_t* mate, RrtRows& R, double conn_r = 0, int conn_cap = 0,
                 int alt_cap = 0, int32_t* alt_slot = nullptr, int32_t* alt_mate = nullptr, int32_t* n_alt = nullptr, RrtRows* R2 = nullptr);
  void rrt_chain_alt(const int32_t* slot, const int32_t* mate, int n_alt, double dist, int kmax, bool by_gridk, RrtRows& R,
                     double conn_r = 0, int conn_cap = 0);
  int rr_rows0 = 0;   // slots of the last rrt_chain
  DevBuf rr_q1, rr_q2, rr_a, rr_out, rr_sq, rr_np, rr_alt, rr_q2b, rr_sqb;
  // up to four edge batches of references in flight (Rrt::run_wave: the member edges of an RRT* wave in parts)
  struct SegJob { DevBuf ids, a, b, c; PinBuf hids, hc; hipEvent_t ev = nullptr; int n = 0; };
  SegJob seg_job[4];
  void seg_refs_begin(int which, const int32_t* ida, const int32_t* idb, int n, int n_hint);
  void seg_refs_end(int which, uint8_t* is_free, int32_t* first_hit, int32_t* n_samples);
  void seg_finish(const int32_t* hn_in, int n, const double* a6, const double* b6, const void* dev_a, const void* dev_b, uint8_t* is_free,
                  int32_t* first_hit, int32_t* n_samples);
  hipEvent_t rr_ev[3] = {nullptr, nullptr, nullptr};   // rrt_chain: fork after each steer, join before the copy back
  bool rr_fork = true;                                  // SFFGPU_RRT_FORK=0: the chain's queries on the one stream
  PinBuf rr_hq, rr_hout;
  double sweep_eps() const;
  // one sweep launch over the first n_store entries; per-query hit lists sorted by (dist, id)
  void sweep_lists(const double* q6, int nq, const std::vector<double>& r, const int32_t* tree, const int32_t* max_id,
                   const std::vector<uint8_t>& active, int cap, int n_store, std::vector<int32_t>& cnt,
                   std::vector<std::vector<HitRec>>& out, bool sort_lists = true);
};

struct FNode {
  double pos[6];
  int tree, parent, idx_in_tree;
  double d_closest, d_root;
  unsigned iter;
};
struct Border {
  int n1, n2;
  double dist;
};

// Priority frontier (src/heap.h): binary min-heap of node ids keyed by Distance(node, refPoint)
struct PHeap {
  std::vector<int> v;
  const std::vector<FNode>* nodes = nullptr;
  double ref[6];
  double cost(int i) const;
  void bubble_down(int index);
  void bubble_up(int index);
  void push(int n);
  int pop();
  int pop_at(int id);
};

// open-addressing set of 64-bit keys (key 0 is not used by its callers): the border dedup does one insert per
// border event, std::unordered_set spends ~100 ns in each
struct FlatSet64 {
  std::vector<uint64_t> tab;
  size_t used = 0;
  bool insert(uint64_t key) {   // true if the key was new
    if (tab.empty()) tab.assign(1024, 0);
    if ((used + 1) * 10 > tab.size() * 7) grow();
    size_t m = tab.size() - 1, h = (size_t)((key * 0x9E3779B97F4A7C15ULL) >> 20) & m;
    while (tab[h] != 0) {
      if (tab[h] == key) return false;
      h = (h + 1) & m;
    }
    tab[h] = key;
    ++used;
    return true;
  }
  void grow() {
    std::vector<uint64_t> old;
    old.swap(tab);
    tab.assign(old.size() * 2, 0);
    used = 0;
    for (uint64_t k : old) if (k) insert(k);
  }
};

// buffers and bookkeeping of the device-resident engine (forest_dev.cpp / devforest.hip)
struct DevEngine {
  bool on = false;        // this forest runs its waves through the device engine
  bool inited = false;    // fixed-size buffers allocated
  bool active = false;    // the authoritative state currently lives on the device
  bool host_stale = false;// ... and is ahead of the host mirror
  bool table_dirty = false, ring_pending = false;
  DevBuf frontier2, rm_words, rm_pref, slot_pos, act_slot2, w_acc, acc_pref, ustate32, wg_pub, commit_seq, kc_trace;
  DevBuf ord_hist, ord_start, ord_key, ord_rank, ord_pos, ord_lst, ord_cnt;
  bool ord_enabled = true;
  int ord_min_wave = 4096;   // (SFFGPU_ORDER_MIN_WAVE, read when the forest is created)
  DevBuf w_ev, ev_h, ev_nb, ev_raw;   // border events of a round, entered by the append launch (sffk::DevForestView)   // spatial order of a wave's slots (sffk::OrderView)
  DevBuf ctrl, parent, d_root, d_closest, iter, nflag, frontier, closed, claim, slot_node, slot_fail, act_slot, b_n1,
      b_n2, b_ta, b_tb, b_dist, bt_key, bt_val, pair, ring, ulist, d_parent, d_parent2, d_force, fault_pending;
  // priority-frontier mode on the device (devprio.hip; PrioView in kernels.h)
  DevBuf hp_base, hp_size, hp_v, hp_key, hp_pos, hp_ref, hp_gen, hp_cnt, slot_tree, slot_heap, slot_idx, slot_word, hp_plan;
  int prio_heaps = 0, prio_cap = 0;
  // SFF* on the device (devstar.hip; StarView in kernels.h)
  DevBuf s_ktab, s_tree_cnt, s_head, s_mcnt, s_mid, s_md, s_next, s_prop, s_best, s_psel, s_dcl, s_cnt, s_accs, s_hdr, s_changed,
      s_ew, s_ida, s_idb, s_sub, s_segns, s_fh, s_sovf, s_evs, s_evn, s_eve, s_evd, s_acc, s_backup, s_items, s_dbg, s_hist;
  int hist_cap = 0;
  int s_items_cap = 0;
  bool star_inited = false;
  // a wave's launch chain (k_wave_begin, ThresholdMisses rounds, k_wave_end: 30-100 launches) as ONE hipGraph: the
  // host's share of a wave drops from ~4 us per launch to one graph launch.  The graph bakes every kernel argument in,
  // so it is re-captured whenever the signature of those arguments (buffer addresses, sizes, grid geometry) changes.
  hipGraphExec_t wave_graph = nullptr;
  uint64_t wave_graph_sig = 0;
  bool graph_enabled = true;
  int round_parity = 0;              // which set of sample arrays the round being enqueued uses (fused append + sampling)
  int force_timing = -1;             // >= 0: dev_enqueue_round_eval takes this instead of the per-round stride
  bool round_timing = false;         // the timing decision of the round being enqueued (evaluation -> commit)
  uint64_t graph_calls[T_KINDS] = {0, 0, 0, 0, 0};   // timed-kernel launches one replay stands for
  uint64_t waves_enqueued = 0, graph_launches = 0, graph_captures = 0;
  PinBuf h_ctrl, h_ring, h_trig;
  DevBuf trig;   // libm parity mode: the C library's cos / sin / acos of every ring word (3 doubles per word)
  bool dev_trig = false;   // waves of one slot without the parity mode: the same table, filled on the device (k_ring_trig)
  hipEvent_t ev_ring = nullptr, ev_wave = nullptr, ev_wave2 = nullptr;   // (two status slots: one wave may be enqueued ahead)
  int node_cap = 0, border_cap = 0, temp_base = 0;
  uint64_t bt_size = 0, ring_words = 0, max_wave_words = 0;
  uint64_t produced = 0;        // engine words generated so far (absolute position of the generator)
  uint64_t rounds_enqueued = 0;
  int host_nodes = 0, host_borders = 0;   // how much of the device arrays the host mirror holds
  sffk::DevCtrl last{};         // status block after the last completed wave
  bool zc_status = true;        // the wave's last kernel writes the status block into the pinned ring itself (sffk::status_publish)
  uint32_t status_next = 0;     // sequence number the next published block will carry
  uint32_t status_of[2] = {0, 0};   // ... and the ones the two enqueued waves' blocks carry
  bool status_copied[2] = {true, true};   // the slot's block came by a copy into h_ctrl[slot] (k_seq_waves, SFFGPU_NO_ZC_STATUS)
  // waves of one slot, speculated (k_spec_waves): scenario tree, control blocks, records (sffk::SpecArgs)
  DevBuf spec_tab, spec_area;
  DevBuf qclk_sh;               // the query kernel's clock bracket, 64 shards (sffk::DevForestView::qclk_sh)
  int spec_n_sc = 0, spec_sets = 0, spec_tm = 0;
  bool spec_off = false;        // SFFGPU_SPEC=0, or a launch stalled (its workgroups were not resident together)
  int spec_depth = 0 /* 0 = 3 (the tree), SFF*: 4 (the chain) */, spec_sets_want = 1, spec_test_stall = 0;   // (SFFGPU_SPEC_DEPTH / _SETS / SFFGPU_TEST_SPEC_STALL, read when the forest is created)
  bool dev_trig_off = false;
  bool spec_pipe = true;        // SFFGPU_SPEC_PIPE
};

struct Forest {
  Ctx* ctx;
  sffgpu_forest_cfg cfg;
  DevEngine dev;
  bool device_eligible() const;
  sffk::DevForestView dev_view() const;
  sffk::StarView star_view() const;
  void dev_star_setup();            // buffers of the SFF* stage (first use) + per-tree node counts from the host mirror
  void dev_size_node_arrays();
  void dev_size_border_arrays(int want_cap);
  void dev_ring_append(const uint64_t* words, size_t n);
  void dev_ring_top_up(uint64_t cursor, uint64_t ahead);
  void dev_upload_state();
  void dev_to_host();
  void dev_enqueue_begin();
  void dev_enqueue_round_eval(void* send_dev, bool sample = true);
  void dev_enqueue_round_commit(const void* recv_dev, bool sample_next = false);
  void dev_enqueue_end(int slot = 0);
  int dev_finish_wave(double* wait_ms, int slot = 0, bool stream_idle = true);
  void dev_enqueue_wave(int slot);
  void dev_enqueue_wave_kernels(bool sharded, size_t words);   // begin .. end kernels of one wave (what a graph captures)
  uint64_t dev_launch_signature();
  DevBuf x_send, x_recv;   // answer records of a round: this rank's, all ranks' (native RCCL exchange)
  bool need_host_exchange = false;   // run(): a sharded wave has to be finished through round_begin / round_commit
  bool exchange_open = false;        // a T_EXCHANGE timer waits for its closing event (recorded behind the unpack):
  Ctx::Timed exchange_t{};           // its event pair is kept HERE until then (never as an index into Ctx::pending, which a
  void exchange_timer_drop();        // sync in between clears), and returned to the pool if the commit never comes
  bool dev_wave_begin();
  size_t dev_exchange_bytes() const;
  void run_device(int max_waves);
  bool seq_eligible() const;        // waves of one slot, plain SFF: the persistent single-wavefront loop (k_seq_waves)
  void run_device_seq(int max_waves);
  bool spec_setup();                // the speculative kernel's scenario tree and buffers; false = k_seq_waves runs the loop
  bool seq_suspended = false;
  bool test_exchange_self = false;  // SFFGPU_TEST_EXCHANGE_SELF (read when the forest is created): a one-rank forest packs / unpacks its records too
  void sync_host();             // refresh the host mirror (nodes, frontier, borders, counters) from the device
  void fill_stats(sffgpu_forest_stats* out);
  ~Forest();
  Mt64 rng;
  std::vector<uint64_t> rng_ahead;   // engine words generated while the GPU works (fixed size: Mt64 keeps a pointer)
  // cfg.record_parents (SFF*): every node creation and applied rewire as (node, parent from then on, iteration)
  struct HistRec { int32_t node, parent; uint32_t iter; };
  std::vector<HistRec> hist;
  size_t hist_sorted = 0;   // hist[0, hist_sorted) is in iteration order (sffgpu_forest_get_parent_history merges the tail in)
  bool hist_overflow = false;
  std::vector<FNode> nodes;
  // per node, kept apart from the 88-byte records because whole-frontier passes and the per-sample input only
  // need these bits: 1 = Node::ForceChildren, 2 = currently on the frontier deque
  std::vector<uint8_t> nflag;
  std::vector<std::vector<int>> trees;
  std::vector<int> frontier, closed;
  std::map<std::pair<int, int>, std::vector<Border>> borders;
  FlatSet64 border_keys;   // (n1, n2) pairs present in any border list (the reference scans the list)
  std::vector<int> connected;
  int num_roots = 0;   // Problem::GetNumRoots(): roots + the goal tree
  int goal_node = -1;
  int iter = 0;
  bool solved = false, empty_frontier = false;
  sffgpu_forest_stats st{};

  struct Slot { int node; bool from_closed; bool failing; int tree = -1, heap = -1; };
  std::vector<Slot> slots;
  std::vector<std::vector<PHeap>> heaps;   // Tree::frontiers (priorityBias != 0 only)
  bool use_priority() const { return cfg.priority_bias != 0; }
  void dev_prio_upload(int gen = -2);   // host heaps -> PrioView arrays (entries per heap = the device's node capacity); gen: the wave
                                         // whose pops the heaps have behind them (DevCtrl::prio_gen; -2 = none on the device yet)
  void dev_prio_download();   // and back
  void dev_prio_regrow();     // the node capacity has grown: the heaps' rows are laid out again, on the device
  bool tree_frontiers_empty(int t) const;
  bool all_frontiers_empty() const;
  int round = 0;
  bool in_wave = false;

  // one neighbour that can end the reference's neighbour loop (src/forest.h:270-300)
  struct Nb {
    int tree;        // sort key 1: trees are visited in ascending id (:262)
    double d;        // sort key 2: a tree's hits come by ascending distance
    int order;       // sort key 3: index inside the tree (store) / after all store nodes (wave-mates)
    int id;          // store node id, or -1-c for candidate c of this round
    bool same_tree;
    int seg;         // local edge task that decides it
    bool free = false;
    int fh = -1, ns = 0;
  };
  struct Member {    // SFF*: one potential k-nearest neighbour, both edge directions answered
    int id;          // store node id, or -1-c for candidate c of this round
    bool fwd_free = false, bwd_free = false;  // isPathFree(new, nb) (:323) / isPathFree(nb, new) (:336)
    int fwd_fh = -1, fwd_ns = 0, bwd_fh = -1, bwd_ns = 0;
    int seg_f = -1, seg_b = -1;
  };
  struct Cand {      // one sample of the current round
    // --- what reset() writes, in one cache line
    int slot, expanded;
    int accepted_id = -1;
    bool in_lim = false;
    bool answered = false, pose_hit = false, par_free = false;
    bool has_members = false;   // members is non-empty (SFF* only)
    int par_fh = -1, par_ns = 0;
    double pdist = 0;
    // --- filled when the sample is answered
    double pos[6];
    int pose_task = -1, seg_parent = -1;
    std::vector<Nb> nbs;          // valid only while answered (stale entries of earlier rounds otherwise)
    std::vector<Member> members;  // SFF*: candidates for the k-nearest set (src/forest.h:317)
    void reset(int s, int e) {   // reuse across rounds: keeps the vectors' capacity
      slot = s; expanded = e; accepted_id = -1; in_lim = false;
      answered = pose_hit = par_free = false; par_fh = -1; par_ns = 0; pdist = 0;
      if (has_members) { members.clear(); has_members = false; }
    }
  };
  std::vector<Cand> cands;   // storage (only grows); the current round uses the first n_cands
  std::vector<int> round_todo;       // the other candidates, ascending (own shard after round_begin, all ranks' after the exchange)
  const double* round_hpos = nullptr; // the round's sample positions / parent distances / limit flags for ALL
  const double* round_hpd = nullptr;  // candidates (pinned early block, valid until the next round_begin)
  const uint8_t* round_hlim = nullptr;
  std::vector<uint8_t> round_skip;   // per candidate: 1 = the replay has nothing to do (outside the limits / settled by this rank)
  int n_cands = 0;
  std::vector<int32_t> records;  // this rank's answers of the pending round (int32 stream)
  uint64_t bulk_counts[4] = {0, 0, 0, 0};  // collide / path_free / nn counters + number of bulk-settled samples (own shard)
  bool pending_round = false;
  int iter0 = 0, N0 = 0, Tb = 0;  // Tb: 4-aligned base of the round's temporary store entries
  double knn_r = 0;  // running guess of the k-nearest radius (SFF*)
  std::vector<std::vector<HitRec>> knn_out;   // scratch of the SFF* k-nearest passes
  std::vector<int32_t> edge_ia, edge_ib;      // ... and of its edge batch
  int hit_cap = 64, nb_cap = 15;  // device list capacities (env SFFGPU_TEST_HITCAP / _NBCAP shrink them in tests)
  // which neighbour-query kernel serves the rounds: k_query_block (flat work lists of a workgroup, 24 hits per sample,
  // neighbourhood lists) where a sample sees few neighbours - steps at least as long as the angular range, or 2-D - and
  // k_query_classify (one wavefront per sample, 64 hits) where the forest fills the angular dimensions too; a list fault
  // switches to the wide kernel for good
  bool query_wide = false;
  void on_list_fault();
  int star_pass_limit = 0;        // SFF* device stage: most passes of a round's fixed point (0 = the kernels' own limit; SFFGPU_TEST_STAR_PASSES)
  bool star_tail = true;          // ... the passes after the first as one launch (k_star_tail); SFFGPU_STAR_TAIL=0: one launch per pass
  int star_tail_wgs = 0;          // ... its grid's upper bound (0 = one workgroup per CU; SFFGPU_STAR_TAIL_WGS)
  int star_tail_stall = 0;        // ... tests: in every n-th round one workgroup never arrives at the first barrier (SFFGPU_TEST_STAR_STALL=n)

  // post-loop path extraction (src/forest.h:420-462, src/problemStruct.h:184-253)
  struct Holder {            // DistanceHolder (src/primitives.h:598-655)
    int n1 = -1, n2 = -1;
    double dist = 1.7976931348623157e308;
    std::vector<int> plan;
    bool exists() const { return n1 >= 0; }
  };
  std::vector<Holder> nm;    // Solver::neighboringMatrix
  Holder& NM(int i, int j) { return nm[(size_t)(i < j ? i : j) * num_roots + (i < j ? j : i)]; }
  void get_paths();
  void get_all_paths();
  void smooth_paths();   // src/forest.h:464-511, edge checks batched on the GPU

  Forest(Ctx* c, const sffgpu_forest_cfg& cf, const double* roots6, int n_roots);
  int add_node(const double* pos, int tree, int parent, double dclosest, double droot, unsigned it);
  std::vector<Border>& border(int i, int j);
  int max_connected();
  template <class HasBorder> int max_connected_over(HasBorder has_border, std::vector<int>& out_connected) const;
  bool budget_hit() const { return cfg.node_budget > 0 && (int)nodes.size() >= cfg.node_budget; }
  bool terminated() const { return solved || iter >= cfg.max_iterations || budget_hit(); }
  void begin_wave();
  void end_wave();
  void round_begin();
  void round_commit(const int32_t* all, int total_words, const int32_t* counts, int world);
  void run(int max_waves);
  uint64_t fingerprint() const;
};

struct RNode {
  double pos[6];
  int root_tree;   // Node::Root
  int tree;        // Node::ExpandedRoot: the tree whose list currently holds the node
  int parent, idx_in_tree;
  double d_closest, d_root;
  unsigned iter;
};
struct RLink {
  int n1, n2;
  double dist;
};

// RapidExpTree (src/rrt.h:25-44)
struct Rrt {
  Ctx* ctx;
  sffgpu_rrt_cfg cfg;
  Mt64 rng;
  std::vector<RNode> nodes;
  std::vector<std::vector<int>> trees;
  std::vector<std::vector<RLink>> links;
  std::vector<std::vector<int>> eaten;
  std::vector<int> tree_frontier;
  int num_trees = 0, goal_node = -1, iter = 0;
  bool solved = false;
  int lazy_last = -1;                                   // lazy_edge: the node that reached the goal
  double lazy_distance = 1.7976931348623157e308;
  bool lazy_goal_check(int new_id);                     // src/lazy.h:258-273
  sffgpu_rrt_stats st{};

  Rrt(Ctx* c, const sffgpu_rrt_cfg& cf, const double* roots6, int n_roots);
  int add_node(const double* pos, int root_tree, int tree, int parent, double dc, double dr, unsigned it);
  RLink make_link(int a, int b);
  void knn(const double* q, int nq, const int32_t* tree, int k, std::vector<std::vector<int>>& out);
  bool knn_by_grid(const int32_t* tree, int nq, int k) const;
  bool chain_on = true;   // SFFGPU_RRT_CHAIN (read when the session is created): nearest -> steer -> pose -> parent edge -> k nearest as one chain
  int split_parts = 2;    // SFFGPU_RRT_SPLIT (1..4): an RRT* wave's member edges in that many batches, the later ones checked while the earlier ones' rows are replayed
  bool one_chain = true;  // SFFGPU_RRT_ONE_CHAIN: the repaired rows inside the wave's one chain (the device lists them) instead of a second chain
  bool dry_on = true;     // SFFGPU_RRT_DRY: the replay's nearest-node walk done once ahead, so that only the rows it takes get edges
  bool repair_on = true;  // SFFGPU_RRT_REPAIR: slots whose nearest node would be an earlier new point of the wave are evaluated from it too
  int small_mul = 4, small_cap = 48;   // SFFGPU_RRT_SMALL (cap): ... or small_mul x that, up to small_cap slots
  int grow_pct = 150;     // SFFGPU_RRT_GROW: after a cut wave the next one speculates grow_pct % of what survived (+ 1)
  void expand(int tree_to_expand, unsigned iteration);
  void draw_target(double rnd[6]);
  int merge_or_link(int tree_to_expand, int new_id, int nb, bool edge_free, int fh, int ns, int& i);
  int run_wave(int B);   // speculative wave of up to B iterations; returns how many were committed
  void run(int max_iters);
  // post-loop (src/rrt.h:324-352, :381-393)
  struct PathHolder { int n1 = -1, n2 = -1; double dist = 1.7976931348623157e308; std::vector<int> plan; };
  std::vector<PathHolder> nm;         // n_trees x n_trees, symmetric
  std::vector<int> connected;
  void get_paths();
  std::vector<std::vector<int>> link_plans;   // plans of the central tree's links (what smoothPaths works on)
  void smooth_paths();                        // src/rrt.h:354-379, edge checks batched on the GPU
  std::vector<double> pend_pos;       // accepted nodes of the current wave not yet in the device store
  std::vector<int32_t> pend_tree;
  bool defer_append = false;
};

}  // namespace sff
