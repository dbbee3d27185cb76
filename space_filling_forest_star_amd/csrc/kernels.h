// kernels.h — device views and launchers of the gfx950 kernels (kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sffk {

// fp32 SoA node store (the sweep streams these six columns: 24 B per node) plus the
// authoritative fp64 positions and the tree-id column.
struct NodeStoreView {
  const float* x;
  const float* y;
  const float* z;
  const float* yaw;
  const float* pitch;
  const float* roll;
  const int32_t* tree;
  const double* pos;  // n x 6
};

struct NodeStoreMut {
  float *x, *y, *z, *yaw, *pitch, *roll;
  int32_t* tree;
  double* pos;
};

struct SweepQuery {   // 56 bytes, read through the scalar cache
  float x, y, z, yaw, pitch, roll;
  float r2f;          // inflated fp32 squared radius (superset filter)
  int32_t tree;       // -1 = all trees, -2 - t = every tree but t
  double r;           // exact radius (strict <)
  int32_t max_id;     // only node ids < max_id
  int32_t active;
  int32_t pad, pad2;
};

// Uniform grid over xyz (cell edge >= the neighbour radius of the planner): every cell owns a bucket
// of `bk` items; the rare extra items of an over-full cell go to one shared overflow list.
struct GridItem {   // 64 bytes = one sector pair: the node's authoritative fp64 position travels with the item, so a
  double p[6];      // query needs no second, dependent gather for the exact distance (nor for the edge task it writes)
  int32_t id;
  int32_t tree;
  int32_t pad[2];
};
// The same item as a 32-byte fp32 filter record (the store columns' own double -> float casts): the paired query kernel
// (k_query_block) reads these - half the bytes per candidate - and fetches the authoritative fp64 position of the few
// candidates that pass the superset filter from the store (one dependent gather for ~5 of ~34 candidates).
struct GridItem32 {
  float x, y, z, yaw, pitch, roll;
  int32_t id;
  int32_t tree;
};
struct GridView {
  float ox, oy, oz, inv_cell;
  int nx, ny, nz, bk;
  int32_t* cnt;       // per cell: items ever inserted (may exceed bk)
  GridItem* items;    // ncells x bk
  int32_t* ovf_cnt;   // [0] entries in the overflow list
  GridItem* ovf;
  int ovf_cap;
  GridItem32* lite;   // optional ncells x bk: the items as filter records (same cell / slot)
  GridItem32* ovf_lite;
  uint32_t* occ;      // optional occupancy bits (one per cell): the round's own grid is nearly empty, its 100 KB of
                      // bits stay in L2 and spare the queries 27 scattered count loads
};

// What k_query_block needs of a sample, written by the kernel that draws it (one thread per sample, lane-dense) instead
// of being derived by a few lanes of every query workgroup: 32 words, read back with one coalesced load.
struct QRec {
  float q[6];        // 0-5   the sample as the fp32 filter sees it
  float r2f;         // 6     inflated fp32 squared radius
  int32_t max_id;    // 7     only ids below
  int32_t mine;      // 8     tree of the expanded node
  int32_t evaluate;  // 9     inside the limits and owned by this rank
  int32_t q_tree;    // 10    -1 = all trees
  int32_t total;     // 11    cells the query ball's box touches (0 when not evaluated)
  int32_t lx, ly, lz;  // 12-14 its low corner
  int32_t wx;        // 15    its extent in x ...
  int32_t ns0;       // 16    samples of the parent edge (src/problemStruct.h:154-168)
  float t0[6];       // 17-22 parent edge in cells of the clearance grid: start point, step per sample
  int32_t wy;        // 23    ... and in y
  double pdist;      // 24-25 parentDistance (src/forest.h:250)
  double qr;         // 26-27 exact query radius
  int32_t scratch[4];  // 28-31 (the query kernel's own per-sample words)
};
struct SampleParams {
  double limits[6];
  double dist_tree;
  double sweep_abs_eps;
  int rank, world;  // only the queries of candidates i with i % world == rank are active
};

#define SFFK_MAX_LEVELS 4
// 64-ary box hierarchy over the environment triangles (leaf order = Morton order).
struct EnvView {
  const double* tri;      // n_tri x 9, world frame
  const double* tri_box;  // n_tri x 6 (lo xyz, hi xyz), exact
  const double* tri_plane; // n_tri x 5: unnormalised normal, offset n.p1, |n|
  const double* tri_ext;   // n_tri x 2 or null: extent of the un-rotated robot's vertices along that normal (min, max of n.v)
  const double* cand;      // n_tri x 22 or null: {box 6, plane 5, vertices 9, ext 2} of a triangle in one 176-byte record - what the exact
                           // kernels stage per candidate triangle (round 5: one coalesced pass instead of 22 gathers from four arrays)
  int n_tri;
  int n_levels;           // level 0 groups 64 triangles, level k groups 64 boxes of level k-1
  const double* level_box[SFFK_MAX_LEVELS];
  int level_count[SFFK_MAX_LEVELS];
  // clearance bits (built once both meshes are known): bit = 1 when every point of the cell is farther from
  // every triangle than the radius of the sphere around the robot's model origin that holds the robot in any
  // rotation (plus slack), so a robot posed there cannot touch the environment.  The grid spans the environment box inflated by clear_thr; null = not built.
  const uint32_t* clear_bits;
  // second plane over the same cells (round 5): the same statement for a robot that is NOT rotated - what every sample of
  // an edge check is (src/problemStruct.h:157-165) - with the robot's own box instead of its bounding sphere, and the reach
  // of a group of eight consecutive edge samples on top (the cull tests one sample for the group)
  const uint32_t* clear_bits_edge;
  double clear_org[3];
  double clear_inv;       // 1 / cell edge
  int clear_n[3];
  // triangle grid (built once both meshes are known): cell -> the triangles that touch it, CSR.  A query box gets its
  // candidate triangles in two dependent loads (cell ranges, triangle ids) instead of walking the box hierarchy
  // group by group (11 us of the exact kernel's 21 us per work item went there).  null = not built.
  const int32_t* tg_start; // n cells + 1
  const int32_t* tg_list;
  double tg_org[3];
  double tg_inv;
  int tg_n[3];
};
// triangle grid build: per cell the number of triangles touching it (cnt), then - after the host's prefix sum - their ids
void launch_tgrid_build(hipStream_t s, const EnvView& env, int32_t* cnt_or_start, int32_t* list, bool fill);
// clearance bits, two planes (kernels.hip k_clear_scatter).  thr_* = distance from the cell centre below which a triangle can
// block the cell (robot radius + half the cell diagonal + slack; the edge plane adds the reach of a group of eight samples);
// *_lo / *_hi = the box around the cell centre, relative to it, that holds the robot placed anywhere in the cell (pose
// plane: cube of the bounding radius; edge plane: the un-rotated robot's own box + the group's reach)
struct ClearBuildArgs {
  double thr_pose, thr_edge;
  double pose_lo[3], pose_hi[3];
  double edge_lo[3], edge_hi[3];
};
void launch_clear_build(hipStream_t s, const EnvView& env, const ClearBuildArgs& P, uint32_t* bits_pose, uint32_t* bits_edge,
                        long long n_cells);

struct RobotView {
  const double* tri;  // n_tri x 9, model frame
  int n_tri;
  double center[3];   // bounding sphere (model frame)
  double radius;
  double lo[3], hi[3];  // exact box of the un-rotated model
};

// ---- device-resident forest (devforest.hip): the round loop of SpaceForest::Solve with the in-order commit on the
// GPU.  DevCtrl lives in HBM, is advanced by the last workgroup of k_wave_begin / k_commit and by the one-workgroup k_wave_end and is
// copied to the host once per wave.  Every round kernel reads the number of samples from it (the host launches
// grids sized for the wave), and returns at once when `halt` is set (solver terminated, or a fault the host has
// to handle: a device list overflowed and the round must be redone on the host path).
struct DevCtrl {
  int32_t n_act;            // [0] samples (active slots) of the current round
  int32_t halt;             // [1] != 0: every kernel returns immediately
  int32_t n_nodes, iter, round, in_wave;
  int32_t n_slots, frontier_n, closed_n, use_closed;
  int32_t terminated, solved, empty_frontier, fault;
  int32_t N0, iter0, n_borders, n_unsettled;
  int32_t fault_pending, redraws, grid_ovf, tgrid_ovf;
  int32_t front_sel;        // which of the two frontier buffers is current
  int32_t compact_from;     // entries of the other buffer k_frontier_compact has to sift (0 = nothing to do)
  int32_t act_sel, act_cnt; // which of the two active-slot lists is current; its length (>= n_act: the still-failing slots)
  int32_t app_n, app_N0, app_fn0, app_act_sel;   // the commit k_append has to apply (app_n = 0: none)
  int32_t iter0_app, app_act_cnt;
  int32_t claims_done;      // the wave's last k_append(_sample) has posted the exhausted slots' claims (k_wave_end skips that pass)
  int32_t clear_n;          // claims k_frontier_compact has to clear (the failing slots' nodes are parked in ulist)
  unsigned long long cursor;        // engine words consumed so far
  unsigned long long words_base;    // cursor at which the current round's sample words start
  unsigned long long collide_calls, path_free_calls, nn_queries;          // reference-equivalent counters
  unsigned long long poses_executed, segments_executed, samples_executed; // what the GPU actually ran
  unsigned long long waves, rounds, round_nodes, round_queries;
  unsigned long long epoch;         // commit counter (border dedup stamps)
  unsigned long long work_items;    // (edge, chunk) items of all rounds
  // k_commit phase clocks (wall_clock64 ticks of 10 ns, thread 0 of the round's LAST workgroup, SFFGPU_PROFILE only):
  // walks + waits for earlier samples, lower workgroups' counts (ids), borders, totals + control block;
  // [5] = polls of the walk phase, [6] = commits, [7] = most polls
  unsigned long long prof[8];
  // device-clock bracket of the neighbour-query kernel of the current round (first wave in .. last wave out, 100 MHz
  // wall_clock64 ticks) and its sum over all committed rounds: the duration rocprofv3 reports, without the ~3 us a
  // HIP event pair adds around a 20 us kernel
  unsigned long long q_t0, q_t1, q_ticks, q_launches;
  // k_wave_end phase clocks (thread 0): claims, owner flags, closed list, clear claims, removal prefix, termination;
  // [6] = k_wave_begin as a whole, [7] = waves
  unsigned long long wprof[8];
  // priority-frontier mode: nodes when the wave began (the wave's new nodes are pushed at its end), whether the wave took
  // its nodes from the heaps, whether every heap is empty (the mode's "frontier empty")
  int32_t prio_n0, prio_wave, prio_all_empty, prio_gen;   // (prio_gen: the wave whose pops are due, PrioView::gen)
  // SFF* on the device (devstar.hip): rounds whose choose-parent / rewire step ran here, the fixed-point passes they
  // took, the k-nearest members they looked at, the rewires they applied (folded in from StarView::acc by k_wave_end)
  unsigned long long star_rounds, star_passes, star_members, star_rewires;
  // spatial order of the wave's slots (OrderView): 1 = the wave's slots have sorted positions and ord.pos_i is maintained
  int32_t ord_valid;
  // plain SFF: the border events of the committed round are entered by the append launch (k_border_finalize): border list
  // length before that round; n_borders holds an upper bound (every event counted) until the append has run
  int32_t app_nb0;
  // status blocks published so far (DevForestView::host_status): the wave's last kernel writes the control block straight
  // into a ring of pinned host memory, numbered by this counter - no copy launch behind every wave
  int32_t status_seq, status_pad;
  // waves of one slot, speculated (k_spec_waves): steps (publish -> evaluate -> commit), attempts the workers evaluated,
  // attempts that were committed; stalled = a worker's record did not arrive in time (the host goes back to k_seq_waves)
  unsigned long long spec_steps, spec_evaluated, spec_committed;
  int32_t spec_stalled, spec_pad;
};
#define SFFK_STATUS_RING 4
#define SFFK_DEV_MAX_GROUPS 1024   // single-workgroup list kernels: 64 x this many slots per wave at most
#define SFFK_FAULT_LISTS 1        // a hit / neighbour / triangle-candidate list overflowed: redo the round on the host
#define SFFK_FAULT_BORDER_TABLE 2 // the border hash table is full: the host grows it
#define SFFK_FAULT_CAPACITY 4     // node / frontier / border arrays would overflow: the host grows them
#define SFFK_FAULT_INTERNAL 16    // a workgroup of the commit gave up waiting for a lower one's word (never): the host ends the run
#define SFFK_FAULT_PRIO_REDRAW 8   // priority mode: a random-entry draw fell into Lemire's rejection zone (p ~ heap size / 2^64)

// Spatial order of a wave's slots (round 5).  A slot's five samples all lie within one sampling distance of its node, and
// the slots are frontier picks in random spatial order - the query kernel's workgroups (8 samples each) and the 8 XCDs'
// L2s therefore shared nothing: 37 missed 128-byte lines per sample.  k_wave_begin counts the slots per COARSE grid cell
// of their node (hist), its last workgroup turns the counts into bucket starts, the wave's first sampling launch gives
// every slot its sorted position (slot_pos).  The positions are cut into sub-ranges of 64; per sub-range a list of the
// sample indices of its still-failing slots in the CURRENT round (two buffers, named like the active-slot lists by
// DevCtrl::act_sel: the wave's first sampling launch fills the first one position by position, k_append(_sample) appends a
// slot that tries again to the other buffer's list of its sub-range - one atomic per slot, spread over hundreds of
// counters on lines of their own; k_commit empties the buffer the append is about to fill).  The query kernel takes its
// samples from these lists, 8 per workgroup, every XCD one contiguous run of sub-ranges; records, commit order and
// everything else stay keyed by the sample index.  Measured on the bench job: FETCH_SIZE of the query kernel 22.3 ->
// 11.9 MB per launch, L2 misses 384 k -> 236 k.
#define SFFK_ORD_BUCKETS 4096
#define SFFK_ORD_CNT_STRIDE 32    // ints between two sub-range counters (128-byte lines of their own)
struct OrderView {
  int32_t* hist;        // SFFK_ORD_BUCKETS counters, zero between waves; null = off
  int32_t* start;       // SFFK_ORD_BUCKETS bucket starts of the current wave
  int32_t* slot_key;    // per slot: bucket
  int32_t* slot_rank;   // per slot: arrival rank within its bucket
  int32_t* slot_pos;    // per slot: sorted position
  int32_t* lst[2];      // per buffer: 64 entries per sub-range - sample indices of the current round (-1 = none)
  int32_t* cnt[2];      // per buffer: entries of every sub-range's list (SFFK_ORD_CNT_STRIDE ints apart)
  int32_t n_sub;        // sub-ranges the buffers hold (ceil(wave / 64))
  const double* pos;    // node positions (6 doubles per node: the three coordinates of a node lie in ONE line; the float columns would be three)
  float ox, oy, oz, inv_cell;                        // the node grid's cells
  int32_t nx, ny, nz, shift, cnx, cny;              // coarse cell = cell >> shift; cnx x cny x .. coarse cells
  int32_t n_buckets;                                 // coarse cells = buckets in use (<= SFFK_ORD_BUCKETS)
};

struct DevRound {
  const DevCtrl* ctrl;
  const int32_t* act_slot;     // n_act slot indices (ascending): list 0 / list 1, DevCtrl::act_sel names the current one
  const int32_t* act_slot2;
  const int32_t* slot_node;    // node expanded by each slot
  const uint8_t* nflag;        // per node: 1 = ForceChildren, 2 = on the frontier
  const uint64_t* ring;        // engine words (std::mt19937_64 outputs), ring of ring_mask + 1 words
  uint64_t ring_mask;
  const double* trig;          // libm parity mode: 3 doubles per ring word {cos, sin of the word as an angle in [-pi, pi),
                               // acos(1 - 2 u) of the word as u in [0, 1)} evaluated by the HOST's C library; null = the
                               // kernel's own portable trig
  int32_t words_per;           // words per sample: 6 (3-D) / 1 (2-D)
  int32_t* parent_out;         // n: expanded node of every sample (read by k_classify)
  uint8_t* force_out;          // n: its ForceChildren flag
  unsigned long long* qclk;    // DevCtrl::q_t0 / q_t1, reset here for the query kernel that follows
  unsigned long long* qclk_sh; // ... and the 64 shards EVERY workgroup of that kernel reports into (DevForestView::qclk_sh)
  OrderView ord;               // the wave's first sampling launch gives the slots their sorted positions (hist == null: off)
};

// forest rounds only: where k_sample_steer writes the round's temporary store entries and which per-round
// counters it resets (all null / zero for the plain batch entry point)
struct RoundTemps {
  GridView tg;     // per-round grid of the round's own samples (cnt == nullptr: none); same cells as the node grid
  NodeStoreMut st;
  int32_t* cnt;    // n hit counters
  int32_t* ctrl;   // 32 ints zeroed per round (see launch_collide_segments_dyn)
  int32_t* sub;    // optional: the survivor list's SFFK_SUBLISTS counters (SFFK_SUB_STRIDE ints apart), zeroed per round
  int n_perm;      // permanent nodes in the store
  int base;        // 4-aligned index of the first temporary (>= n_perm)
  const double* preset;   // optional n x 6: sample positions computed by the caller (libm parity mode); the kernel
                          // then only applies the limits test and does its bookkeeping
  double* center_out;     // optional n x 6: the expanded node's position, for k_query_classify (which would otherwise
                          // wait for the sample's parent id before it can ask for that position: one dependent load less)
  QRec* qrec;             // optional n: the sample as k_query_block wants it (cells of tg, clearance grid below)
  double clear_org[3];
  double clear_inv;
};

size_t collide_lds_bytes(int n_robot_tri, int waves);

void launch_sample_steer(hipStream_t s, const uint64_t* words, const int32_t* parent, const double* node_pos,
                         const double* center_in, int n, double dist, int dim, const SampleParams& prm, double* out6,
                         uint8_t* in_lim, double* parent_dist, SweepQuery* queries, int32_t q_max_base,
                         const RoundTemps& tmp, const DevRound* dev = nullptr);
struct SampleLaunch;
void launch_sample_steer(hipStream_t s, const SampleLaunch& P);

// grid != nullptr: the written nodes are also inserted into the neighbour grid in the same launch
void launch_store_write(hipStream_t s, const NodeStoreMut& st, const double* pos6, const int32_t* tree,
                        const int32_t* parent, const uint8_t* active, int n, int base, const GridView* grid);

// linear sweep over store entries [first, first + n_nodes) (first must be a multiple of 4)
void launch_sweep(hipStream_t s, const NodeStoreView& st, int first, int n_nodes, const SweepQuery* queries,
                  const double* qpos, int nq, int32_t* cnt, int32_t* hit_idx, double* hit_dist, int cap);
// grid: insert store entries [first, first+n) / answer the queries from the cells their ball touches
void launch_grid_insert(hipStream_t s, const GridView& g, const NodeStoreView& st, int first, int n);
// tg (optional): a second grid with the same cells that holds the round's own samples (filled by k_sample_steer,
// emptied again by k_seg_compact); query i sees its entries with id < max_id like any other
// dev_n (optional, device memory): {number of queries, halt flag} read by the kernel instead of nq
void launch_grid_query(hipStream_t s, const GridView& g, const GridView* tg, const NodeStoreView& st,
                       const SweepQuery* queries, const double* qpos, int nq, int32_t* cnt, int32_t* hit_idx,
                       double* hit_dist, int cap, const int32_t* dev_n = nullptr);
void launch_set_tree(hipStream_t s, int32_t* tree_col, const int32_t* ids, int n, int32_t value);

// ---- exact k nearest (replaces flann::Index::knnSearch: src/forest.h:317, src/rrt.h:143,166,228)
// One wavefront per query keeps the k best (distance, id) pairs it has seen in its lanes (lane j = j-th nearest,
// k <= 64): a batch of 64 candidates is compared with the current k-th distance, the few that beat it are inserted by
// rank (__ballot / __popcll / __shfl_up) - no sort, no atomics, nothing spilled to memory before the end.
struct KnnQuery {
  double pos[6];
  int32_t tree;       // only nodes of this tree (-1 = all)
  int32_t max_id;     // only node ids < max_id
  int32_t k;          // <= 64
  int32_t mate_base;  // grid variant: ids >= mate_base are the round's temporaries, reported separately (below); INT_MAX = none
  int32_t whole_tree; // grid variant: k is the size of the whole tree: every temporary of the tree counts as a mate
  int32_t pad_;
};
#define SFFK_KNN_MATES 64
// linear variant: every store entry [0, n_store) is a candidate (coalesced column reads, fp32 pre-filter against the
// current k-th distance, exact fp64 distance for what passes).  idx / dist: nq x kcap, cnt: nq.
void launch_knn_linear(hipStream_t s, const NodeStoreView& st, int n_store, const KnnQuery* q, int nq, int kcap,
                       int32_t* idx, double* dist, int32_t* cnt, double abs_eps);
// grid variant (forest engines): cubes of cells around the query grow ring by ring until the k-th distance lies
// inside the covered ball.  Besides the k nearest STORE nodes it reports the round's temporaries (ids in
// [mate_base, max_id)) that are not farther than the k-th store node: mate_idx nq x mate_cap, mate_cnt nq
// (> mate_cap = overflow: the caller asks again with a larger list; SFFK_KNN_MATES is the first pass's capacity).
// RRT session: nearest node -> steered point on the device (k_rrt_steer), see Ctx::rrt_chain
void launch_rrt_steer(hipStream_t s, const KnnQuery* q1, const int32_t* idx1, int k1, const double* store_pos, double dist,
                      double* a6, double* np6, KnnQuery* q2, int kmax, int n, SweepQuery* sq, double sq_r, float sq_r2f,
                      double* np_copy, int32_t* seg_ns, int32_t* conn_cnt, const int32_t* alt_slot = nullptr,
                      const int32_t* alt_mate = nullptr, int row0 = 0);
// (seg_ns: 3 n + 16 words - the parent edges' sample counts, result presets and the edge kernels' control words, zeroed here;
// conn_cnt: n hit counters of the other-trees query, zeroed here)
// (sq: the other-trees radius query of every new point; alt_slot / alt_mate: the rows are repaired slots, written from row0 on)
void launch_rrt_alt_list(hipStream_t s, const int32_t* mate, int n, int cap, int32_t* alt_slot, int32_t* alt_mate, int32_t* cnt);
void launch_rrt_mates(hipStream_t s, const KnnQuery* q1, const double* near_d, int k1, const double* np6, const uint8_t* hit,
                      const int32_t* fh, const int32_t* ov, int n, int32_t* mate);
void launch_knn_grid(hipStream_t s, const GridView& g, const GridView* tg, const NodeStoreView& st, const KnnQuery* q, int nq,
                     int kcap, int32_t* idx, double* dist, int32_t* cnt, int32_t* mate_idx, int32_t* mate_cnt, double cell,
                     double slack, int mate_cap = SFFK_KNN_MATES, int n_store = 0);   // n_store > 0: far queries fall back to a sweep of the store

// explicit_rt: pos6 holds n x 12 doubles (row-major rotation + translation) instead of n x 6 pose parameters
void launch_collide_poses(hipStream_t s, const EnvView& env, const RobotView& rob, const double* pos6, int n,
                          const int32_t* live_flags, uint8_t* hit, bool explicit_rt = false);

// arguments of k_classify (one thread per sample of the round)
struct ClassifyArgs {
  int n, N0, cap, nbcap, rank, world;
  int goal_id;              // store id of the goal node, -1 without a goal (src/forest.h:286-287)
  double dist_tree;
  const double* newpos;     // n x 6
  const uint8_t* in_lim;    // n
  const double* pdist;      // n
  const int32_t* parent;    // n (store id of the expanded node)
  const double* center;     // optional n x 6: its position (RoundTemps::center_out); the sample's tree is the tree of its
                            // temporary store entry N0 + i either way
  const uint8_t* force;     // n (Node::ForceChildren of the expanded node)
  const int32_t* cnt;       // n sweep hit totals
  int32_t* hit_idx;         // n x cap (reordered in place)
  double* hit_dist;         // n x cap
  const int32_t* tree;      // store column (incl. the round's temporaries)
  const double* pos;        // store positions (incl. temporaries)
  int32_t* rec_flags;       // n: bit0 = evaluated here, bit1 = list overflow -> host path
  int32_t* rec_nnb;         // n
  int32_t* rec_nb;          // n x nbcap neighbour ids
  int32_t* rec_meta;        // n x nbcap: tree << 1 | same_tree
  double* seg_a;            // n x (1+nbcap) x 6 edge tasks (slot 0 = parent edge)
  double* seg_b;
  int32_t* seg_ns;          // n x (1+nbcap): samples of the edge, -1 = no task
  int32_t* first_hit;       // n x (1+nbcap): preset to INT32_MAX
  int32_t* seg_ovf;         // n x (1+nbcap)
  int32_t* ctrl;            // [1] next task slot of the persistent edge kernel
  const QRec* qrec;         // n records written by the sampling kernel (RoundTemps::qrec), or null (then no k_query_block)
  int lazy_nb;              // device engine: more than nbcap qualifying neighbours cut the record (flag bit 2) instead of faulting
  int wide;                 // the forest asks for k_query_classify (many neighbours per sample: see Forest::query_wide)
  const int32_t* dev_n;     // device mode: {n, halt} (n above is then the launch bound only)
  unsigned long long* qclk; // device mode: {first wave in, last wave out} clock bracket of the query kernel
  unsigned long long* qclk_sh;   // ... its 64 shards (DevForestView::qclk_sh)
  // fused clearance cull (k_query_classify): the wave that wrote a sample's edge tasks looks the clearance bits of
  // their samples (and of the sample's own pose) up right away and appends only the (edge, 64-sample chunk, mask) /
  // pose items that need the exact test to `items` (ctrl[2] = count) - no work-list compaction, no cull kernel
  void* items;              // SurvivorItem[items_cap], split into SFFK_SUBLISTS equal sub-lists (workgroup b appends to
  int items_cap;            // sub-list b % SFFK_SUBLISTS): one counter would see every append of the round, and
  int32_t* sub;             // returning atomics on ONE word saturate near 90 per microsecond chip-wide
  uint8_t* pose_hit;        // n: preset to 0 here, 1 written by the exact kernel
  // device engine: k_query_block takes its samples from the sub-range lists of the wave's spatial order (OrderView) when
  // *ord_valid; ord_nslots = the wave's slots, *ord_sel = the current buffer (DevCtrl::act_sel)
  const int32_t* ord_valid; const int32_t* ord_nslots; const int32_t* ord_sel;
  const int32_t* ord_lst[2]; const int32_t* ord_cnt[2];
};
#define SFFK_SUBLISTS 64
#define SFFK_SUB_STRIDE 64   // ints between two sub-list counters (their own cache lines)
struct SurvivorItem {       // 16 bytes
  int32_t slot;             // edge task slot, or -1 - sample for a pose
  int32_t chunk;
  unsigned long long mask;  // samples of the chunk that need the exact test
};
void launch_classify(hipStream_t s, const ClassifyArgs& a);
// neighbour query + classification in one launch (one wavefront per sample): the hits never leave the wave
// env != nullptr: with the fused clearance cull (a.items / a.pose_hit / a.ctrl[2]); the exact work is then done by
// launch_collide_items
// returns true when the block kernel (k_query_block) ran: it writes the end points of the edge tasks that left a survivor
// only and does not clear unused task slots - launch_collide_items then needs `block_src` (the same arguments)
bool launch_query_classify(hipStream_t s, const GridView& g, const GridView* tg, const NodeStoreView& st,
                           const SweepQuery* queries, const ClassifyArgs& a, const EnvView* env = nullptr);
bool query_block_mode(const GridView& g, const GridView* tg, const ClassifyArgs& a, const EnvView* env);
// where k_collide_items finds a task's end points when the survivor list ran over after k_query_block
struct TaskSource {
  const int32_t* rec_nnb; const int32_t* rec_nb; const int32_t* rec_meta; const int32_t* parent;
  const double* center; const double* pos;
  int nbcap, goal_id, on;
};
// exact collision work of a round from the survivor list k_query_classify wrote (count in ctrl[2])
struct TempGridRef;
void launch_collide_items(hipStream_t s, const EnvView& env, const RobotView& rob, const double* pos6, int n_pose,
                          const int32_t* live_flags, uint8_t* pose_hit, const double* a6, const double* b6,
                          const int32_t* seg_ns, int stride, int32_t* ctrl, const void* items, int items_cap,
                          const int32_t* sub, int32_t* first_hit, int32_t* overflow_flag, const TempGridRef* temps,
                          const int32_t* dev_n = nullptr, const ClassifyArgs* block_src = nullptr);
struct SettleArgs {
  int n, Tb, nbcap, stride, n_trees;
  const uint8_t* in_lim;
  const int32_t* rec_flags;
  const int32_t* rec_nnb;
  const int32_t* rec_nb;
  const int32_t* rec_meta;
  const int32_t* seg_ns;
  const int32_t* first_hit;
  const uint8_t* pose_hit;
  uint8_t* code;                  // n
  unsigned long long* bulk;       // 4 counters (zeroed by k_sample_steer with the ctrl words)
  // device mode (all optional): {n, halt}; a flag raised when a sample needs the host path (a bounded list
  // overflowed); 3 more counters behind bulk[3]: poses / edges / edge samples this rank executed
  const int32_t* dev_n;
  int32_t* fault;
  int count_executed;
};
void launch_settle(hipStream_t s, const SettleArgs& a);
// end points of edges given as store ids -> a6 / b6
// (ids < 0: row -1 - id of `extra`, device points that are not in the store)
void launch_seg_gather(hipStream_t s, const double* store_pos, const int32_t* ida, const int32_t* idb, int n, double* a6,
                       double* b6, const double* extra = nullptr);
void launch_seg_prepare(hipStream_t s, const double* a6, const double* b6, int n, int32_t* seg_ns, int32_t* first_hit,
                        int32_t* ovf);
// ctrl = 16 zeroed ints: [1] scan cursor, [2] work items, [3] list overflow, [4..11] settle counters.
// list = list_cap work items of SFFK_ITEM_BYTES each; masks = list_cap u64 (samples of an item that survive the
// clearance cull).  An overflowing list only costs speed (the exact kernel then scans the slot table).
#define SFFK_ITEM_BYTES 64
void launch_collide_segments_dyn(hipStream_t s, const EnvView& env, const RobotView& rob, const double* a6,
                                 const double* b6, const int32_t* seg_ns, int n_slots, int32_t* ctrl,
                                 void* list, int list_cap, void* masks, int32_t* first_hit, int32_t* overflow_flag);
// a forest round: the same pipeline, and the round's poses go through the cull and the exact kernel with the
// edges (pose_hit: 0 free / 1 hit on return)
struct TempGridRef {   // the round's own grid + the fp32 coordinates of its n samples (store columns at the temp base)
  GridView tg;
  const float *x, *y, *z;
  int n;
};
// empties the cells (and occupancy bits) the round's n samples used in the round's own grid
void launch_tgrid_clear(hipStream_t s, const TempGridRef& t);
void launch_round_collide(hipStream_t s, const EnvView& env, const RobotView& rob, const double* pos6, int n_pose,
                          const int32_t* live_flags, uint8_t* pose_hit, const double* a6, const double* b6,
                          const int32_t* seg_ns, int n_slots, int32_t* ctrl, void* list, int list_cap, void* masks,
                          int32_t* first_hit, int32_t* overflow_flag, const TempGridRef* temps,
                          const int32_t* dev_n = nullptr, int stride = 0);

// ---- device-resident forest: state views + the single-workgroup kernels that advance it
// Priority-frontier mode (Problem::priorityBias != 0; src/forest.h:78-88,126-147,160-181,360-363): every tree keeps one
// binary heap per OTHER tree (ordered by the distance to that tree's root; with a goal: one heap, ordered by the distance
// to it), a slot takes its node from a random heap of a random tree - the minimum with probability priorityBias, a random
// ENTRY of the heap array otherwise, so the array order of src/heap.h (BubbleUp / BubbleDown / pop-at-index) is part of
// the result.  The heaps live in HBM as (node id, key) arrays with a position map (the reference finds a node in a heap
// by linear search, src/forest.h:169-171); heaps are independent of each other, so their operations of a wave - pops at
// its beginning, pushes of the accepted nodes and the exhausted slots' removals at its end, in the reference's order -
// run one workgroup per heap (devprio.hip).
struct PrioView {
  int n_heaps;                 // 0 = mode off
  int cap;                     // entries per heap (= node capacity)
  const int32_t* base;         // per tree: its first heap; base[n_trees] = n_heaps
  int32_t* size;               // per heap: entries
  int32_t* v;                  // n_heaps x cap: node ids in the reference's array order
  double* key;                 // n_heaps x cap: Distance(node, refPoint) of the entry
  int32_t* pos;                // n_heaps x cap: where node id sits in the heap, -1 = not in it
  const double* ref;           // n_heaps x 6: refPoint
  int32_t* slot_tree; int32_t* slot_heap; int32_t* slot_idx;   // per slot: tree, heap within the tree, pop-at index (-1 = pop the minimum, -2 = draw it from slot_word)
  unsigned long long* slot_word;   // per slot: the engine word of its random-entry draw (k_prio_plan)
  int32_t* plan;               // k_prio_plan's jump tables: (log2(wave) + 2) x (4 wave + 16) ints, or null (sequential picks only)
  int32_t* counters;           // [0] heaps through with k_prio_end, [1] non-empty ones among them
  int32_t* gen;                // per heap: the wave (DevCtrl::prio_gen) whose pops it has done
  double bias;
};
struct DevForestView {
  DevCtrl* ctrl;
  PrioView prio;
  OrderView ord;
  // node records beside the store columns (store_view / NodeStoreMut)
  int32_t* parent; double* d_root; double* d_closest; uint32_t* iter; uint8_t* nflag;
  int32_t* frontier; int32_t* frontier2;                   // two buffers: compaction sifts from one into the other
  int32_t* closed; int32_t* claim;                         // claim: per node scratch (INT_MAX between uses)
  unsigned long long* rm_words; int32_t* rm_pref;          // frontier positions removed by the wave (bit per entry)
  int32_t* slot_node; int32_t* slot_pos;
  int32_t* act_slot; int32_t* act_slot2;                   // the still-failing slots in slot order (two buffers)
  // borders: append-only list + open-addressing table of (n1, n2) keys with a commit stamp
  int32_t* b_n1; int32_t* b_n2; int32_t* b_ta; int32_t* b_tb; double* b_dist;
  unsigned long long* bt_key; unsigned long long* bt_val; unsigned long long bt_mask;
  uint8_t* pair;               // n_trees x n_trees: 1 = the pair has a border entry
  const uint64_t* ring; uint64_t ring_mask;
  int32_t node_cap, border_cap, wave, n_trees, words_per, threshold_misses, max_iterations, node_budget;
  int32_t temp_base;           // store index of the round's temporaries
  int32_t goal_id;             // Problem::hasGoal: the goal's node (a one-node tree that is searched, never expanded), -1 = none
  int32_t* ulist;              // scratch list of a wave's slots (k_wave_end)
  // one 64-bit word per 64 samples: the accepted samples, and how many were accepted before the word (k_commit ->
  // k_append_sample / the star stage: node ids and the next round's active list without walking the samples)
  unsigned long long* w_acc; int32_t* acc_pref;
  // border events of a round (plain SFF: entered by the append launch): the samples with an event, per 64 samples; per
  // sample the table entry its stamp went to, the neighbour's node id and the raw neighbour (store id or temporary)
  unsigned long long* w_ev; unsigned long long* ev_h; int32_t* ev_nb; int32_t* ev_raw;
  // k_commit (the wide commit kernel): what its workgroups tell each other.  Every word carries the launch's sequence
  // number (commit_seq[0] + 1) in its upper half, so nothing is ever cleared and a stale word is never taken for news.
  int32_t* ustate32;           // per sample: (seq << 2 | state), state 1 rejected / 2 accepted / 3 rejected + border event
  unsigned long long* wg_pub;  // SFFK_PUB_WORDS words per workgroup of 64 samples (one 128-byte line), see k_commit
  int32_t* commit_seq;         // [0] = launches that reached their end so far; [1] workgroups of k_wave_begin that are through, [2] one of them met a redraw;
                               // [3] = k_wave_end_wide launches that ended a wave, [4] its workgroups that are through
  unsigned long long* kc_trace; int32_t kc_trace_round;   // SFFGPU_KC_TRACE=<round>: 8 clock reads per workgroup of that round's k_commit
  DevCtrl* host_status;        // SFFK_STATUS_RING control blocks in pinned host memory (device-visible); null = the host copies
  // Device-clock bracket of the neighbour-query kernel, by EVERY workgroup (round 6; round 5 sampled every 16th one and so
  // could miss the last workgroup out): 64 shards of {latest end, earliest start} on lines of their own (16 words apart) -
  // thread 0 of workgroup b adds its two clock reads to shard b % 64 with atomics that return nothing; the sampling launch
  // in front resets them, k_commit's last workgroup folds them into DevCtrl::q_t0 / q_t1
  unsigned long long* qclk_sh;
  int32_t profile;             // SFFGPU_PROFILE: the single-workgroup kernels read their phase clocks (a clock read is a scalar
                               // memory round trip: a dozen of them is microseconds)
};
#define SFFK_PUB_WORDS 16
#define SFFK_PRIO_MAX_HEAPS 1024
// ---- SFF* (optimize = true) on the device engine: choose-parent + rewire of src/forest.h:307-351 (devstar.hip).
// The accept / reject logic does not depend on costs, so k_commit settles WHICH samples of the round become
// nodes (and their ids) exactly as for plain SFF; then, for the accepted samples only:
//   k_star_knn(_wg) one wavefront (_wg: one workgroup, kernels.hip) per accepted sample: its k = floor(2e log10(#nodes at its turn)) nearest nodes of its
//                 tree among the store AND the samples accepted earlier in the round (replaces knnSearch, :317); every
//                 member joins the toucher list of its node (per-node linked lists, heads stamped with the round's
//                 epoch: no clearing)
//   k_star_pass   the sequential semantics "sample i sees the rewires of every accepted sample before it" as a fixed
//                 point: a sample's view of a member's DistanceToRoot = the proposal of the LATEST earlier sample that
//                 rewires that node (walk of the node's toucher list), else the node's stored cost; from its views it
//                 recomputes its parent / cost / rewire proposals.  Dependencies only point backwards in slot order, so
//                 the iteration reaches the unique fixed point in (longest chain + 1) passes; a pass that changes
//                 nothing proves it.  The first pass is a launch of its own; the later ones run inside ONE launch,
//                 k_star_tail (kernels.hip: pass / exact phases of the first workgroups of a resident grid, a barrier
//                 over them in between, the exchanged words written through / read from memory) - or, without it, as a
//                 chain of one launch per pass whose later launches return at once.
//                 Member edges (new -> member :323, member -> new :336) are answered lazily: only the edges the two loops
//                 can reach given the views are looked at - the pass culls their samples against the clearance bits
//                 itself (most edges are answered right there), the rest goes to k_star_exact between two passes.
//   k_star_apply  accepted samples -> nodes (store, records, grid, frontier); per node the LAST active rewire in slot
//                 order is written (descendants' costs are NOT propagated, as in the reference, :344-348)
#define SFFK_STAR_KC 64        // member slots per sample (k <= 50 for any int32 node count; lane k holds the expanded node)
#define SFFK_STAR_KMAX 56
#define SFFK_STAR_PASSES 8     // counter sets / changed flags (pass p uses set p mod 8); the fixed chain's most passes per round
#define SFFK_STAR_TAIL_PASSES 48  // k_star_tail: most passes per round; not converged by then = fault (the round is redone on the host)
#define SFFK_STAR_BAR 32       // index in StarView::changed of k_star_tail's barrier counter (its own cache line)
#define SFFK_STAR_ACC 8        // sub-counter words per line of StarView::acc
#define SFFK_STAR_SUB 16       // ints between two survivor sub-list counters (their own cache lines)
struct StarView {
  const int32_t* ktab;         // ktab[m] = smallest node count N with floor(2e log10 N) >= m (host libm, the reference's expression)
  int32_t* tree_cnt;           // nodes per tree in the store, one counter per 64 bytes (16 ints apart)
  unsigned long long* head;    // per node: (epoch << 32 | pair + 1) = head of this round's toucher list
  int32_t* m_cnt;              // per sample: members
  int32_t* m_id;               // W x KC node ids: store id, or N0 + rank of an earlier accepted sample of the round
  double* m_d;                 // W x KC distances sample <-> member
  int32_t* next;               // per pair (sample * KC + m): next pair + 1 in its node's list, 0 = end
  double* prop;                // per pair: proposed DistanceToRoot when the rewire is active (:336), +inf otherwise
  double* best; int32_t* psel; double* dcl;   // per sample: cost, chosen parent (node id), distance to it (:320-329)
  unsigned long long* cnt;     // per sample: {Collide calls, isPathFree calls} of its choose-parent / rewire loops
  int32_t* acc_sample;         // rank among the accepted samples -> sample (k_commit)
  int32_t* hdr;                // {accepted samples, skip, border entries of the round, first of them, fault, passes run, converged}
  int32_t* changed;            // SFFK_STAR_PASSES flags: pass t changed something / still waits for an edge; [SFFK_STAR_BAR]: k_star_tail's barrier
  // member edges, answered LAZILY: edge slot = (sample * KC + m) * 2 + dir (0: new -> member :323, 1: member -> new :336).
  // ew: 0 = never asked for; -1 = on the exact kernel's list (answer in first_hit after the launch that follows the pass);
  // else ((Collide calls << 1 | free) << 1) | 1.  Only the edges the loops can reach at all are ever looked at.
  int32_t* ew; int32_t* ens; int32_t* first_hit; int32_t* seg_ovf; int32_t* ida; int32_t* idb;
  int32_t* sub;                // survivor sub-list counters: SFFK_STAR_PASSES x SFFK_SUBLISTS x SFFK_STAR_SUB ints
  void* items; int items_cap;  // SurvivorItem list of the pass in flight (SFFK_SUBLISTS equal sub-lists)
  // border entries created by the round (k_commit): the two nodes' costs are read "at the time of the sample"
  int32_t* ev_sample; int32_t* ev_nb; int32_t* ev_ex; double* ev_dist;
  unsigned long long* acc;     // 64 lines x SFFK_STAR_ACC: Collide calls, isPathFree calls, rounds, passes, members, rewires
  DevCtrl* backup;             // the control block as a rolled-back round leaves it (restored when the star stage faults)
  unsigned long long* dbg;     // SFFGPU_PROFILE: 32 counters of the star kernels (null = off)
  // record_parents: (node, parent, iteration) triples appended by k_star_apply - one per created node and per ACTIVE
  // rewire (also those a later sample of the round overrides); hist[0] of hist_ctl = entries, [1] = ran over
  int32_t* hist; int32_t* hist_ctl; int hist_cap;
};
// per-sample verdicts (A.code)
#define SFFK_DEPENDS 0     // (host engine's k_settle: the neighbour walk reached a sample of the same round first)
#define SFFK_REJECTED 1
#define SFFK_OUTSIDE 2
#define SFFK_ACCEPT 3
#define SFFK_REJECT_EVENT 4 // rejected by a free edge to another tree: border entry with neighbour dk
struct ResolveArgs {
  DevForestView f;
  NodeStoreMut st;
  GridView g;
  int nbcap, stride;
  int rank, world;             // multi-GPU: which samples this rank evaluated itself (executed-work counters)
  const double* newpos; const double* pdist; const int32_t* parent; uint8_t* code;
  const uint8_t* in_lim; int32_t* rec_flags; uint8_t* pose_hit;
  int32_t* rec_nnb; int32_t* rec_nb; int32_t* rec_meta; int32_t* seg_ns; int32_t* first_hit;
  unsigned long long* bulk;    // (host engine's k_settle: counters of the samples it settled, 7 words)
  const int32_t* round_ctrl;   // the round's scratch block ([2] = work items)
  int32_t* fault_pending;
  int star;                    // SFF*: the accepted samples are appended by the star stage (S below)
  StarView S;
};
void launch_wave_begin(hipStream_t s, const DevForestView& f);
// priority-frontier mode (devprio.hip): picks + pops before the rounds, pushes + removals behind them; the position map
// of freshly uploaded heaps
void launch_prio_begin(hipStream_t s, const DevForestView& f);
void launch_prio_end(hipStream_t s, const DevForestView& f, const NodeStoreView& st);
void launch_prio_index(hipStream_t s, const PrioView& p);
// ---- waves of ONE slot = the reference's own loop order (src/forest.h:122-202): one persistent wavefront runs whole
// outer iterations - frontier pick, up to ThresholdMisses x (sample, pose check, parent edge, 27-cell neighbour query,
// the neighbour edges in the order the reference reaches them, append), closed list / frontier erase, termination - for as
// many waves as the engine words last, instead of ~33 launches per wave.  Plain SFF only.  Evaluation is lazy exactly like
// the reference's (an edge is only checked when the loop gets to it), so the reference-equivalent counters ARE the
// executed ones.  Stops early (halt + fault in the control block, the faulted attempt rolled back) where the round
// engine would: a bounded list overflowed, arrays / border table to grow, the grid's overflow list to re-cell.
struct SeqArgs {
  DevForestView f;
  NodeStoreMut st;
  GridView g;
  EnvView env;
  RobotView rob;
  double limits[6];
  double dist_tree, sampling_dist, sweep_abs_eps;
  const double* trig;                 // libm parity mode (DevRound::trig) or null
  unsigned long long words_end;       // engine words resident in the ring (absolute position)
  int32_t* grid_ovf_src;              // the node grid's overflow counter
  int dim, max_waves, hit_cap, grid_ovf_limit;
  // SFF* (optimize): choose-parent + rewire run in the loop itself, in the reference's order (src/forest.h:307-351)
  int optimize;
  const int32_t* ktab;                // StarView::ktab
  int32_t* tree_cnt;                  // StarView::tree_cnt
  int32_t* hist; int32_t* hist_ctl; int hist_cap;   // StarView's parent history (record_parents) or null
  double cell_edge, knn_slack;
  // SFFGPU_SEQ_TRACE=<file>: per wave {node, pick, iteration at its start, cursor at its start, outcome (attempt that was
  // accepted, ThresholdMisses = none), nodes} - 8 ints, indexed by the launch's wave number; null = off
  int32_t* trace; int trace_cap;
};
void launch_seq_waves(hipStream_t s, const SeqArgs& a);
// ---- the same loop, SPECULATED over many wavefronts (round 6, k_spec_waves).  One wavefront computes an attempt in
// ~10-15 us however idle the other 1 023 SIMDs are; but what an attempt of the NEXT waves will be is known in advance
// up to one small unknown per wave - which of its ThresholdMisses attempts, if any, is accepted (src/forest.h:138-178: the
// frontier pick is a Lemire draw from the pool, a wave that fails erases its node order-preservingly, a wave that accepts
// at attempt r has consumed 1 + 6 (r + 1) engine words and appended one node).  So a STEP evaluates a small tree of
// scenarios side by side, one workgroup of one wavefront per (scenario, attempt): scenario = the outcomes (accept at
// attempt r | all fail) assumed for the waves before it.  Every worker derives its scenario's frontier pick and word
// position from the published control block alone (scalar arithmetic on ring words), evaluates its attempt against the
// frozen store plus the samples its scenario assumes accepted (read from their workers' records), and writes a record;
// nothing else.  The LEADER (workgroup 0) owns the forest: it walks the records in the reference's order - it recomputes
// every pick itself and only takes a scenario's records when node, word position, iteration and node count agree with its
// own state - applies counters, border events, the accepted node, the closed list / frontier erase and the termination
// tests exactly like k_seq_waves, follows the tree by the outcome that really happened, and publishes the next step.
// Everything a worker reads of what the leader wrote in the same launch is stored write-through (`sc1`) and loaded `sc1`;
// control block and records travel as 8-byte {value, step} granules, so no fence and no flag is needed.
// SFF*: a chain of all-fail scenarios only (an accepted node's rewires change costs the later attempts would read).
#define SFFK_SPEC_DEPTH 4          // most waves of one step
#define SFFK_SPEC_TAB 20           // ints per scenario: level | out[DEPTH] | anc[DEPTH] | child[9] | pad
#define SFFK_SPEC_REC 400          // granules per record: row 0 (64) | SFF* rewires (64 x 5) | early row (16)
#define SFFK_SPEC_EARLY 384
struct SpecArgs {
  SeqArgs q;
  const int32_t* sc_tab;              // the scenario tree (host-built): SFFK_SPEC_TAB ints per scenario
  int n_sc, n_slots, n_sets, tm;      // scenarios; workers per set = n_sc x ThresholdMisses; sets take the steps in turn
  unsigned long long* base;           // per set: 16 granules = the control block of its current step
  unsigned long long* rec;            // n_sets x n_slots records of SFFK_SPEC_REC granules
  int32_t* cur_step;                  // the step the leader is at (workers of older steps give up), -1 = the launch is over
  unsigned long long timeout_ticks;   // a record that is not there after this many 10 ns ticks = stalled
  int pipeline;                       // plain SFF: a step is published before the accepted nodes of the one before are written (SFFGPU_SPEC_PIPE=0: after)
  int test_stall;                     // tests (SFFGPU_TEST_SPEC_STALL = 8 x step + slot): that worker never writes that step's record
  unsigned long long* hb;             // debugging (SFFGPU_PROFILE): per worker (step << 8 | phase), [n_sets x n_slots ..]: the leader's last wait; null = off
};
void launch_spec_waves(hipStream_t s, const SpecArgs& a);
// the transcendental values of n engine words (3 doubles per word: cos, sin of the word as an angle, acos of it as the pitch draw)
void launch_ring_trig(hipStream_t s, const uint64_t* words, double* trig, int n);
// multi-GPU: the answer record of one sample as it travels in the all-gather of a round:
// flags, nnb, pose_hit, 0 | nb[nbcap] | meta[nbcap] | seg_ns[1 + nbcap] | first_hit[1 + nbcap]
inline int record_words(int nbcap) { return 6 + 4 * nbcap; }
// pack: the samples this rank owns (i % world == rank) -> send[(i / world) * record_words ...];
// unpack: recv = world segments of ceil(n_bound / world) records; every sample another rank owns is copied back
// into the round's arrays (a.code / rec_* / seg_ns / first_hit / pose_hit)
void launch_pack_records(hipStream_t s, const ResolveArgs& a, int rank, int world, int n_bound, int32_t* send);
void launch_unpack_records(hipStream_t s, const ResolveArgs& a, int rank, int world, int n_bound, const int32_t* recv);
struct StarLaunch {            // what the host adds for the SFF* stage of a commit
  GridView g, tg;
  NodeStoreView st;
  EnvView env;
  RobotView rob;
  double cell_edge, slack;
  double cube_reach;         // k_star_knn gathers the cube of cells that covers this radius in one go (~ 2 sampling distances)
  int passes;                // most passes of the fixed point per round (0 = the kernels' limit; tests shrink it to drive the fault path)
  int tail;                  // the passes after the first as one launch (k_star_tail) instead of one launch per pass
  int tail_wgs;              // ... bound of its grid (0 = one workgroup per CU)
  int tail_stall;            // ... tests: every n-th round one workgroup stays away from the first barrier (the time-out's fault path)
};
void launch_star_stage(hipStream_t s, const ResolveArgs& a, int n_bound, const StarLaunch& L);   // devstar.hip
// exact collision test of the member-edge chunks a star pass could not answer from the clearance bits (kernels.hip)
void launch_star_tail(hipStream_t s, const ResolveArgs& a, const EnvView& env, const RobotView& rob, const NodeStoreView& st,
                      int n_bound, int max_passes, int wgs_bound, int test_stall);
void launch_star_exact(hipStream_t s, const EnvView& env, const RobotView& rob, const double* store_pos, const StarView& S,
                       int pass);
// the commit of one round: k_commit (wide) [-> the SFF* stage] -> k_append / k_append_sample (wide);
// n_bound = launch bound
// the arguments of k_sample_steer as one block: the commit's last kernel (k_append_sample) also draws the NEXT round's
// samples - a slot that was not accepted knows its place in the next round's list the moment it writes it
struct SampleLaunch {
  const uint64_t* words; const int32_t* parent; const double* node_pos; const double* center_in;
  int n; double dist; int dim; SampleParams prm;
  double* out6; uint8_t* in_lim; double* parent_dist; SweepQuery* queries; int32_t q_max_base;
  RoundTemps tmp; DevRound dv;
};
// SFF*: the k nearest of every accepted sample, one workgroup per sample (k_star_knn_wg in kernels.hip; devstar.hip's
// k_star_knn - one wavefront per sample - remains behind SFFGPU_STAR_KNN=lone)
void launch_star_knn_wg(hipStream_t s, const ResolveArgs& a, const GridView& g, const GridView& tg, const NodeStoreView& st,
                        double cell_edge, double slack, int n_bound, int R0);
void launch_commit(hipStream_t s, const ResolveArgs& a, int n_bound, const StarLaunch* star = nullptr,
                   const SampleLaunch* next = nullptr);
void launch_wave_end(hipStream_t s, const DevForestView& f, const int32_t* grid_ovf, const int32_t* tgrid_ovf,
                     unsigned long long* star_acc = nullptr);
// border table maintenance: re-insert list entries [0, n) after the host grew the table
void launch_border_rehash(hipStream_t s, const DevForestView& f, int n);

#ifdef SFFK_DEBUG_COUNTERS
void debug_counters(unsigned long long* out16);   // exact-kernel phase clocks (make EXTRA=-DSFFK_DEBUG_COUNTERS)
void debug_counters_query(unsigned long long* out16);
#endif
void debug_counters_prio(unsigned long long* out16);   // heap-kernel clocks (make EXTRA=-DSFFK_PRIO_DEBUG; zeros otherwise)
#ifdef SFFK_CI_TRACE
void debug_ci_trace(unsigned long long* out);     // make EXTRA=-DSFFK_CI_TRACE=<launch>: one launch of k_collide_items, per wave
#endif

}  // namespace sffk
