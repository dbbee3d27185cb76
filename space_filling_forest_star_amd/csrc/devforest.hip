// devforest.hip — the device-resident part of the SFF wave engine (gfx950).
//
// The reference's outer loop (SpaceForest::Solve, src/forest.h:122-202) and the accept / reject logic of
// expandNode (:240-376) are sequential.  The wave engine evaluates a whole round of samples speculatively
// (kernels.hip) and then has to COMMIT them in slot order.  Round 1 did that on the host (csrc/forest.cpp); here
// the commit runs on the GPU, so a round needs no host round trip at all and the host only reads a 256-byte
// status block once per wave:
//
//   k_wave_begin  one workgroup: frontier picks of every slot (uniform_int_distribution on the engine-word ring,
//                 exact incl. the rejection redraw: src/forest.h:136-151), then the first round's active list
//   k_commit      wide (64 samples per workgroup): the in-order commit of one round.  A sample whose neighbour walk
//                 reaches an EARLIER sample of the same round polls that sample's state; accepted samples get their
//                 node ids from the lower workgroups' published counts (slot order); border events are de-duplicated
//                 "first in slot order wins" through a stamped hash table; the round's last workgroup writes the
//                 control block.  Workgroups wait for lower ones only (see the kernel)
//   k_append_sample  wide: accepted samples -> node store, neighbour grid, frontier; the slots that were not accepted
//                 form the next round's active list AND draw that round's samples in the same launch
//   k_wave_end_wide  256 slots per workgroup: exhausted slots move their node to the closed list (first occurrence in slot
//                 order) and mark their frontier position; termination tests (src/forest.h:184-201)
//   k_frontier_compact  wide: order-preserving removal of the marked positions (the reference erases them one by
//                 one, :160-163) from one frontier buffer into the other
//
// Everything here is integer / index bookkeeping plus the few fp64 expressions of expandNode, evaluated in the same
// order as the host engine (-ffp-contract=off), so the forests are bit-identical.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "kernels.h"
#include "sff_geom.h"
#include "kernels_dev.h"

namespace sffk {

using namespace sffg;

__device__ __forceinline__ int record_words_dev(int nbcap) { return 6 + 4 * nbcap; }

// libstdc++ uniform_int_distribution<int>(0, range - 1) on one 64-bit engine word (Lemire's multiply-shift):
// returns the draw, or -1 when the word falls into the rejection zone (the reference then draws again)
__device__ __forceinline__ int lemire_pick(unsigned long long word, unsigned long long range) {
  const unsigned long long lo = word * range;
  const unsigned long long hi = __umul64hi(word, range);
  if (lo < range) {
    const unsigned long long thr = (0ULL - range) % range;
    if (lo < thr) return -1;
  }
  return (int)hi;
}

__device__ __forceinline__ int32_t* frontier_now(const DevForestView& f) { return f.ctrl->front_sel ? f.frontier2 : f.frontier; }
__device__ __forceinline__ int32_t* act_now(const DevForestView& f) { return f.ctrl->act_sel ? f.act_slot2 : f.act_slot; }

// Sizes the next round from the active list (already in place, act_cnt entries): src/forest.h:155 - i <
// ThresholdMisses && expandResult && iter < maxIterations.  Thread 0 of the single workgroup.
__device__ void round_begin_scalars(const DevForestView& f, DevCtrl* c) {
  const int cnt = c->act_cnt;
  int n = 0;
  if (c->round < f.threshold_misses && cnt > 0 && c->iter < f.max_iterations && !c->solved) {
    const int left = f.max_iterations - c->iter;
    n = cnt < left ? cnt : left;
  }
  c->n_act = n;
  if (n > 0) {
    // the arrays a round can grow: one node / frontier entry / border per sample at most
    if (c->n_nodes + n > f.node_cap - 8 || c->n_borders + n > f.border_cap) {
      c->fault = SFFK_FAULT_CAPACITY;
      c->halt = 1;
      c->n_act = 0;
    } else if ((unsigned long long)(c->n_borders + n) * 2ULL > f.bt_mask + 1ULL) {
      c->fault = SFFK_FAULT_BORDER_TABLE;
      c->halt = 1;
      c->n_act = 0;
    } else {
      c->round += 1;
      c->iter0 = c->iter;
      c->iter += n;
      c->N0 = c->n_nodes;
      c->words_base = c->cursor;
      c->cursor += (unsigned long long)f.words_per * (unsigned long long)n;
      c->rounds += 1;
      c->round_nodes += (unsigned long long)(c->n_nodes + n);
      c->round_queries += (unsigned long long)n;
    }
  }
}

// ------------------------------------------------------------------ wave begin
// Wide: one slot per thread; the workgroup that gets through last writes the control block (the picks themselves are
// independent of each other - only the astronomically rare rejection redraw needs the words one after the other).
#define WB_BLOCKS 64
__global__ __launch_bounds__(256) void k_wave_begin(DevForestView f) {
  __shared__ int s_last;
  __shared__ int s_scan[4];
  DevCtrl* c = f.ctrl;
  const unsigned long long tb0 = f.profile ? wall_clock64() : 0ULL;
  if (c->halt) {
    if (blockIdx.x == 0 && threadIdx.x == 0) { c->compact_from = 0; c->app_n = 0; }   // (no stale order for the wide follow-up kernels)
    return;
  }
  if (c->in_wave) {   // resuming inside a wave (after the host handled a fault): the active list is in place
    // (the slots' sorted positions are not: the query kernel walks the sample indices for the rest of this wave)
    if (blockIdx.x == 0 && threadIdx.x == 0) { c->compact_from = 0; c->app_n = 0; c->ord_valid = 0; round_begin_scalars(f, c); }
    return;
  }
  // priority-frontier mode: the slots take their nodes from the trees' heaps (k_prio_begin, launched behind this kernel)
  // unless every heap is empty
  if (f.prio.n_heaps && !c->empty_frontier) return;
  // node selection of every slot, src/forest.h:136-151 (non-priority mode): a uniform pick from the frozen frontier,
  // or from the closed list once the frontier has run empty
  const bool use_closed = c->closed_n > 0 && c->empty_frontier;
  const int pool = use_closed ? c->closed_n : c->frontier_n;
  int n_slots = f.wave < pool ? f.wave : pool;
  if (n_slots < 1) n_slots = 1;
  const int32_t* from = use_closed ? f.closed : frontier_now(f);
  int32_t* act = act_now(f);
  const unsigned long long cur = c->cursor;
  bool redraw = false;
  for (int sl = blockIdx.x * 256 + threadIdx.x; sl < n_slots; sl += gridDim.x * 256) {
    const unsigned long long wd = f.ring[(cur + (unsigned long long)sl) & f.ring_mask];
    const int pick = lemire_pick(wd, (unsigned long long)pool);
    if (pick < 0) redraw = true;
    else {
      const int node = from[pick];
      f.slot_node[sl] = node; f.slot_pos[sl] = pick;
      if (f.ord.hist) {   // the slot's bucket in the wave's spatial order: the coarse grid cell of its node (OrderView)
        const double* np = f.ord.pos + 6 * (size_t)node;   // (the grid's own cell: the store columns are these casts)
        const int cx = grid_coord((float)np[0], f.ord.ox, f.ord.inv_cell, f.ord.nx) >> f.ord.shift,
                  cy = grid_coord((float)np[1], f.ord.oy, f.ord.inv_cell, f.ord.ny) >> f.ord.shift,
                  cz = grid_coord((float)np[2], f.ord.oz, f.ord.inv_cell, f.ord.nz) >> f.ord.shift;
        const int key = (cz * f.ord.cny + cy) * f.ord.cnx + cx;
        f.ord.slot_key[sl] = key;
        f.ord.slot_rank[sl] = atomicAdd(&f.ord.hist[key], 1);
      }
    }
    act[sl] = sl;                 // every slot starts the wave failing
  }
  if (redraw) atomicOr(&f.commit_seq[2], 1);
  __threadfence();                // (this workgroup's slots are written before it counts itself through)
  __syncthreads();
  if (threadIdx.x == 0) s_last = atomicAdd(&f.commit_seq[1], 1) == (int)gridDim.x - 1;
  __syncthreads();
  if (!s_last) return;
  if (f.ord.hist) {   // bucket counts -> bucket starts (exclusive scan; the counters are zero again for the next wave)
    // only the buckets the grid's coarse cells use (1 575 of 4 096 on the bench job); wave-level scans, one trip through LDS
    constexpr int PERMAX = SFFK_ORD_BUCKETS / 256;
    const int nb = f.ord.n_buckets, per = (nb + 255) >> 8, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int v[PERMAX], sum = 0;
#pragma unroll
    for (int k = 0; k < PERMAX; ++k) {
      const int at = (int)threadIdx.x * per + k;
      v[k] = (k < per && at < nb) ? __hip_atomic_load(&f.ord.hist[at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
      sum += v[k];
    }
    int inc = sum;
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(inc, off);
      if (lane >= off) inc += o;
    }
    if (lane == 63) s_scan[wv] = inc;
    __syncthreads();
    int run = inc - sum;
    for (int w = 0; w < wv; ++w) run += s_scan[w];
#pragma unroll
    for (int k = 0; k < PERMAX; ++k) {
      const int at = (int)threadIdx.x * per + k;
      if (k < per && at < nb) {
        f.ord.start[at] = run;
        run += v[k];
        f.ord.hist[at] = 0;
      }
    }
  }
  if (threadIdx.x != 0) return;
  c->ord_valid = (f.ord.hist && !__hip_atomic_load(&f.commit_seq[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) ? 1 : 0;
  unsigned long long used = (unsigned long long)n_slots;
  if (__hip_atomic_load(&f.commit_seq[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
    // (probability ~ pool / 2^64 per pick) redo the picks one after another, words as they come
    unsigned long long at = cur;
    for (int s = 0; s < n_slots; ++s) {
      int pick;
      do { pick = lemire_pick(f.ring[at & f.ring_mask], (unsigned long long)pool); ++at; } while (pick < 0);
      f.slot_node[s] = from[pick];
      f.slot_pos[s] = pick;
    }
    used = at - cur;
    c->redraws += 1;
  }
  f.commit_seq[1] = 0;
  f.commit_seq[2] = 0;
  c->compact_from = 0;
  c->app_n = 0;
  c->cursor = cur + used;
  c->n_slots = n_slots;
  c->act_cnt = n_slots;
  c->use_closed = use_closed ? 1 : 0;
  c->round = 0;
  c->in_wave = 1;
  c->waves += 1;
  c->prio_wave = 0;
  c->prio_n0 = c->n_nodes;
  round_begin_scalars(f, c);
  if (f.profile) c->wprof[6] += wall_clock64() - tb0;
}

// ------------------------------------------------------------------ the commit of one round
// border de-duplication: open addressing on the key (n1 << 32 | n2 + 1); the value is a stamp (epoch << 32 | sample)
// that only ever decreases, so among the events of one round the smallest sample index owns the key and every
// entry of an earlier round (smaller epoch) beats them all
__device__ __forceinline__ size_t border_slot(const DevForestView& f, unsigned long long key) {
  size_t h = (size_t)((key * 0x9E3779B97F4A7C15ULL) >> 17) & (size_t)f.bt_mask;
  while (true) {
    const unsigned long long cur = f.bt_key[h];
    if (cur == key) return h;
    if (cur == 0ULL) {
      const unsigned long long old = atomicCAS(&f.bt_key[h], 0ULL, key);
      if (old == 0ULL || old == key) return h;
    }
    h = (h + 1) & (size_t)f.bt_mask;
  }
}


// ------------------------------------------------------------------ the commit of one round as ONE wide kernel
// k_commit = the accept / reject logic of expandNode (src/forest.h:246-300) for all samples of a round at once, with no
// single-workgroup step (rounds 1-2 had k_decide + a one-workgroup k_resolve): 64 samples per workgroup, 16 lanes per sample (lane 0 the
// parent edge, lane l the l-th neighbour).  Everything that needs the slot order only ever looks BACKWARDS in it:
//   - a sample whose walk (src/forest.h:262-300) reaches a sample of the same round waits for that sample's state
//     (ustate32, polled; the earlier sample sits in this or in a lower workgroup);
//   - node ids / border list positions are prefixes over the LOWER workgroups' published counts;
//   - a border key belongs to the first event in slot order: an event knows its fate once the lower workgroups (and its
//     own) have posted their stamps (atomicMin on the table entry) - later stamps can only be larger.
// So a workgroup waits for lower ones only, and workgroups start in index order: the lowest unfinished one never waits.
// The LAST workgroup of the round adds everything up and writes the control block.
// Published words are (launch sequence number << 32 | value): nothing is cleared between launches.
// (a fault raised in a walk has reached the L2 before its workgroup publishes a word: the round's last workgroup reads the
// flag as soon as it has every lower workgroup's count)
#define KC_FAULT_SETTLE() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define KC_SPIN_LIMIT (1 << 20)   // polls before a wait gives up (then: fault, the host redoes the round) - never reached
#define KC_ACC 0      // accepted samples of the workgroup
#define KC_PREF 1     // accepted samples of all lower workgroups
#define KC_WLO 2      // accepted word, low / high half
#define KC_WHI 3
#define KC_POSTED 4   // its border stamps are in the table
#define KC_OWN 5      // border events it owns
#define KC_CNT 6      // 6 counters of the walks + [12] dependent samples
#define KC_DEP 12
#define KC_EVN 13     // border events (before the first-in-slot-order rule)

__device__ __forceinline__ unsigned long long kc_load(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void kc_publish(unsigned long long* p, unsigned seq, unsigned v) {
  __hip_atomic_store(p, ((unsigned long long)seq << 32) | (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// waits for this launch's word; 0 + fault when it never comes
__device__ __forceinline__ unsigned kc_wait(const unsigned long long* p, unsigned seq, int32_t* fault) {
  for (int spin = 0; spin < KC_SPIN_LIMIT; ++spin) {
    const unsigned long long v = kc_load(p);
    if ((unsigned)(v >> 32) == seq) return (unsigned)v;
    __builtin_amdgcn_s_sleep(1);
  }
  atomicOr(fault, 1);
  return 0u;
}
// sum over the workgroup (every thread calls it; barriers inside)
__device__ __forceinline__ unsigned long long kc_block_sum(unsigned long long v, unsigned long long* slot) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  __syncthreads();
  if (threadIdx.x == 0) *slot = 0ULL;
  __syncthreads();
  if ((threadIdx.x & 63) == 0 && v) atomicAdd(slot, v);
  __syncthreads();
  return *slot;
}

// eight sums at once (entries below `from` are skipped)
__device__ __forceinline__ void kc_block_sum8(unsigned long long* v, unsigned long long* slots, int from) {
  for (int q = from; q < 8; ++q)
    for (int off = 32; off > 0; off >>= 1) v[q] += __shfl_xor(v[q], off);
  __syncthreads();
  if (threadIdx.x < 8) slots[threadIdx.x] = 0ULL;
  __syncthreads();
  if ((threadIdx.x & 63) == 0)
    for (int q = from; q < 8; ++q) if (v[q]) atomicAdd(&slots[q], v[q]);
  __syncthreads();
  for (int q = from; q < 8; ++q) v[q] = slots[q];
}

__global__ __launch_bounds__(1024) void k_commit(ResolveArgs A, int n_bound) {
  __shared__ int s_state[64], s_dk[64], s_dep[64];
  __shared__ unsigned long long s_cnt[64][6];
  __shared__ unsigned long long s_sum;
  __shared__ unsigned long long s_wa, s_we, s_c6[7];
  __shared__ int s_ev_total;
  __shared__ unsigned long long s_tot[8];
  __shared__ DevCtrl K;
  const DevForestView& f = A.f;
  DevCtrl* c = f.ctrl;
  const int b = blockIdx.x;
  const int grp = threadIdx.x >> 4, gl = threadIdx.x & 15, lane = threadIdx.x & 63;
  const int i = b * 64 + grp;
  // every answer the verdict may need is requested before anything is looked at (after a kernel boundary each dependent
  // step is a trip to the memory side of the chip); the buffers hold n_bound samples whatever the round's size is
  const bool inb = i < n_bound;
  const size_t s0 = (size_t)i * A.stride;
  const bool inl = inb && A.in_lim[i] != 0;
  const int flags = inb ? A.rec_flags[i] : 0;
  int nnb = inb ? A.rec_nnb[i] : 0;
  const bool pose_hit = inb && A.pose_hit[i] != 0;
  int fh = 0x7fffffff, ns = 0, nb = 0, meta = 0;
  if (inb) {
    fh = A.first_hit[s0 + gl]; ns = A.seg_ns[s0 + gl];
    if (gl > 0) { nb = A.rec_nb[(size_t)i * A.nbcap + gl - 1]; meta = A.rec_meta[(size_t)i * A.nbcap + gl - 1]; }
  }
  const int halt = c->halt, n = c->n_act;
  if (halt || n == 0) {
    if (b == 0 && threadIdx.x == 0) {
      c->app_n = 0;
      if (A.star) { A.S.hdr[0] = 0; A.S.hdr[1] = 1; A.S.hdr[2] = 0; A.S.hdr[4] = 0; }   // (no star stage unless a round commits)
    }
    return;
  }
  if (b == 0 && f.ord.hist)   // the sub-range lists the append behind this commit fills start empty (OrderView)
    for (int r = threadIdx.x; r < f.ord.n_sub; r += 1024) (c->act_sel ? f.ord.cnt[0] : f.ord.cnt[1])[r * SFFK_ORD_CNT_STRIDE] = 0;
  if (b * 64 >= n) return;
  const int nwg = (n + 63) >> 6;
  const bool last = b == nwg - 1;
  const unsigned seq = (unsigned)f.commit_seq[0] + 1u;
  const int seq30 = (int)(seq & 0x3fffffffu);
  const int Tb = f.temp_base, N0 = c->N0, nb0 = c->n_borders;
  const unsigned long long stamp_hi = (c->epoch + 1ULL) << 32;
  unsigned long long* pub = f.wg_pub + (size_t)b * SFFK_PUB_WORDS;
  unsigned long long tk[6];   // (phase clocks of the last workgroup; reading the clock is a scalar memory round trip)
  const bool clk = last && f.profile;
  tk[0] = clk ? wall_clock64() : 0ULL;
  unsigned long long* trace = (f.kc_trace && (int)c->rounds == f.kc_trace_round && threadIdx.x == 0) ? f.kc_trace + 8 * (size_t)b : nullptr;
#define KC_TRACE(k) do { if (trace) trace[k] = wall_clock64(); } while (0)
  KC_TRACE(0);
  int work_items = 0;
  if (last) {   // (the control block does not change before this workgroup rewrites it: requested now, used at the end)
    for (int w = threadIdx.x; w < (int)(sizeof(DevCtrl) / 4); w += 1024)
      reinterpret_cast<int32_t*>(&K)[w] = reinterpret_cast<const int32_t*>(c)[w];
    work_items = A.round_ctrl[2];
  }

  // ---- 1. every sample's walk, as far as the states of the earlier samples allow
  auto calls_of = [](int fh, int ns) -> unsigned long long {
    return fh != 0x7fffffff ? (unsigned long long)fh : (unsigned long long)ns;
  };
  const int gsh = lane & 48;                         // first lane of the group in its wavefront
  auto gballot = [&](bool p) -> unsigned { return (unsigned)((__ballot(p) >> gsh) & 0xffffULL); };
  auto gsum = [&](unsigned long long v) -> unsigned long long {
    for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
  };
  unsigned long long cnt[6] = {0, 0, 0, 0, 0, 0};   // cc, pf, nq, ex_pose, ex_seg, ex_smp
  int st = 1, code = SFFK_REJECTED, dk = 0, was_dep = 0;
  const bool live = i < n;
  bool pending = false;
  if (nnb > 15) nnb = 15;
  const bool have = gl <= nnb;
  if (!have) { fh = 0x7fffffff; ns = 0; nb = 0; meta = 0; }
  const bool is_nb = have && gl > 0;
  const bool mate = is_nb && nb >= Tb;
  const bool same = (meta & 1) != 0;
  const bool fr = fh == 0x7fffffff;
  const unsigned long long my_calls = calls_of(fh, ns);
  const unsigned stops = gballot(is_nb && (mate || !same || fr));
  const unsigned mates = gballot(mate), sames = gballot(same), frees = gballot(fr);
  // single-goal mode (src/forest.h:283-299 with hasGoal): a neighbour of another tree rejects the sample without a look at
  // its edge - unless it is the goal: a free edge to it ends the run IN THE MIDDLE of the round, which is the host
  // engine's business (the round is rolled back like a faulted one and replayed there)
  const bool goal_mode = f.goal_id >= 0;
  const unsigned goals = gballot(goal_mode && is_nb && nb == f.goal_id);
  bool count_end = true;                            // (the neighbour that ended the walk had its edge looked at)
  if (live) {
    if (!inl) code = SFFK_OUTSIDE;
    else if (flags & 2) { if (gl == 0) { atomicOr(A.fault_pending, 1); atomicAdd(A.fault_pending + 1, 1); KC_FAULT_SETTLE(); } }   // hit / neighbour list overflow: host path
    else if ((flags & 3) == 1) {
      const bool ovf = gballot(have && fh == 0) != 0;    // 0 = the edge's triangle candidate list ran over
      const bool mine = A.world <= 1 || i % A.world == A.rank;   // (executed work is counted by the rank that ran it)
      const unsigned long long smp = gsum(have ? (unsigned long long)ns : 0ULL);
      if (mine) { cnt[3] = 1; cnt[4] = 1 + (unsigned long long)nnb; cnt[5] = smp; }
      const int fh0 = __shfl(fh, gsh), ns0 = __shfl(ns, gsh);
      if (ovf) { if (gl == 0) { atomicOr(A.fault_pending, 1); atomicAdd(A.fault_pending + 2, 1); KC_FAULT_SETTLE(); } }
      else {
        cnt[0] = 1;                                // :246 env.Collide(newPoint)
        if (!pose_hit) {
          cnt[1] = 1;
          cnt[0] += calls_of(fh0, ns0);
          if (fh0 == 0x7fffffff) {                 // parent edge free: the neighbour loop decides
            cnt[2] = (unsigned long long)f.n_trees;   // :262-267 one radiusSearch per tree
            pending = true;
          }
        }
      }
    }
  }
  // the walk: from stop to stop (a stop = a round-mate, a neighbour of another tree, or one of the same tree with a
  // free edge; the neighbours in between are visited and change nothing).  A round-mate only exists if that sample
  // became a node: its state is polled - all unknown mates of a sample at once, so a step of the outer loop is one
  // level of the dependency chain.
  int ks = stops ? __ffs((int)stops) - 1 : 16;
  int end = 16;                                     // (group lane of the stop that ended the walk; 16 = none: accepted)
  unsigned long long m_pf = 0, m_cc = 0;            // round-mates visited as nodes
  int mate_state = mate ? 0 : -1;                   // this lane's round-mate: 0 unknown
  if (pending && stops && ((mates >> ks) & 1u)) was_dep = 1;
  bool published = false;
  int passes = 0;
  for (int spin = 0; ; ++spin) {
    if (pending && mate && mate_state == 0 && gl >= ks) {
      const int v = __hip_atomic_load(&f.ustate32[nb - Tb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((v >> 2) == seq30) mate_state = v & 3;
    }
    bool blocked = false;
    while (true) {
      const int kq = ks < 16 ? ks : 0;
      const int ms = __shfl(mate_state, gsh + kq);
      const unsigned long long ck = __shfl(my_calls, gsh + kq);
      if (pending && !blocked) {
        if (ks >= 16 && (flags & 4)) {   // the walk ran off the end of a CUT neighbour record: the host path has the whole list
          if (gl == 0) { atomicOr(A.fault_pending, 1); atomicAdd(A.fault_pending + 3, 1); KC_FAULT_SETTLE(); }
          st = 1; code = SFFK_REJECTED; end = 16; pending = false;
        } else if (ks >= 16) { st = 2; code = SFFK_ACCEPT; end = 16; pending = false; }
        else {
          const bool k_mate = (mates >> ks) & 1u, k_same = (sames >> ks) & 1u, k_fr = (frees >> ks) & 1u;
          const unsigned after = stops & ~((2u << ks) - 1u);
          const int nxt = after ? __ffs((int)after) - 1 : 16;
          if (k_mate && ms == 0) blocked = true;                       // not known yet
          else if (k_mate && ms != 2) ks = nxt;                        // that sample never became a node
          else if (goal_mode && !k_same) {
            const bool is_goal = (goals >> ks) & 1u;
            if (is_goal && k_fr && gl == 0) { atomicOr(A.fault_pending, 1); KC_FAULT_SETTLE(); }   // :286-287 goal reached
            count_end = is_goal;
            st = 1; code = SFFK_REJECTED; end = ks; pending = false;         // :296-299
          } else {
            if (k_mate) { m_pf += 1; m_cc += ck; }                     // (an ordinary neighbour now)
            if (k_same) {
              if (k_fr) { st = 1; code = SFFK_REJECTED; end = ks; pending = false; }   // :276-280 overcrowded
              else ks = nxt;
            } else {                                                   // :288-299
              st = k_fr ? 3 : 1; code = k_fr ? SFFK_REJECT_EVENT : SFFK_REJECTED;
              dk = ks - 1; end = ks; pending = false;
            }
          }
        }
      }
      if (!__any(pending && !blocked)) break;
    }
    if (live && !pending && !published) {
      published = true;
      if (gl == 0) __hip_atomic_store(&f.ustate32[i], (seq30 << 2) | st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!__any(pending)) break;
    ++passes;
    if (spin >= KC_SPIN_LIMIT) {                    // (never: the states it waits for are earlier samples')
      if (gl == 0 && pending) { atomicOr(A.fault_pending, 1); KC_FAULT_SETTLE(); }
      pending = false; st = 1; code = SFFK_REJECTED;
      if (live && !published && gl == 0) __hip_atomic_store(&f.ustate32[i], (seq30 << 2) | 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      published = true;
      break;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  {
    const bool walked = cnt[2] != 0;
    const bool vis = walked && is_nb && !mate && (gl < end || (gl == end && count_end));
    const unsigned long long v_pf = gsum(vis ? 1ULL : 0ULL), v_cc = gsum(vis ? my_calls : 0ULL);
    if (walked) { cnt[1] += v_pf + m_pf; cnt[0] += v_cc + m_cc; }
  }
  if (gl == 0) {
    if (live) A.code[i] = (uint8_t)code;
    s_state[grp] = live ? st : 0;
    s_dk[grp] = dk;
    s_dep[grp] = live ? was_dep : 0;
    for (int q = 0; q < 6; ++q) s_cnt[grp][q] = cnt[q];
  }
  __syncthreads();
  if (clk) tk[1] = wall_clock64();
  KC_TRACE(1);
  // ---- 2. the workgroup's words, counts and counters
  if (threadIdx.x < 64) {
    const int v = s_state[threadIdx.x];
    const unsigned long long wa = __ballot(v == 2), we = __ballot(v == 3), wd = __ballot(s_dep[threadIdx.x] != 0);
    unsigned long long c6[6];
    for (int q = 0; q < 6; ++q) {
      c6[q] = s_cnt[threadIdx.x][q];
      for (int off = 32; off > 0; off >>= 1) c6[q] += __shfl_xor(c6[q], off);
    }
    if (threadIdx.x == 0) {
      s_wa = wa; s_we = we;
      for (int q = 0; q < 6; ++q) s_c6[q] = c6[q];
      s_c6[6] = (unsigned long long)__popcll(wd);
      f.w_acc[b] = wa;                              // (k_append, the star stage)
      // (the counters first, the accepted count last: the words share one 128-byte line, so whoever has seen the count
      // finds the counters in place - the round's last workgroup used to ask for them early and mostly had to ask again)
      for (int q = 0; q < 6; ++q) kc_publish(pub + KC_CNT + q, seq, (unsigned)c6[q]);
      kc_publish(pub + KC_DEP, seq, (unsigned)__popcll(wd));
      kc_publish(pub + KC_EVN, seq, (unsigned)__popcll(we));
      kc_publish(pub + KC_WLO, seq, (unsigned)(wa & 0xffffffffULL));
      kc_publish(pub + KC_WHI, seq, (unsigned)(wa >> 32));
      kc_publish(pub + KC_ACC, seq, (unsigned)__popcll(wa));
      KC_TRACE(2);
      f.w_ev[b] = we;                               // (k_border_finalize)
    }
  }
  __syncthreads();
  const unsigned long long wa = s_wa, we = s_we;
  // plain SFF: which event owns its key (the first in slot order) and where the border list gets it is settled by the
  // append launch that follows (k_border_finalize) - by then every stamp is in the table, and the two rounds of "wait
  // for every lower workgroup" (stamps posted, owned counts) leave this kernel's critical path.  SFF* needs the entries
  // in its passes: there they are settled here.
  const bool defer = !A.star;
  // ---- 2b. border events (a free edge to a neighbour of another tree, :288-294) whose neighbour is a STORE node need
  // nothing of the other workgroups: their stamps go out now, and a workgroup none of whose events waits for a
  // round-mate's id says "posted" before it collects the lower workgroups' counts - the higher workgroups then find the
  // word in place when they get to their own events (it used to be published behind two waits)
  int e_nb = 0, e_ex = 0, e_i = 0;
  size_t e_h = 0;
  bool is_ev = false, stamped = false, posted = false;
  if (threadIdx.x < 64) {
    if ((we >> threadIdx.x) & 1ULL) {
      is_ev = true;
      e_i = b * 64 + (int)threadIdx.x;
      const int raw = A.rec_nb[(size_t)e_i * A.nbcap + s_dk[threadIdx.x]];
      e_ex = A.parent[e_i];
      if (raw < Tb) {
        e_nb = raw;
        const int a = e_nb < e_ex ? e_nb : e_ex, bb = e_nb < e_ex ? e_ex : e_nb;
        const unsigned long long key = ((unsigned long long)(uint32_t)a << 32) | ((unsigned long long)(uint32_t)bb + 1ULL);
        e_h = border_slot(f, key);
        atomicMin(&f.bt_val[e_h], stamp_hi | (unsigned long long)(uint32_t)e_i);
        stamped = true;
        if (defer) { f.ev_h[e_i] = (unsigned long long)e_h; f.ev_nb[e_i] = e_nb; f.ev_raw[e_i] = raw; }
      }
    }
    if (!defer && __ballot(is_ev && !stamped) == 0ULL) {
      if (we != 0ULL) __threadfence();         // the stamps are in the table before the word says so
      if (threadIdx.x == 0) {
        kc_publish(pub + KC_POSTED, seq, 1u);
        if (we == 0ULL) kc_publish(pub + KC_OWN, seq, 0u);
      }
      posted = true;
    }
  }
  // ---- 3. node ids: N0 + accepted samples before, in slot order
  unsigned long long part = 0;
  if ((int)threadIdx.x < b) part = kc_wait(f.wg_pub + (size_t)threadIdx.x * SFFK_PUB_WORDS + KC_ACC, seq, f.commit_seq + 6);
  // (the last workgroup: the lower workgroups' counters were published before their counts - requested now, looked at
  // when the control block is written)
  unsigned long long early[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (last && (int)threadIdx.x < b)   // (behind the count: published before it)
    for (int q = 0; q < 8; ++q) early[q] = kc_load(f.wg_pub + (size_t)threadIdx.x * SFFK_PUB_WORDS + KC_CNT + q);
  const int acc_pref = (int)kc_block_sum(part, &s_sum);
  // The round's fault flag (a bounded list that overflowed, the goal reached: raised in a workgroup's WALKS, before it
  // publishes anything) is final once every lower workgroup's count is in - the block sum's barriers are behind all the
  // waits, and behind this workgroup's own walks: asked for now, looked at when the control block is written (it used to
  // be asked for there: a round trip on the round's critical path).  A wait that gives up (never: KC_SPIN_LIMIT) raises
  // commit_seq[6] instead, which ends the run at the wave's end (SFFK_FAULT_INTERNAL).
  int fault_seen = 0;
  if (last && threadIdx.x == 1023) fault_seen = __hip_atomic_load(A.fault_pending, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (threadIdx.x == 0) {
    f.acc_pref[b] = acc_pref;
    kc_publish(pub + KC_PREF, seq, (unsigned)acc_pref);
  }
  if (A.star && threadIdx.x < 64 && ((wa >> threadIdx.x) & 1ULL))   // SFF*: the accepted samples as a list (rank -> sample)
    A.S.acc_sample[acc_pref + __popcll(wa & ((1ULL << threadIdx.x) - 1ULL))] = b * 64 + (int)threadIdx.x;
  if (clk) tk[2] = wall_clock64();
  KC_TRACE(4);
  // ---- 4. border events (a free edge to a neighbour of another tree, :288-294), first in slot order wins
  int n_own = 0, ev_pref = 0;
  if (we != 0ULL || last) {
    if (is_ev && !stamped) {                 // (a round-mate neighbour was accepted: its new id)
      const int raw = A.rec_nb[(size_t)e_i * A.nbcap + s_dk[threadIdx.x]];
      const int j = raw - Tb, bj = j >> 6;
      int pj = acc_pref;
      unsigned long long wj = wa;
      if (bj != b) {
        const unsigned long long* pp = f.wg_pub + (size_t)bj * SFFK_PUB_WORDS;
        pj = (int)kc_wait(pp + KC_PREF, seq, f.commit_seq + 6);
        wj = (unsigned long long)kc_wait(pp + KC_WLO, seq, f.commit_seq + 6) |
             ((unsigned long long)kc_wait(pp + KC_WHI, seq, f.commit_seq + 6) << 32);
      }
      e_nb = N0 + pj + __popcll(wj & ((1ULL << (j & 63)) - 1ULL));
      const int a = e_nb < e_ex ? e_nb : e_ex, bb = e_nb < e_ex ? e_ex : e_nb;
      const unsigned long long key = ((unsigned long long)(uint32_t)a << 32) | ((unsigned long long)(uint32_t)bb + 1ULL);
      e_h = border_slot(f, key);
      atomicMin(&f.bt_val[e_h], stamp_hi | (unsigned long long)(uint32_t)e_i);
      if (defer) { f.ev_h[e_i] = (unsigned long long)e_h; f.ev_nb[e_i] = e_nb; f.ev_raw[e_i] = raw; }
    }
  }
  if (!defer && (we != 0ULL || last)) {
    if (threadIdx.x < 64 && !posted) {
      if (we != 0ULL) __threadfence();         // the stamps are in the table before the word says so
      if (threadIdx.x == 0) kc_publish(pub + KC_POSTED, seq, 1u);
    }
    KC_TRACE(3);
    // (an event's fate needs every lower workgroup's stamps; the last workgroup without events of its own is only here
    // for the totals)
    if (we != 0ULL && (int)threadIdx.x < b) (void)kc_wait(f.wg_pub + (size_t)threadIdx.x * SFFK_PUB_WORDS + KC_POSTED, seq, f.commit_seq + 6);
    __syncthreads();
    KC_TRACE(5);
    bool own = false;
    if (is_ev) {
      // (atomic read: the stamps were written by L2 atomics a moment ago)
      const unsigned long long owner = __hip_atomic_load(&f.bt_val[e_h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      own = owner == (stamp_hi | (unsigned long long)(uint32_t)e_i);
    }
    unsigned long long wo = 0ULL;
    if (threadIdx.x < 64) {
      wo = __ballot(own);
      if (threadIdx.x == 0) { s_ev_total = __popcll(wo); kc_publish(pub + KC_OWN, seq, (unsigned)__popcll(wo)); }
    }
    __syncthreads();
    n_own = s_ev_total;
    if (n_own > 0 || last) {
      unsigned long long pe = 0;
      if ((int)threadIdx.x < b) pe = kc_wait(f.wg_pub + (size_t)threadIdx.x * SFFK_PUB_WORDS + KC_OWN, seq, f.commit_seq + 6);
      ev_pref = (int)kc_block_sum(pe, &s_sum);
    }
    KC_TRACE(6);
    if (own) {
      const int at = nb0 + ev_pref + __popcll(wo & ((1ULL << threadIdx.x) - 1ULL));
      // d = costs to the roots + the edge (:291).  A neighbour accepted in this very round is not in the store yet
      // (k_append runs next): its tree is its parent's, its position the sample's, its cost the one k_append will write.
      double pn[6], pe6[6], dn;
      int ta;
      const int raw = A.rec_nb[(size_t)e_i * A.nbcap + s_dk[threadIdx.x]];
      if (raw >= Tb) {
        const int j = raw - Tb;
        for (int q = 0; q < 6; ++q) pn[q] = A.newpos[6 * (size_t)j + q];
        dn = A.pdist[j] + f.d_root[A.parent[j]];
        ta = A.st.tree[A.parent[j]];
      } else {
        for (int q = 0; q < 6; ++q) pn[q] = A.st.pos[6 * (size_t)e_nb + q];
        dn = f.d_root[e_nb];
        ta = A.st.tree[e_nb];
      }
      const int tb = A.st.tree[e_ex];
      f.b_n1[at] = e_nb < e_ex ? e_nb : e_ex; f.b_n2[at] = e_nb < e_ex ? e_ex : e_nb;
      f.b_ta[at] = ta < tb ? ta : tb; f.b_tb[at] = ta < tb ? tb : ta;
      for (int q = 0; q < 6; ++q) pe6[q] = A.st.pos[6 * (size_t)e_ex + q];
      if (A.star) {
        // SFF*: the two costs are the ones the sample's turn finds (earlier samples of the round may have rewired either
        // node): k_star_pass adds them to the distance
        const int e = at - nb0;
        A.S.ev_sample[e] = e_i; A.S.ev_nb[e] = e_nb; A.S.ev_ex[e] = e_ex; A.S.ev_dist[e] = dist6(pn, pe6);
        f.b_dist[at] = 0.0;
      } else
      f.b_dist[at] = dn + f.d_root[e_ex] + dist6(pn, pe6);
      f.pair[(size_t)ta * f.n_trees + tb] = 1;
      f.pair[(size_t)tb * f.n_trees + ta] = 1;
    }
  }
  KC_TRACE(7);
  if (!last) return;
  // ---- 5. the last workgroup of the round: totals, the control block (the next round's active list = the slots of this
  // round that were not accepted, then the slots the iteration cap kept out of it: k_append writes it)
  if (clk) tk[3] = wall_clock64();
  // the query kernel's clock bracket: every workgroup of it reported into one of 64 shards (DevForestView::qclk_sh)
  unsigned long long q_end = 0ULL, q_beg = ~0ULL;
  if (f.qclk_sh && threadIdx.x < 64) {
    q_end = f.qclk_sh[threadIdx.x * 16]; q_beg = f.qclk_sh[threadIdx.x * 16 + 1];
    for (int off = 32; off > 0; off >>= 1) {
      const unsigned long long e2 = __shfl_xor(q_end, off), b2 = __shfl_xor(q_beg, off);
      q_end = e2 > q_end ? e2 : q_end; q_beg = b2 < q_beg ? b2 : q_beg;
    }
  }
  unsigned long long tot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if ((int)threadIdx.x < b) {   // (every lower workgroup has published its counters long ago: one batch of loads)
    const unsigned long long* pp = f.wg_pub + (size_t)threadIdx.x * SFFK_PUB_WORDS;
    for (int q = 0; q < 8; ++q) tot[q] = (unsigned)(early[q] >> 32) == seq ? (unsigned)early[q] : kc_wait(pp + KC_CNT + q, seq, f.commit_seq + 6);
  }
  // (a fault is raised before its workgroup publishes anything: with every lower workgroup's words in, the flag is final)
  if (threadIdx.x == 1023) s_ev_total = fault_seen;
  kc_block_sum8(tot, s_tot, 0);
  for (int q = 0; q < 7; ++q) tot[q] += s_c6[q];
  tot[7] += (unsigned long long)__popcll(we);
  // (deferred: every event counted - an upper bound until the append launch has entered the ones that own their key)
  const int n_acc = acc_pref + __popcll(wa), n_ev = defer ? (int)tot[7] : ev_pref + n_own, n_dep = (int)tot[6];
  const bool faulted = s_ev_total != 0;
  auto roll_back = [&](DevCtrl* k) {
    // a bounded device list overflowed somewhere in this round: nothing is committed, the bookkeeping of the round's
    // begin is rolled back and the host redoes the round on its unbounded path
    k->fault = SFFK_FAULT_LISTS;
    k->halt = 1;
    k->round -= 1;
    k->iter = k->iter0;
    k->cursor = k->words_base;
    k->rounds -= 1;
    k->round_nodes -= (unsigned long long)(k->N0 + n);
    k->round_queries -= (unsigned long long)n;
    k->n_act = 0;
    k->app_n = 0;
  };
  if (A.star && !faulted) {
    // SFF*: the star stage may still fault after this kernel has committed the round's bookkeeping (a member edge's
    // candidate list, the fixed point's launch budget): the control block as a rolled-back round leaves it, for k_star_apply
    for (int w = threadIdx.x; w < (int)(sizeof(DevCtrl) / 4); w += 1024)
      reinterpret_cast<int32_t*>(A.S.backup)[w] = reinterpret_cast<const int32_t*>(&K)[w];
    __syncthreads();
    if (threadIdx.x == 0) roll_back(A.S.backup);
  }
  if (threadIdx.x == 0) {
    if (faulted) {
      roll_back(&K);
      *A.fault_pending = 0;
      if (A.star) { A.S.hdr[0] = 0; A.S.hdr[1] = 1; A.S.hdr[2] = 0; A.S.hdr[4] = 0; }
    } else {
      const int fn0 = K.frontier_n, act_cnt = K.act_cnt, act_sel = K.act_sel;
      K.collide_calls += tot[0];
      K.path_free_calls += tot[1];
      K.nn_queries += tot[2];
      K.poses_executed += tot[3];
      K.segments_executed += tot[4];
      K.samples_executed += tot[5];
      K.work_items += (unsigned long long)work_items;
      if (f.qclk_sh) { K.q_t0 = q_beg; K.q_t1 = q_end; }
      if (K.q_t1 > K.q_t0) { K.q_ticks += K.q_t1 - K.q_t0; K.q_launches += 1ULL; }
      K.app_n = n;                  // k_append applies this commit
      K.app_N0 = N0;
      K.app_fn0 = fn0;
      K.app_act_sel = act_sel;
      K.app_act_cnt = act_cnt;
      K.iter0_app = K.iter0;
      K.n_nodes = N0 + n_acc;
      K.frontier_n = f.prio.n_heaps ? fn0 : fn0 + n_acc;
      K.n_borders = nb0 + n_ev;
      K.app_nb0 = nb0;
      K.n_unsettled += n_dep;
      K.epoch += 1ULL;
      if (A.star) {
        K.nn_queries += (unsigned long long)n_acc;       // one knnSearch per accepted sample (src/forest.h:317)
        A.S.hdr[0] = n_acc; A.S.hdr[1] = 0; A.S.hdr[2] = n_ev; A.S.hdr[3] = nb0; A.S.hdr[4] = 0;
      }
      K.act_sel = act_sel ^ 1;
      K.act_cnt = (n - n_acc) + (act_cnt - n);
      round_begin_scalars(f, &K);
      if (clk) {
        tk[4] = wall_clock64();
        for (int q = 0; q < 4; ++q) K.prof[q] += tk[q + 1] - tk[q];
      }
      K.prof[5] += (unsigned long long)passes;
      K.prof[6] += 1ULL;
      if ((unsigned long long)passes > K.prof[7]) K.prof[7] = (unsigned long long)passes;
    }
    f.commit_seq[0] = (int32_t)seq;
  }
  __syncthreads();
  for (int w = threadIdx.x; w < (int)(sizeof(DevCtrl) / 4); w += 1024)
    reinterpret_cast<int32_t*>(c)[w] = reinterpret_cast<const int32_t*>(&K)[w];
}

// k_append (wide): the accepted samples become nodes - store columns, node records, neighbour grid, frontier
// (src/forest.h:353-367).  Runs although the NEXT round may already be halted: this commit is final.
// One slot's share of the append.  Returns the slot's index in the NEXT round's active list (-1: none - accepted, or
// nothing was committed) and the slot itself.
// part: 0 = everything, 1 = only the accepted sample's node, 2 = only the next round's list (+ the claims of a wave that
// is over) - k_append_sample runs the two parts in different workgroups.
__device__ __forceinline__ void append_ord_store(const ResolveArgs& A, int ord_at, int next) {
  if (ord_at >= 0) (A.f.ctrl->app_act_sel ? A.f.ord.lst[0] : A.f.ord.lst[1])[ord_at] = next;
}
__device__ __forceinline__ int append_one(const ResolveArgs& A, int i, int& slot_out, int& ord_at, int part = 0) {
  ord_at = -1;
  const DevForestView& f = A.f;
  const DevCtrl* c = f.ctrl;
  const int n = c->app_n;
  slot_out = -1;
  if (n <= 0) return -1;
  const int act_cnt = c->app_act_cnt;
  const int32_t* act_old = c->app_act_sel ? f.act_slot2 : f.act_slot;
  int32_t* act_new = c->app_act_sel ? f.act_slot : f.act_slot2;
  // the wave is over with this commit (no next round): the slots still on the list are exhausted - each claims its node
  // (atomicMin of its place in the list: a node held by several slots moves to the closed list once, at its first slot,
  // src/forest.h:160-178) and parks it for k_wave_end_wide, which would otherwise post the claims itself and meet at a counter first
  const bool wave_over = c->n_act == 0 && !c->halt && c->in_wave && !c->use_closed;
  const bool ord_on = f.ord.hist && c->ord_valid;
  auto claim = [&](int e, int slot) {
    const int nd = f.slot_node[slot];
    atomicMin(&f.claim[nd], e);
    f.ulist[e] = nd;
  };
  if (wave_over && i == 0 && part != 1) f.ctrl->claims_done = 1;
  if (i >= n) {
    // the slots the iteration cap kept out of the committed round stay on the list, behind the still-failing ones
    // (the cap has been reached: they never draw again)
    if (i < act_cnt && part != 1) {
      const int e = (c->act_cnt - (act_cnt - n)) + (i - n), slot = act_old[i];
      act_new[e] = slot;
      if (wave_over) claim(e, slot);
    }
    return -1;
  }
  const unsigned long long w = f.w_acc[i >> 6];
  const int rank = f.acc_pref[i >> 6] + __popcll(w & ((1ULL << (i & 63)) - 1ULL));
  if (!((w >> (i & 63)) & 1ULL)) {
    // its sample's index in the next round joins the list of its sub-range (OrderView): a returning atomic behind two
    // dependent loads - in k_append_sample the NODE-creating half of the workgroups does it (for a sample that was not
    // accepted it has nothing else to do), so that the sampling half's chain stays as short as it was
    auto ord_append = [&](int slot) {
      const int sub = f.ord.slot_pos[slot] >> 6;
      const bool first = c->app_act_sel != 0;   // (the buffer that is NOT the committed round's)
      ord_at = sub * 64 + atomicAdd(&(first ? f.ord.cnt[0] : f.ord.cnt[1])[sub * SFFK_ORD_CNT_STRIDE], 1);
    };
    if (part == 1) {
      if (ord_on) { ord_append(act_old[i]); append_ord_store(A, ord_at, i - rank); }
      return -1;
    }
    slot_out = act_old[i];
    act_new[i - rank] = slot_out;       // not accepted: the slot tries again (rank = accepted samples before it)
    if (ord_on && part == 0) ord_append(slot_out);   // (k_append: everything in one thread; the caller stores the entry)
    if (wave_over) claim(i - rank, slot_out);
    return i - rank;
  }
  if (A.star || part == 2) return -1;   // SFF*: k_star_apply has made the accepted samples nodes
  const int N0 = c->app_N0, fn0 = c->app_fn0;
  int32_t* const frontier = frontier_now(f);
  const int id = N0 + rank;
  const int ex = A.parent[i];
  const double* p = A.newpos + 6 * (size_t)i;
  const size_t o = (size_t)id;
  GridItem it;
  for (int k = 0; k < 6; ++k) it.p[k] = p[k];
  it.id = id;
  it.tree = A.st.tree[ex];
  it.pad[0] = it.pad[1] = 0;
  A.st.x[o] = (float)p[0]; A.st.y[o] = (float)p[1]; A.st.z[o] = (float)p[2];
  A.st.yaw[o] = (float)p[3]; A.st.pitch[o] = (float)p[4]; A.st.roll[o] = (float)p[5];
  for (int k = 0; k < 6; ++k) A.st.pos[6 * o + k] = p[k];
  A.st.tree[o] = it.tree;
  const double pd = A.pdist[i];
  f.parent[o] = ex;                                  // src/forest.h:353
  f.d_closest[o] = pd;
  f.d_root[o] = pd + f.d_root[ex];
  f.iter[o] = (uint32_t)(c->iter0_app + i + 1);
  f.nflag[o] = f.prio.n_heaps ? 0 : 2;
  if (!f.prio.n_heaps) frontier[fn0 + rank] = id;    // :365 (priority mode: the tree's heaps, at the wave's end - k_prio_end)
  grid_put(A.g, it);                                 // flannIndex->addPoints, :367
  return -1;
}
// Plain SFF: the committed round's border events (src/forest.h:288-294) are entered here, one workgroup per 64 samples
// beside the append's own workgroups: an event owns its key when its stamp is the table entry's minimum (every stamp of
// the round is in the table - k_commit has ended), its place in the list is border count before the round + the owned
// events of the LOWER workgroups (published like k_commit's words, polled; a workgroup waits for lower ones only) + its
// rank in the workgroup.  The round's last workgroup writes the exact border count over k_commit's upper bound.
__device__ void border_finalize(const ResolveArgs& A, int b) {
  __shared__ unsigned long long s_bsum;
  __shared__ int s_bown;
  const DevForestView& f = A.f;
  DevCtrl* c = f.ctrl;
  const int n = c->app_n;
  if (n <= 0 || A.star) return;
  const int nwg = (n + 63) >> 6;
  if (b >= nwg) return;
  const unsigned seq = (unsigned)f.commit_seq[0];                 // (the commit's launch number)
  const unsigned long long stamp_hi = c->epoch << 32;             // (the commit's epoch: incremented when it committed)
  const int Tb = f.temp_base, nb0 = c->app_nb0;
  const unsigned long long we = f.w_ev[b];
  unsigned long long* pub = f.wg_pub + (size_t)b * SFFK_PUB_WORDS;
  const int tid = threadIdx.x;
  bool own = false;
  int e_i = 0;
  unsigned long long wo = 0ULL;
  if (tid < 64) {
    if ((we >> tid) & 1ULL) {
      e_i = b * 64 + tid;
      // (atomic read: the stamps were written by L2 atomics)
      const unsigned long long owner = __hip_atomic_load(&f.bt_val[f.ev_h[e_i]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      own = owner == (stamp_hi | (unsigned long long)(uint32_t)e_i);
    }
    wo = __ballot(own);
    if (tid == 0) { s_bown = __popcll(wo); kc_publish(pub + KC_OWN, seq, (unsigned)__popcll(wo)); }
  }
  __syncthreads();
  const int n_own = s_bown;
  const bool last = b == nwg - 1;
  if (n_own == 0 && !last) return;
  unsigned long long pe = 0;
  for (int t = tid; t < b; t += 256) pe += kc_wait(f.wg_pub + (size_t)t * SFFK_PUB_WORDS + KC_OWN, seq, f.commit_seq + 6);
  for (int off = 32; off > 0; off >>= 1) pe += __shfl_xor(pe, off);
  if (tid == 0) s_bsum = 0ULL;
  __syncthreads();
  if ((tid & 63) == 0 && pe) atomicAdd(&s_bsum, pe);
  __syncthreads();
  const int ev_pref = (int)s_bsum;
  if (own) {
    const int at = nb0 + ev_pref + __popcll(wo & ((1ULL << tid) - 1ULL));
    // d = costs to the roots + the edge (:291).  A neighbour accepted in this very round: its tree is its parent's, its
    // position the sample's, its cost the one this launch's append writes (other workgroups: computed, not read).
    double pn[6], pe6[6], dn;
    int ta;
    const int raw = f.ev_raw[e_i], e_nb = f.ev_nb[e_i], e_ex = A.parent[e_i];
    if (raw >= Tb) {
      const int j = raw - Tb;
      for (int q = 0; q < 6; ++q) pn[q] = A.newpos[6 * (size_t)j + q];
      dn = A.pdist[j] + f.d_root[A.parent[j]];
      ta = A.st.tree[A.parent[j]];
    } else {
      for (int q = 0; q < 6; ++q) pn[q] = A.st.pos[6 * (size_t)e_nb + q];
      dn = f.d_root[e_nb];
      ta = A.st.tree[e_nb];
    }
    const int tb = A.st.tree[e_ex];
    f.b_n1[at] = e_nb < e_ex ? e_nb : e_ex; f.b_n2[at] = e_nb < e_ex ? e_ex : e_nb;
    f.b_ta[at] = ta < tb ? ta : tb; f.b_tb[at] = ta < tb ? tb : ta;
    for (int q = 0; q < 6; ++q) pe6[q] = A.st.pos[6 * (size_t)e_ex + q];
    f.b_dist[at] = dn + f.d_root[e_ex] + dist6(pn, pe6);
    f.pair[(size_t)ta * f.n_trees + tb] = 1;
    f.pair[(size_t)tb * f.n_trees + ta] = 1;
  }
  if (last && tid == 0) c->n_borders = nb0 + ev_pref + n_own;
}
__global__ __launch_bounds__(256) void k_append(ResolveArgs A, int append_blocks) {
  if ((int)blockIdx.x >= append_blocks) { border_finalize(A, (int)blockIdx.x - append_blocks); return; }
  int slot, ord_at;
  const int next = append_one(A, blockIdx.x * 256 + threadIdx.x, slot, ord_at);
  append_ord_store(A, ord_at, next);
}
// k_append + the NEXT round's k_sample_steer in one launch: a slot that was not accepted draws its next sample right
// away (its place in the next round's list is its old place minus the accepted samples before it; the round's size,
// word base and epoch are in the control block since k_commit).  The sample reads what the append of this very launch
// writes nowhere: its centre is a node of an earlier wave, its temporaries lie behind the node arrays.  The append's
// inputs (newpos, pdist, parent of the committed round) are the sampling's outputs: the rounds of a wave alternate
// between two sets of these arrays (A = the committed round's, P = the next round's).
// The first half of the workgroups creates the accepted samples' nodes, the second half writes the list and samples: a
// wavefront that had to do both would run the two chains of dependent loads one after the other.
__global__ __launch_bounds__(256) void k_append_sample(ResolveArgs A, SampleLaunch P, int append_blocks) {
  if ((int)blockIdx.x >= 2 * append_blocks) { border_finalize(A, (int)blockIdx.x - 2 * append_blocks); return; }
  const int nb = append_blocks;
  const bool sampler = (int)blockIdx.x >= nb;
  const int tid = ((int)blockIdx.x - (sampler ? nb : 0)) * 256 + threadIdx.x;
  int slot, ord_at;
  if (!sampler) { (void)append_one(A, tid, slot, ord_at, 1); return; }
  const int next = append_one(A, tid, slot, ord_at, 2);
  sample_steer_one(tid, next, slot < 0 ? -2 : slot, P);   // (-1 is k_sample_steer's "look the slot up")
  append_ord_store(A, ord_at, next);
}

// the control block at the end of a wave: closed list / frontier sizes, termination (src/forest.h:184-201); one thread
__device__ void wave_end_control(const DevForestView& f, DevCtrl* c, int removed, int fn, bool from_closed, int n_fail,
                                 const int32_t* grid_ovf, const int32_t* tgrid_ovf, const unsigned long long* star_s) {
  c->closed_n += removed;
  if (f.prio.n_heaps) {                       // priority mode: no frontier list; "empty" = every heap is (k_prio_end)
    c->compact_from = 0;
    c->empty_frontier = c->prio_all_empty;
  } else {
    c->compact_from = removed > 0 ? fn : 0;   // k_frontier_compact: entries of the old buffer to sift
    c->frontier_n = fn - removed;
    if (removed > 0) c->front_sel ^= 1;
    c->empty_frontier = c->frontier_n == 0 ? 1 : 0;
  }
  if (!c->solved && c->empty_frontier && f.goal_id < 0) {   // (with a goal only reaching it solves, :204-206)
    // maxConnected() == numRoots (:379-418): every tree reachable from tree 0 over pairs that hold a border
    const int R = f.n_trees;
    int reached = 1;
    // (claim[] is free again: use its first R ints as the visited marks, restored afterwards)
    for (int t = 0; t < R; ++t) f.claim[t] = t == 0 ? 1 : 0;
    bool grew = true;
    while (grew) {
      grew = false;
      for (int a = 0; a < R; ++a) {
        if (f.claim[a] != 1) continue;
        f.claim[a] = 2;
        for (int b = 0; b < R; ++b)
          if (f.claim[b] == 0 && f.pair[(size_t)a * R + b]) { f.claim[b] = 1; ++reached; grew = true; }
      }
    }
    for (int t = 0; t < R; ++t) f.claim[t] = 0x7fffffff;
    c->solved = reached == R ? 1 : 0;
  }
  c->claims_done = 0;
  c->clear_n = from_closed ? 0 : n_fail;
  const bool budget = f.node_budget > 0 && c->n_nodes >= f.node_budget;
  c->terminated = (c->solved || c->iter >= f.max_iterations || budget) ? 1 : 0;
  c->halt = c->terminated;
  if (f.commit_seq[6]) { c->fault = SFFK_FAULT_INTERNAL; c->halt = 1; }   // (a workgroup gave up waiting for a lower one: never)
  c->in_wave = 0;
  c->n_act = 0;
  c->grid_ovf = grid_ovf[0];
  c->tgrid_ovf = tgrid_ovf[0];
  if (star_s) {
    c->collide_calls += star_s[0]; c->path_free_calls += star_s[1];
    c->star_rounds += star_s[2]; c->star_passes += star_s[3]; c->star_members += star_s[4]; c->star_rewires += star_s[5];
  }
}

// ------------------------------------------------------------------ wave end, wide
// The end of a wave (src/forest.h:160-201) as a launch of many workgroups - the one-workgroup kernel of rounds 2-3 spent
// 30 of its 34 us on two passes of scattered accesses that a single CU retires at about one address per cycle.  The
// exhausted slots' claims are normally posted by the wave's last append; after a resumed or an empty wave the launch posts
// them itself and its workgroups meet at a counter first.  256 slots per workgroup:
//   owner flags (the slot that holds its node's claim) -> the workgroup's count, published like k_commit's words
//   ((sequence << 32) | count, word KW_CNT of the workgroup's line) -> closed-list positions = closed_n + the LOWER
//   workgroups' counts + rank in the workgroup (decoupled look-back: a workgroup waits for lower ones only, and
//   workgroups start in index order) -> closed list, node flags, removed frontier positions (atomicOr on rm_words)
//   -> the workgroup that finishes last (a counter) adds the removal prefix per frontier word and the termination tests.
#define KW_CNT 15          // word of the workgroup's wg_pub line (k_commit uses 0..12)
// The wave's status for the host: the control block as it stands, into slot (status_seq % ring) of the pinned ring (round 5:
// a copy launch behind every wave cost ~5 us of stream time).  One wavefront (threadIdx < 64) of ONE workgroup calls it,
// after everything this launch writes to the control block; the reads go past the vector L1 (this CU read the block
// earlier in the launch).  The block's own status_seq names it; the counter then moves on.
__device__ void status_publish(const DevForestView& f) {
  if (!f.host_status || threadIdx.x >= 64) return;
  DevCtrl* c = f.ctrl;
  const int sq = __hip_atomic_load(&c->status_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int32_t* dst = reinterpret_cast<int32_t*>(f.host_status + (sq & (SFFK_STATUS_RING - 1)));
  const int32_t* src = reinterpret_cast<const int32_t*>(c);
  for (int w = threadIdx.x; w < (int)(sizeof(DevCtrl) / 4); w += 64)
    dst[w] = __hip_atomic_load(src + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __builtin_amdgcn_wave_barrier();
  if (threadIdx.x == 0) c->status_seq = sq + 1;
}
__global__ __launch_bounds__(256) void k_wave_end_wide(DevForestView f, const int32_t* __restrict__ grid_ovf,
                                                       const int32_t* __restrict__ tgrid_ovf, unsigned long long* star_acc) {
  __shared__ unsigned long long s_words[4];
  __shared__ unsigned long long s_sum;
  __shared__ int s_last;
  __shared__ int s_pref[256];
  __shared__ unsigned long long star_s[SFFK_STAR_ACC];
  DevCtrl* c = f.ctrl;
  if (c->halt || !c->in_wave) {   // (nothing to end: the block is the wave's status as it is)
    if (blockIdx.x == 0) status_publish(f);
    return;
  }
  const bool from_closed = c->use_closed != 0;
  const bool post_claims = !from_closed && !c->claims_done;   // (not posted by the wave's last append: a resumed or empty wave)
  const int n_fail = from_closed ? 0 : c->act_cnt;
  const int nwg = n_fail > 0 ? (n_fail + 255) >> 8 : 1;
  const int b = blockIdx.x;
  if (b >= nwg) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned seq = (unsigned)f.commit_seq[3] + 1u;
  const int fn = c->frontier_n, cn0 = c->closed_n;
  const int32_t* act = act_now(f);
  const int32_t* nodes = f.ulist;
  unsigned long long* pub = f.wg_pub + (size_t)b * SFFK_PUB_WORDS + KW_CNT;
  const int e = b * 256 + (int)threadIdx.x;
  const bool on = e < n_fail;
  const int sl = on ? act[e] : 0;
  if (post_claims) {
    // every exhausted slot claims its node (atomicMin of its place in the list), then all workgroups of the launch meet:
    // at most wave / 256 of them, all resident, so a counter they spin on is a safe barrier
    if (on) { const int nd0 = f.slot_node[sl]; atomicMin(&f.claim[nd0], e); f.ulist[e] = nd0; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      atomicAdd(&f.commit_seq[5], 1);
      for (int spin = 0; spin < KC_SPIN_LIMIT; ++spin) {
        if (__hip_atomic_load(&f.commit_seq[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= nwg) break;
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
  }
  // ---- the slot that owns its node's claim; what the closed-list pass needs is requested with it
  const int nd = on ? __hip_atomic_load(&nodes[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1;
  const int own = nd >= 0 ? __hip_atomic_load(&f.claim[nd], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1;
  const int ps = on ? f.slot_pos[sl] : 0;
  const int fl = nd >= 0 ? f.nflag[nd] : 0;
  // (priority mode: a node may be held again after it has been closed - put back by another slot - and is closed once,
  // src/forest.h:174-177)
  const bool mine = nd >= 0 && own == e && !(f.prio.n_heaps && (fl & 1));
  const unsigned long long m = __ballot(mine);
  if (lane == 0) s_words[wave] = m;
  __syncthreads();
  int cnt = 0, before = 0;
  for (int w = 0; w < 4; ++w) { const int pc = __popcll(s_words[w]); cnt += pc; if (w < wave) before += pc; }
  if (threadIdx.x == 0) kc_publish(pub, seq, (unsigned)cnt);
  // ---- closed-list positions: the lower workgroups' counts
  unsigned long long part = 0;
  if ((int)threadIdx.x < b) part = kc_wait(f.wg_pub + (size_t)threadIdx.x * SFFK_PUB_WORDS + KW_CNT, seq, &c->fault_pending);
  const int base = (int)kc_block_sum(part, &s_sum);
  if (mine) {
    const int rank = base + before + __popcll(m & ((1ULL << lane) - 1ULL));
    f.closed[cn0 + rank] = nd;
    f.nflag[nd] = (uint8_t)((fl & ~2) | 1);
    if (!f.prio.n_heaps) atomicOr(&f.rm_words[ps >> 6], 1ULL << (ps & 63));
  }
  // ---- the workgroup that is through last: removal prefix, termination
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) s_last = atomicAdd(&f.commit_seq[4], 1) == nwg - 1;
  __syncthreads();
  if (!s_last) return;
  if (star_acc) {   // SFF*: the wave's sub-counters (64 lines, k_star_apply) are folded into the control block below
    if (threadIdx.x < SFFK_STAR_ACC) star_s[threadIdx.x] = 0ULL;
    __syncthreads();
    for (int q = threadIdx.x; q < 64 * SFFK_STAR_ACC; q += 256) {
      const unsigned long long v = star_acc[q];
      if (v) { atomicAdd(&star_s[q % SFFK_STAR_ACC], v); star_acc[q] = 0ULL; }
    }
    __syncthreads();
  }
  unsigned long long tot = 0;
  for (int w = threadIdx.x; w < nwg && n_fail > 0; w += 256) tot += (unsigned long long)(unsigned)kc_wait(f.wg_pub + (size_t)w * SFFK_PUB_WORDS + KW_CNT, seq, &c->fault_pending);
  const int removed = (int)kc_block_sum(tot, &s_sum);
  const int nw = (fn + 63) >> 6;
  if (removed > 0 && !f.prio.n_heaps) {
    // exclusive prefix of the removed positions per 64-entry word (k_frontier_compact shifts by it): every thread sums a
    // contiguous run of words, the runs are scanned through LDS
    const int per = (nw + 255) / 256;
    const int w0 = threadIdx.x * per;
    int mine_n = 0;
    for (int k = 0; k < per; ++k) {
      const int w = w0 + k;
      if (w < nw) mine_n += __popcll(__hip_atomic_load(&f.rm_words[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    s_pref[threadIdx.x] = mine_n;
    __syncthreads();
    if (threadIdx.x < 64) {
      int run = 0;
      for (int q = 0; q < 256; q += 64) {
        const int v = s_pref[q + lane];
        int inc = v;
        for (int off = 1; off < 64; off <<= 1) {
          const int o = __shfl_up(inc, off);
          if (lane >= off) inc += o;
        }
        s_pref[q + lane] = run + inc - v;
        run += __shfl(inc, 63);
      }
    }
    __syncthreads();
    int run = s_pref[threadIdx.x];
    for (int k = 0; k < per; ++k) {
      const int w = w0 + k;
      if (w >= nw) break;
      f.rm_pref[w] = run;
      run += __popcll(__hip_atomic_load(&f.rm_words[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    f.commit_seq[3] = (int32_t)seq;
    f.commit_seq[4] = 0;
    f.commit_seq[5] = 0;
    wave_end_control(f, c, removed, fn, from_closed, n_fail, grid_ovf, tgrid_ovf, star_acc ? star_s : nullptr);
    c->wprof[7] += 1ULL;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the control block's stores have reached the L2)
  }
  __syncthreads();          // (... before the first wavefront reads the block back past the L1)
  status_publish(f);
}

// order-preserving removal of the marked positions: old buffer -> the other one (selected by k_wave_end_wide already)
__global__ __launch_bounds__(256) void k_frontier_compact(DevForestView f) {
  const DevCtrl* c = f.ctrl;
  // the exhausted slots' claims of the wave that just ended (k_wave_end_wide has looked at them)
  for (int e = blockIdx.x * 256 + threadIdx.x; e < c->clear_n; e += gridDim.x * 256) f.claim[f.ulist[e]] = 0x7fffffff;
  const int n = c->compact_from;
  if (n <= 0) return;                       // nothing was removed in this wave (or the kernel ran before a wave ended)
  const int32_t* src = c->front_sel ? f.frontier : f.frontier2;   // (front_sel already names the NEW buffer)
  int32_t* dst = c->front_sel ? f.frontier2 : f.frontier;
  // (a word's 64 positions are one wavefront's lanes in one step: all of them have read it when lane 0 clears it for
  // the next wave)
  const int n_up = (n + 63) & ~63;
  for (int r = blockIdx.x * 256 + threadIdx.x; r < n_up; r += gridDim.x * 256) {
    const unsigned long long w = f.rm_words[r >> 6];
    const int pref = f.rm_pref[r >> 6];
    __builtin_amdgcn_wave_barrier();
    if ((r & 63) == 0 && w) f.rm_words[r >> 6] = 0ULL;
    if (r >= n || ((w >> (r & 63)) & 1ULL)) continue;
    dst[r - pref - __popcll(w & ((1ULL << (r & 63)) - 1ULL))] = src[r];
  }
}

// ------------------------------------------------------------------ multi-GPU: answer records of a round
__global__ __launch_bounds__(256) void k_pack_records(ResolveArgs A, int rank, int world, int32_t* __restrict__ send) {
  const DevForestView& f = A.f;
  if (f.ctrl->halt) return;
  const int n = f.ctrl->n_act;
  const int W = record_words_dev(A.nbcap);
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const int j = (int)(t / W), k = (int)(t % W);
  const int i = j * world + rank;
  if (i >= n) return;
  const size_t s0 = (size_t)i * A.stride;
  int32_t v;
  if (k == 0) v = A.rec_flags[i];
  else if (k == 1) v = A.rec_nnb[i];
  else if (k == 2) v = A.pose_hit[i];
  else if (k == 3) v = 0;
  else if (k < 4 + A.nbcap) v = A.rec_nb[(size_t)i * A.nbcap + (k - 4)];
  else if (k < 4 + 2 * A.nbcap) v = A.rec_meta[(size_t)i * A.nbcap + (k - 4 - A.nbcap)];
  else if (k < 4 + 2 * A.nbcap + A.stride) v = A.seg_ns[s0 + (k - 4 - 2 * A.nbcap)];
  else v = A.first_hit[s0 + (k - 4 - 2 * A.nbcap - A.stride)];
  send[(size_t)j * W + k] = v;
}
__global__ __launch_bounds__(256) void k_unpack_records(ResolveArgs A, int rank, int world, int per_rank,
                                                        const int32_t* __restrict__ recv) {
  const DevForestView& f = A.f;
  if (f.ctrl->halt) return;
  const int n = f.ctrl->n_act;
  const int W = record_words_dev(A.nbcap);
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const int i = (int)(t / W), k = (int)(t % W);
  if (i >= n || i % world == rank) return;
  const int32_t v = recv[((size_t)(i % world) * per_rank + (size_t)(i / world)) * W + k];
  const size_t s0 = (size_t)i * A.stride;
  if (k == 0) A.rec_flags[i] = v;
  else if (k == 1) A.rec_nnb[i] = v;
  else if (k == 2) A.pose_hit[i] = (uint8_t)v;
  else if (k == 3) { }
  else if (k < 4 + A.nbcap) A.rec_nb[(size_t)i * A.nbcap + (k - 4)] = v;
  else if (k < 4 + 2 * A.nbcap) A.rec_meta[(size_t)i * A.nbcap + (k - 4 - A.nbcap)] = v;
  else if (k < 4 + 2 * A.nbcap + A.stride) A.seg_ns[s0 + (k - 4 - 2 * A.nbcap)] = v;
  else A.first_hit[s0 + (k - 4 - 2 * A.nbcap - A.stride)] = v;
}

__global__ __launch_bounds__(256) void k_border_rehash(DevForestView f, int n) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const unsigned long long key = ((unsigned long long)(uint32_t)f.b_n1[e] << 32) | ((unsigned long long)(uint32_t)f.b_n2[e] + 1ULL);
  const size_t h = border_slot(f, key);
  f.bt_val[h] = 0ULL;   // older than every future stamp
}

void launch_wave_begin_only(hipStream_t s, const DevForestView& f);
void launch_wave_begin(hipStream_t s, const DevForestView& f) {
  launch_wave_begin_only(s, f);
  if (f.prio.n_heaps) launch_prio_begin(s, f);
}
void launch_wave_begin_only(hipStream_t s, const DevForestView& f) {
  hipLaunchKernelGGL(k_wave_begin, dim3(WB_BLOCKS), dim3(256), 0, s, f);
}
void launch_commit(hipStream_t s, const ResolveArgs& a, int n_bound, const StarLaunch* star, const SampleLaunch* next) {
  if (n_bound <= 0) return;
  hipLaunchKernelGGL(k_commit, dim3((n_bound + 63) / 64), dim3(1024), 0, s, a, n_bound);
  if (a.star && star) launch_star_stage(s, a, n_bound, *star);
  // (plain SFF: + one workgroup per 64 samples that enters the round's border events, border_finalize)
  const int ab = (n_bound + 255) / 256, fb = a.star ? 0 : (n_bound + 63) / 64;
  if (next) hipLaunchKernelGGL(k_append_sample, dim3(2 * ab + fb), dim3(256), 0, s, a, *next, ab);
  else hipLaunchKernelGGL(k_append, dim3(ab + fb), dim3(256), 0, s, a, ab);
}
void launch_wave_end(hipStream_t s, const DevForestView& f, const int32_t* grid_ovf, const int32_t* tgrid_ovf,
                     unsigned long long* star_acc) {
  hipLaunchKernelGGL(k_wave_end_wide, dim3((f.wave + 255) / 256), dim3(256), 0, s, f, grid_ovf, tgrid_ovf, star_acc);
  hipLaunchKernelGGL(k_frontier_compact, dim3(512), dim3(256), 0, s, f);
}
void launch_pack_records(hipStream_t s, const ResolveArgs& a, int rank, int world, int n_bound, int32_t* send) {
  const int per_rank = (n_bound + world - 1) / world;
  const long long threads = (long long)per_rank * record_words(a.nbcap);
  hipLaunchKernelGGL(k_pack_records, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, a, rank, world, send);
}
void launch_unpack_records(hipStream_t s, const ResolveArgs& a, int rank, int world, int n_bound, const int32_t* recv) {
  const int per_rank = (n_bound + world - 1) / world;
  const long long threads = (long long)n_bound * record_words(a.nbcap);
  hipLaunchKernelGGL(k_unpack_records, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, a, rank, world, per_rank, recv);
}
void launch_border_rehash(hipStream_t s, const DevForestView& f, int n) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_border_rehash, dim3((n + 255) / 256), dim3(256), 0, s, f, n);
}

}  // namespace sffk
