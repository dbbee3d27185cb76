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
//   k_resolve     one workgroup: the in-order commit of one round.  Samples the wide k_settle could not settle are
//                 resolved by a fixed-point iteration over their (rare) dependencies on EARLIER samples of the same
//                 round; accepted samples get their node ids by a prefix sum in slot order and are appended to the
//                 node store, the neighbour grid and the frontier; border events are de-duplicated "first in slot
//                 order wins" through a stamped hash table; then the next round's active list is built
//   k_wave_end    one workgroup: exhausted slots move their node to the closed list (first occurrence in slot
//                 order), order-preserving frontier compaction (src/forest.h:160-163), termination tests (:184-201)
//
// Everything here is integer / index bookkeeping plus the few fp64 expressions of expandNode, evaluated in the same
// order as the host engine and the CPU oracle (-ffp-contract=off), so the forests are bit-identical.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"
#include "sff_geom.h"
#include "kernels_dev.h"

namespace sffk {

using namespace sffg;

#define DF_THREADS 1024
#define DF_WAVES (DF_THREADS / 64)

// exclusive prefix sum of one int per thread over the workgroup; returns the thread's offset, *total = sum
__device__ __forceinline__ int block_scan(int v, int* total, int* wsum /* DF_WAVES + 1 ints of LDS */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(inc, off);
    if (lane >= off) inc += o;
  }
  __syncthreads();   // (wsum may still be read from a previous call)
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int w = 0; w < DF_WAVES; ++w) { const int t = wsum[w]; wsum[w] = run; run += t; }
    wsum[DF_WAVES] = run;
  }
  __syncthreads();
  *total = wsum[DF_WAVES];
  return wsum[wave] + inc - v;
}

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

// The next round's active list: the failing slots in slot order, cut at the iteration cap (src/forest.h:155:
// i < ThresholdMisses && expandResult && iter < maxIterations).  Called by all threads of the single workgroup.
__device__ void round_begin(const DevForestView& f, int* wsum) {
  DevCtrl* c = f.ctrl;
  const int n_slots = c->n_slots;
  __shared__ int run_s;
  if (threadIdx.x == 0) run_s = 0;
  __syncthreads();
  for (int base = 0; base < n_slots; base += DF_THREADS) {
    const int s = base + threadIdx.x;
    const int fail = (s < n_slots && f.slot_fail[s]) ? 1 : 0;
    int tot;
    const int off = block_scan(fail, &tot, wsum);
    if (fail) f.act_slot[run_s + off] = s;
    __syncthreads();
    if (threadIdx.x == 0) run_s += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int cnt = run_s;
    int n = 0;
    if (c->round < f.threshold_misses && cnt > 0 && c->iter < f.max_iterations && !c->solved) {
      const int left = f.max_iterations - c->iter;
      n = cnt < left ? cnt : left;
    }
    c->n_act = n;
    if (n > 0) {
      c->round += 1;
      c->iter0 = c->iter;
      c->iter += n;
      c->N0 = c->n_nodes;
      c->words_base = c->cursor;
      c->cursor += (unsigned long long)f.words_per * (unsigned long long)n;
      c->rounds += 1;
      c->round_nodes += (unsigned long long)(c->n_nodes + n);
      c->round_queries += (unsigned long long)n;
      // the arrays a round can grow: one node / frontier entry / border per sample at most
      if (c->n_nodes + n > f.node_cap - 8 || c->n_borders + n > f.border_cap) {
        c->fault = SFFK_FAULT_CAPACITY;
        c->halt = 1;
      } else if ((unsigned long long)(c->n_borders + n) * 2ULL > f.bt_mask + 1ULL) {
        c->fault = SFFK_FAULT_BORDER_TABLE;
        c->halt = 1;
      }
      if (c->halt) {   // nothing of this round has happened yet: hand the host the state before it
        c->round -= 1;
        c->iter = c->iter0;
        c->cursor = c->words_base;
        c->rounds -= 1;
        c->round_nodes -= (unsigned long long)(c->n_nodes + n);
        c->round_queries -= (unsigned long long)n;
        c->n_act = 0;
      }
    }
  }
  __syncthreads();
}

// ------------------------------------------------------------------ wave begin
__global__ __launch_bounds__(DF_THREADS) void k_wave_begin(DevForestView f) {
  __shared__ int wsum[DF_WAVES + 1];
  __shared__ int any_redraw;
  DevCtrl* c = f.ctrl;
  if (c->halt) return;
  if (c->in_wave) {   // resuming inside a wave (after the host handled a fault): only rebuild the active list
    round_begin(f, wsum);
    return;
  }
  // node selection of every slot, src/forest.h:136-151 (non-priority mode): a uniform pick from the frozen frontier,
  // or from the closed list once the frontier has run empty
  const bool use_closed = c->closed_n > 0 && c->empty_frontier;
  const int pool = use_closed ? c->closed_n : c->frontier_n;
  int n_slots = f.wave < pool ? f.wave : pool;
  if (n_slots < 1) n_slots = 1;
  const int32_t* from = use_closed ? f.closed : f.frontier;
  const unsigned long long cur = c->cursor;
  if (threadIdx.x == 0) any_redraw = 0;
  __syncthreads();
  for (int s = threadIdx.x; s < n_slots; s += DF_THREADS) {
    const int pick = lemire_pick(f.ring[(cur + (unsigned long long)s) & f.ring_mask], (unsigned long long)pool);
    if (pick < 0) any_redraw = 1;
    else f.slot_node[s] = from[pick];
    f.slot_fail[s] = 1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long used = (unsigned long long)n_slots;
    if (any_redraw) {   // (probability ~ pool / 2^64 per pick) redo the picks one after another, words as they come
      unsigned long long at = cur;
      for (int s = 0; s < n_slots; ++s) {
        int pick;
        do { pick = lemire_pick(f.ring[at & f.ring_mask], (unsigned long long)pool); ++at; } while (pick < 0);
        f.slot_node[s] = from[pick];
      }
      used = at - cur;
      c->redraws += 1;
    }
    c->cursor = cur + used;
    c->n_slots = n_slots;
    c->use_closed = use_closed ? 1 : 0;
    c->round = 0;
    c->in_wave = 1;
    c->waves += 1;
  }
  __syncthreads();
  round_begin(f, wsum);
}

// ------------------------------------------------------------------ in-order commit of one round
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

__global__ __launch_bounds__(DF_THREADS) void k_resolve(ResolveArgs A) {
  __shared__ int wsum[DF_WAVES + 1];
  __shared__ int n_uns_s, undecided_s, run_s, n_ev_s;
  __shared__ unsigned long long cnt_s[3];
  const DevForestView& f = A.f;
  DevCtrl* c = f.ctrl;
  if (c->halt) return;
  const int n = c->n_act;
  if (n == 0) {
    if (c->in_wave) round_begin(f, wsum);   // (a wave that is over keeps n_act = 0)
    return;
  }
  if (*A.fault_pending) {
    // a bounded device list overflowed somewhere in this round: nothing is committed, the bookkeeping of
    // round_begin is rolled back and the host redoes the round on its unbounded path
    if (threadIdx.x == 0) {
      c->fault = SFFK_FAULT_LISTS;
      c->halt = 1;
      c->round -= 1;
      c->iter = c->iter0;
      c->cursor = c->words_base;
      c->rounds -= 1;
      c->round_nodes -= (unsigned long long)(c->N0 + n);
      c->round_queries -= (unsigned long long)n;
      c->n_act = 0;
      *A.fault_pending = 0;
    }
    return;
  }
  const int Tb = f.temp_base, N0 = c->N0, iter0 = c->iter0;
  const int stride = A.stride, nbcap = A.nbcap;
  if (threadIdx.x == 0) { n_uns_s = 0; cnt_s[0] = cnt_s[1] = cnt_s[2] = 0ULL; }
  __syncthreads();
  // ---- 1. the samples k_settle left open (code 0), in slot order
  for (int base = 0; base < n; base += DF_THREADS) {
    const int i = base + threadIdx.x;
    const int open = (i < n && A.code[i] == 0) ? 1 : 0;
    if (i < n) f.ustate[i] = open ? 0 : 1;            // 0 undecided, 1 rejected, 2 accepted
    int tot;
    const int off = block_scan(open, &tot, wsum);
    if (open) f.ulist[n_uns_s + off] = i;
    __syncthreads();
    if (threadIdx.x == 0) n_uns_s += tot;
    __syncthreads();
  }
  const int n_uns = n_uns_s;
  // ---- 2. fixed point over the dependencies on earlier samples of the round: an open sample has a free pose and
  // a free parent edge; its neighbour list (order of src/forest.h:262-300) decides.  A round-mate neighbour only
  // exists if that sample was accepted.
  unsigned long long cc = 0, pf = 0, nq = 0;
  for (int pass = 0; pass < n + 2; ++pass) {
    if (threadIdx.x == 0) undecided_s = 0;
    __syncthreads();
    int mine_undecided = 0;
    for (int u = threadIdx.x; u < n_uns; u += DF_THREADS) {
      const int i = f.ulist[u];
      if (f.ustate[i] != 0) continue;
      const size_t s0 = (size_t)i * stride;
      const int nnb = A.rec_nnb[i];
      unsigned long long c1 = 1 + (unsigned long long)A.seg_ns[s0], p1 = 1, q1 = (unsigned long long)f.n_trees;
      int verdict = 2, ev_nb = -1;
      for (int k = 0; k < nnb; ++k) {
        const int id = A.rec_nb[(size_t)i * nbcap + k];
        if (id >= Tb) {
          const int sj = f.ustate[id - Tb];
          if (sj == 0) { verdict = 0; break; }        // not known yet: next pass
          if (sj == 1) continue;                      // that sample never became a node
        }
        const int fh = A.first_hit[s0 + 1 + k];
        const bool fr = fh == 0x7fffffff;
        p1 += 1;
        c1 += fr ? (unsigned long long)A.seg_ns[s0 + 1 + k] : (unsigned long long)fh;
        if (A.rec_meta[(size_t)i * nbcap + k] & 1) {
          if (fr) { verdict = 1; break; }             // :276-280 overcrowded
        } else {
          if (fr) ev_nb = k;                          // :288-294 border entry
          verdict = 1;                                // :296-299
          break;
        }
      }
      if (verdict == 0) { mine_undecided = 1; continue; }
      cc += c1; pf += p1; nq += q1;
      f.uacc[i] = ev_nb;                              // (accepted ids are filled in below; rejected keep the event slot)
      __threadfence_block();
      f.ustate[i] = (uint8_t)verdict;
    }
    if (mine_undecided) undecided_s = 1;
    __syncthreads();
    if (!undecided_s) break;
    __syncthreads();
  }
  // ---- 3. node ids of the accepted samples: N0 + rank in slot order; append to store, grid, frontier
  if (threadIdx.x == 0) { run_s = 0; n_ev_s = 0; }
  __syncthreads();
  for (int base = 0; base < n_uns; base += DF_THREADS) {
    const int u = base + threadIdx.x;
    const int i = u < n_uns ? f.ulist[u] : 0;
    const int acc = (u < n_uns && f.ustate[i] == 2) ? 1 : 0;
    int tot;
    const int off = block_scan(acc, &tot, wsum);
    if (acc) {
      const int id = N0 + run_s + off;
      f.uacc[i] = id;
      const int ex = A.parent[i];
      const double* p = A.newpos + 6 * (size_t)i;
      const size_t o = (size_t)id;
      GridItem it;
      it.x = (float)p[0]; it.y = (float)p[1]; it.z = (float)p[2];
      it.yaw = (float)p[3]; it.pitch = (float)p[4]; it.roll = (float)p[5];
      it.id = id;
      it.tree = A.st.tree[ex];
      A.st.x[o] = it.x; A.st.y[o] = it.y; A.st.z[o] = it.z;
      A.st.yaw[o] = it.yaw; A.st.pitch[o] = it.pitch; A.st.roll[o] = it.roll;
      for (int k = 0; k < 6; ++k) A.st.pos[6 * o + k] = p[k];
      A.st.tree[o] = it.tree;
      const double pd = A.pdist[i];
      f.parent[o] = ex;                                // src/forest.h:353
      f.d_closest[o] = pd;
      f.d_root[o] = pd + f.d_root[ex];
      f.iter[o] = (uint32_t)(iter0 + i + 1);
      f.nflag[o] = 2;
      f.frontier[c->frontier_n + run_s + off] = id;    // :365
      f.slot_fail[f.act_slot[i]] = 0;
      grid_put(A.g, it);                               // flannIndex->addPoints, :367
    }
    __syncthreads();
    if (threadIdx.x == 0) run_s += tot;
    __syncthreads();
  }
  const int n_acc = run_s;
  __threadfence();
  __syncthreads();
  // ---- 4. border events (a free edge to a neighbour of another tree, :288-294), first in slot order wins
  const unsigned long long stamp_hi = (c->epoch + 1ULL) << 32;
  for (int u = threadIdx.x; u < n_uns; u += DF_THREADS) {
    const int i = f.ulist[u];
    if (f.ustate[i] != 1) continue;
    const int k = f.uacc[i];
    if (k < 0) continue;
    const int raw = A.rec_nb[(size_t)i * nbcap + k];
    const int nb = raw >= Tb ? f.uacc[raw - Tb] : raw;
    const int ex = A.parent[i];
    const int a = nb < ex ? nb : ex, b = nb < ex ? ex : nb;
    const unsigned long long key = ((unsigned long long)(uint32_t)a << 32) | ((unsigned long long)(uint32_t)b + 1ULL);
    const size_t h = border_slot(f, key);
    atomicMin(&f.bt_val[h], stamp_hi | (unsigned long long)(uint32_t)i);
  }
  __threadfence();
  __syncthreads();
  for (int base = 0; base < n_uns; base += DF_THREADS) {
    const int u = base + threadIdx.x;
    int keep = 0, i = 0, nb = 0, ex = 0, a = 0, b = 0;
    if (u < n_uns) {
      i = f.ulist[u];
      const int k = f.ustate[i] == 1 ? f.uacc[i] : -1;
      if (k >= 0) {
        const int raw = A.rec_nb[(size_t)i * nbcap + k];
        nb = raw >= Tb ? f.uacc[raw - Tb] : raw;
        ex = A.parent[i];
        a = nb < ex ? nb : ex; b = nb < ex ? ex : nb;
        const unsigned long long key = ((unsigned long long)(uint32_t)a << 32) | ((unsigned long long)(uint32_t)b + 1ULL);
        // (atomic read: the stamps were written by L2 atomics a moment ago)
        const unsigned long long owner = __hip_atomic_load(&f.bt_val[border_slot(f, key)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        keep = owner == (stamp_hi | (unsigned long long)(uint32_t)i) ? 1 : 0;
      }
    }
    int tot;
    const int off = block_scan(keep, &tot, wsum);
    if (keep) {
      const int at = c->n_borders + n_ev_s + off;
      const int ta = A.st.tree[nb], tb = A.st.tree[ex];
      f.b_n1[at] = a; f.b_n2[at] = b;
      f.b_ta[at] = ta < tb ? ta : tb; f.b_tb[at] = ta < tb ? tb : ta;
      double pn[6], pe[6];
      for (int q = 0; q < 6; ++q) { pn[q] = A.st.pos[6 * (size_t)nb + q]; pe[q] = A.st.pos[6 * (size_t)ex + q]; }
      f.b_dist[at] = f.d_root[nb] + f.d_root[ex] + dist6(pn, pe);   // :291
      f.pair[(size_t)ta * f.n_trees + tb] = 1;
      f.pair[(size_t)tb * f.n_trees + ta] = 1;
    }
    __syncthreads();
    if (threadIdx.x == 0) n_ev_s += tot;
    __syncthreads();
  }
  // ---- 5. counters, sizes
  for (int off = 32; off > 0; off >>= 1) { cc += __shfl_xor(cc, off); pf += __shfl_xor(pf, off); nq += __shfl_xor(nq, off); }
  if ((threadIdx.x & 63) == 0 && (cc | pf | nq)) { atomicAdd(&cnt_s[0], cc); atomicAdd(&cnt_s[1], pf); atomicAdd(&cnt_s[2], nq); }
  __syncthreads();
  if (threadIdx.x == 0) {
    c->collide_calls += cnt_s[0] + A.bulk[0];
    c->path_free_calls += cnt_s[1] + A.bulk[1];
    c->nn_queries += cnt_s[2] + A.bulk[2];
    c->poses_executed += A.bulk[4];
    c->segments_executed += A.bulk[5];
    c->samples_executed += A.bulk[6];
    c->work_items += (unsigned long long)A.round_ctrl[2];
    c->n_nodes = N0 + n_acc;
    c->frontier_n += n_acc;
    c->n_borders += n_ev_s;
    c->n_unsettled += n_uns;
    c->epoch += 1ULL;
  }
  __threadfence();
  __syncthreads();
  round_begin(f, wsum);
}

// ------------------------------------------------------------------ wave end
__global__ __launch_bounds__(DF_THREADS) void k_wave_end(DevForestView f, const int32_t* __restrict__ grid_ovf,
                                                         const int32_t* __restrict__ tgrid_ovf) {
  __shared__ int wsum[DF_WAVES + 1];
  __shared__ int run_s, removed_s;
  DevCtrl* c = f.ctrl;
  if (c->halt || !c->in_wave) return;
  const int n_slots = c->n_slots;
  const bool from_closed = c->use_closed != 0;
  // ---- exhausted slots: the node leaves the frontier for the closed list (src/forest.h:160-178); a node held by
  // several slots moves once, at its first slot
  if (!from_closed) {
    for (int s = threadIdx.x; s < n_slots; s += DF_THREADS)
      if (f.slot_fail[s] && (f.nflag[f.slot_node[s]] & 2)) atomicMin(&f.claim[f.slot_node[s]], s);
  }
  if (threadIdx.x == 0) { run_s = 0; removed_s = 0; }
  __threadfence();
  __syncthreads();
  if (!from_closed) {
    for (int base = 0; base < n_slots; base += DF_THREADS) {
      const int s = base + threadIdx.x;
      int win = 0, node = 0;
      if (s < n_slots && f.slot_fail[s]) {
        node = f.slot_node[s];
        win = ((f.nflag[node] & 2) && __hip_atomic_load(&f.claim[node], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == s) ? 1 : 0;
      }
      int tot;
      const int off = block_scan(win, &tot, wsum);
      if (win) f.closed[c->closed_n + run_s + off] = node;
      __syncthreads();
      if (threadIdx.x == 0) run_s += tot;
      __syncthreads();
    }
    // (flags and claims are cleared only now: every slot of a node had to see them)
    for (int s = threadIdx.x; s < n_slots; s += DF_THREADS) {
      if (!f.slot_fail[s]) continue;
      const int node = f.slot_node[s];
      if (__hip_atomic_load(&f.claim[node], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == s) {
        f.nflag[node] = (uint8_t)((f.nflag[node] & ~2) | 1);
        f.claim[node] = 0x7fffffff;
      }
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) { c->closed_n += run_s; removed_s = run_s; run_s = 0; }
    __syncthreads();
  }
  // ---- order-preserving compaction of the frontier (the reference erases the entries one by one)
  if (removed_s > 0) {
    const int fn = c->frontier_n;
    for (int base = 0; base < fn; base += DF_THREADS) {
      const int r = base + threadIdx.x;
      const int node = r < fn ? f.frontier[r] : 0;
      const int keep = (r < fn && (f.nflag[node] & 2)) ? 1 : 0;
      int tot;
      const int off = block_scan(keep, &tot, wsum);   // (its barriers sit between every read and every write of the chunk)
      if (keep) f.frontier[run_s + off] = node;
      __syncthreads();
      if (threadIdx.x == 0) run_s += tot;
      __syncthreads();
    }
    if (threadIdx.x == 0) c->frontier_n = run_s;
    __syncthreads();
  }
  // ---- termination (src/forest.h:184-201)
  if (threadIdx.x == 0) {
    c->empty_frontier = c->frontier_n == 0 ? 1 : 0;
    if (!c->solved && c->empty_frontier) {
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
    const bool budget = f.node_budget > 0 && c->n_nodes >= f.node_budget;
    c->terminated = (c->solved || c->iter >= f.max_iterations || budget) ? 1 : 0;
    c->halt = c->terminated;
    c->in_wave = 0;
    c->n_act = 0;
    c->grid_ovf = grid_ovf[0];
    c->tgrid_ovf = tgrid_ovf[0];
  }
}

__global__ __launch_bounds__(256) void k_border_rehash(DevForestView f, int n) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const unsigned long long key = ((unsigned long long)(uint32_t)f.b_n1[e] << 32) | ((unsigned long long)(uint32_t)f.b_n2[e] + 1ULL);
  const size_t h = border_slot(f, key);
  f.bt_val[h] = 0ULL;   // older than every future stamp
}

void launch_wave_begin(hipStream_t s, const DevForestView& f) {
  hipLaunchKernelGGL(k_wave_begin, dim3(1), dim3(DF_THREADS), 0, s, f);
}
void launch_resolve(hipStream_t s, const ResolveArgs& a) {
  hipLaunchKernelGGL(k_resolve, dim3(1), dim3(DF_THREADS), 0, s, a);
}
void launch_wave_end(hipStream_t s, const DevForestView& f, const int32_t* grid_ovf, const int32_t* tgrid_ovf) {
  hipLaunchKernelGGL(k_wave_end, dim3(1), dim3(DF_THREADS), 0, s, f, grid_ovf, tgrid_ovf);
}
void launch_border_rehash(hipStream_t s, const DevForestView& f, int n) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_border_rehash, dim3((n + 255) / 256), dim3(256), 0, s, f, n);
}

}  // namespace sffk
