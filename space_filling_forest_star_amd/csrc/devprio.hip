// devprio.hip - the priority-frontier mode of SpaceForest::Solve on the device-resident engine (PrioView, kernels.h).
//
//   k_prio_begin  one wavefront: the wave's picks in slot order - tree, heap, "minimum or random entry" - exactly as
//                 src/forest.h:126-147 draws them (libstdc++ uniform_int = Lemire with rejection, uniform_real), with the
//                 heap sizes counted down as the slots take their nodes; control block of the wave
//   k_prio_pops   one workgroup per heap: its slots' pops in slot order (src/heap.h:175-238: pop / pop at index)
//   k_prio_end    one workgroup per heap, behind the wave's rounds: pushes of the wave's new nodes of the heap's tree in
//                 creation order (src/forest.h:360-363), then per slot of the tree in slot order: an exhausted slot's node
//                 leaves the tree's OTHER heaps (:164-173), a slot that expanded its node puts it back onto the heap it
//                 came from (:178-180); the last heap through says whether every heap is empty (:184-191)
// The order of operations per heap is the reference's; heaps do not see each other, so one workgroup per heap is exact.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"
#include "sff_geom.h"
#include "kernels_dev.h"

namespace sffk {

using namespace sffg;

__device__ __forceinline__ int prio_lemire(unsigned long long word, unsigned long long range) {   // (devforest.hip: lemire_pick)
  const unsigned long long lo = word * range;
  const unsigned long long hi = __umul64hi(word, range);
  if (lo < range) {
    const unsigned long long thr = (0ULL - range) % range;
    if (lo < thr) return -1;
  }
  return (int)hi;
}

// ---- one heap in HBM.  All lanes of the wavefront run these with the same values (uniform addresses: one request).
// A launch reads what it has written itself: loads go past the vector L1 (relaxed agent-scope = sc1, served by the L2
// where the stores land), an operation never reads a word it has written, and the stores of one operation are drained
// before the next one starts.
struct HeapRef {
  int32_t* v; double* key; int32_t* pos; int32_t* size_p;
  int n;
};
__device__ __forceinline__ int hl_i32(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double hl_f64(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void heap_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void heap_put(const HeapRef& h, int i, int node, double k) { h.v[i] = node; h.key[i] = k; h.pos[node] = i; }
// Heap::BubbleDown of (node, k) standing at `index` (src/heap.h:122-149: the smaller child, ties to the left)
__device__ void heap_down(HeapRef& h, int index, int node, double k) {
  while (true) {
    const int l = 2 * index + 1, r = l + 1;
    if (l >= h.n) break;
    const double kl = hl_f64(h.key + l);
    const double kr = r < h.n ? hl_f64(h.key + r) : 0.0;
    int mi = index;
    double km = k;
    if (k > kl) { mi = l; km = kl; }
    if (r < h.n && km > kr) { mi = r; km = kr; }
    if (mi == index) break;
    heap_put(h, index, hl_i32(h.v + mi), km);
    index = mi;
  }
  heap_put(h, index, node, k);
}
// Heap::BubbleUp of (node, k) standing at `index` (src/heap.h:151-163)
__device__ void heap_up(HeapRef& h, int index, int node, double k) {
  while (index > 0) {
    const int p = (index - 1) / 2;
    const double kp = hl_f64(h.key + p);
    if (!(kp > k)) break;
    heap_put(h, index, hl_i32(h.v + p), kp);
    index = p;
  }
  heap_put(h, index, node, k);
}
__device__ int heap_pop(HeapRef& h) {                       // Heap::pop()
  const int mn = hl_i32(h.v);
  h.pos[mn] = -1;
  h.n -= 1;
  if (h.n > 0) heap_down(h, 0, hl_i32(h.v + h.n), hl_f64(h.key + h.n));
  heap_drain();
  return mn;
}
__device__ int heap_pop_at(HeapRef& h, int id) {            // Heap::pop(index)
  const int size = h.n;
  if (id >= size) return -1;
  const int val = hl_i32(h.v + id);
  const double old_cost = hl_f64(h.key + id);
  h.pos[val] = -1;
  h.n -= 1;
  if (id != size - 1) {
    const int last = hl_i32(h.v + size - 1);
    const double new_cost = hl_f64(h.key + size - 1);
    if (new_cost < old_cost) heap_up(h, id, last, new_cost); else heap_down(h, id, last, new_cost);
  }
  heap_drain();
  return val;
}
__device__ void heap_push(HeapRef& h, int node, double k) {  // Heap::push
  h.n += 1;
  heap_up(h, h.n - 1, node, k);
  heap_drain();
}
__device__ __forceinline__ int32_t* prio_act_now(const DevForestView& f) { return f.ctrl->act_sel ? f.act_slot2 : f.act_slot; }
__device__ __forceinline__ HeapRef heap_of(const PrioView& P, int h) {
  HeapRef r;
  r.v = P.v + (size_t)h * P.cap; r.key = P.key + (size_t)h * P.cap; r.pos = P.pos + (size_t)h * P.cap;
  r.size_p = P.size + h;
  r.n = P.size[h];
  return r;
}

// Sizes the first round (devforest.hip: round_begin_scalars) - kept in step with it
__device__ void prio_round_begin(const DevForestView& f, DevCtrl* c) {
  const int cnt = c->act_cnt;
  int n = 0;
  if (c->round < f.threshold_misses && cnt > 0 && c->iter < f.max_iterations && !c->solved) {
    const int left = f.max_iterations - c->iter;
    n = cnt < left ? cnt : left;
  }
  c->n_act = n;
  if (n > 0) {
    if (c->n_nodes + n > f.node_cap - 8 || c->n_borders + n > f.border_cap) {
      c->fault = SFFK_FAULT_CAPACITY; c->halt = 1; c->n_act = 0;
    } else if ((unsigned long long)(c->n_borders + n) * 2ULL > f.bt_mask + 1ULL) {
      c->fault = SFFK_FAULT_BORDER_TABLE; c->halt = 1; c->n_act = 0;
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

// ------------------------------------------------------------------ the wave's picks
__global__ __launch_bounds__(64) void k_prio_begin(DevForestView f) {
  __shared__ int s_sz[SFFK_PRIO_MAX_HEAPS];
  __shared__ int s_tne[SFFK_PRIO_MAX_HEAPS];     // per tree: its non-empty heaps
  __shared__ unsigned long long s_w[128];        // engine words [w_lo, w_lo + 128)
  DevCtrl* c = f.ctrl;
  const PrioView& P = f.prio;
  const int lane = threadIdx.x;
  // (launched BEHIND k_wave_begin, which has handled a halt, a wave to resume and - every heap empty - a wave that takes
  // its nodes from the closed list: in all of these a wave is in progress or nothing is to be done)
  if (c->halt || c->in_wave) return;
  const int H = P.n_heaps, T = f.n_trees;
  for (int h = lane; h < H; h += 64) s_sz[h] = P.size[h];
  __builtin_amdgcn_wave_barrier();
  int total_ne = 0, pool = 0;
  for (int t = 0; t < T; ++t) {
    int ne = 0;
    for (int h = P.base[t]; h < P.base[t + 1]; ++h) ne += s_sz[h] > 0 ? 1 : 0;
    if (lane == 0) s_tne[t] = ne;
    total_ne += ne;
    if (P.base[t + 1] > P.base[t]) pool += s_sz[P.base[t]];   // src/forest.h:128-131 (the first heap of every tree)
  }
  __builtin_amdgcn_wave_barrier();
  int n_slots = f.wave < pool ? f.wave : pool;
  if (n_slots < 1) n_slots = 1;
  const unsigned long long cur = c->cursor;
  unsigned long long at = cur, w_lo = cur;
  auto refill = [&]() {   // words [at, at + 64) resident
    w_lo = at;
    s_w[lane] = f.ring[(at + (unsigned long long)lane) & f.ring_mask];
    s_w[64 + lane] = f.ring[(at + 64ULL + (unsigned long long)lane) & f.ring_mask];
    __builtin_amdgcn_wave_barrier();
  };
  refill();
  auto next_word = [&]() -> unsigned long long {
    if (at - w_lo >= 128ULL) refill();
    const unsigned long long w = s_w[at - w_lo];
    ++at;
    return w;
  };
  auto draw_int = [&](int range) -> int {        // RandGen::randomIntMinMax(0, range - 1)
    int v;
    do { v = prio_lemire(next_word(), (unsigned long long)range); } while (v < 0);
    return v;
  };
  int32_t* act = prio_act_now(f);
  int made = 0;
  int my_t = 0, my_h = 0, my_i = 0;
  for (int s = 0; s < n_slots; ++s) {
    if (total_ne == 0) break;                    // every frontier node is already held by a slot
    if (at - w_lo >= 96ULL) refill();            // (a slot draws a handful of words)
    int t;
    do { t = draw_int(T); } while (s_tne[t] == 0);
    const int b0 = P.base[t], nh = P.base[t + 1] - b0;
    int hp;
    do { hp = draw_int(nh); } while (s_sz[b0 + hp] == 0);
    const int size = s_sz[b0 + hp];
    int idx = -1;
    if (!(uniform_real(next_word(), 0.0, 1.0) <= P.bias)) idx = draw_int(size);   // :143-147
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      s_sz[b0 + hp] = size - 1;
      if (size == 1) s_tne[t] -= 1;
    }
    __builtin_amdgcn_wave_barrier();
    if (size == 1) total_ne -= 1;
    if ((s & 63) == lane) { my_t = t; my_h = hp; my_i = idx; }
    if ((s & 63) == 63) {
      const int o = s - 63 + lane;
      P.slot_tree[o] = my_t; P.slot_heap[o] = my_h; P.slot_idx[o] = my_i; act[o] = o;
    }
    ++made;
  }
  if ((made & 63) != 0 && lane < (made & 63)) {
    const int o = (made & ~63) + lane;
    P.slot_tree[o] = my_t; P.slot_heap[o] = my_h; P.slot_idx[o] = my_i; act[o] = o;
  }
  if (lane == 0) {
    c->compact_from = 0;
    c->app_n = 0;
    c->cursor = at;
    c->n_slots = made;
    c->act_cnt = made;
    c->use_closed = 0;
    c->round = 0;
    c->in_wave = 1;
    c->waves += 1;
    c->prio_wave = 1;
    c->prio_gen = (int32_t)c->waves;
    c->prio_n0 = c->n_nodes;
    prio_round_begin(f, c);
  }
}


// ------------------------------------------------------------------ the wave's picks, in parallel
// k_prio_begin draws the picks one slot after the other (0.75 us per slot: 12 ms for a wave of 16 384).  What makes them
// sequential is only WHERE in the engine-word stream a slot starts - a slot takes three words (tree, heap, coin) and a
// fourth when the coin asks for a random entry - and the heap sizes at its turn.  So:
//   len(p) = words of a slot that starts at stream position p (3 or 4: the coin is the word at p + 2) for EVERY position,
//   jump tables next^(2^k)(p) by pointer doubling, every slot s then walks to its own start next^s(0) in log2 steps;
//   the slot's tree / heap / coin come from its words; the random entry's index is drawn by k_prio_pops, which knows the
//   heap's size at that slot's turn (it runs the heap's slots in order);
//   the plan is valid when no draw fell into Lemire's rejection zone and no heap is asked for more nodes than it holds
//   (then no tree or heap was found empty, so the reference would not have redrawn either) - otherwise nothing is written
//   and k_prio_begin, launched behind this kernel, does the wave one slot after the other.
__global__ __launch_bounds__(1024) void k_prio_plan(DevForestView f) {
  __shared__ int s_sz[SFFK_PRIO_MAX_HEAPS];
  __shared__ int s_cnt[SFFK_PRIO_MAX_HEAPS];
  __shared__ int s_bad, s_pool;
  DevCtrl* c = f.ctrl;
  const PrioView& P = f.prio;
  const int tid = threadIdx.x;
  if (c->halt || c->in_wave || !P.plan) return;
  const int H = P.n_heaps, T = f.n_trees;
  for (int h = tid; h < H; h += 1024) { s_sz[h] = P.size[h]; s_cnt[h] = 0; }
  if (tid == 0) { s_bad = 0; s_pool = 0; }
  __syncthreads();
  if (tid < T && P.base[tid + 1] > P.base[tid]) atomicAdd(&s_pool, s_sz[P.base[tid]]);   // src/forest.h:128-131
  __syncthreads();
  int n_slots = f.wave < s_pool ? f.wave : s_pool;
  if (n_slots < 1) n_slots = 1;
  const int M = 4 * n_slots + 8;                 // stream positions a wave without redraws can reach
  const unsigned long long cur = c->cursor;
  int levels = 0;
  while ((1 << levels) < n_slots + 1) ++levels;
  int32_t* J = P.plan;                           // (levels + 1) tables of M + 1 positions
  const size_t stride = (size_t)M + 1;
  for (int p = tid; p <= M; p += 1024) {
    int nx = M;
    if (p + 3 < M) {
      const unsigned long long w = f.ring[(cur + (unsigned long long)p + 2ULL) & f.ring_mask];
      nx = p + 3 + ((uniform_real(w, 0.0, 1.0) <= P.bias) ? 0 : 1);
      if (nx > M) nx = M;
    }
    J[p] = nx;
  }
  for (int k = 1; k <= levels; ++k) {
    heap_drain();
    __syncthreads();
    const int32_t* A = J + (size_t)(k - 1) * stride;
    int32_t* B = J + (size_t)k * stride;
    for (int p = tid; p <= M; p += 1024) B[p] = hl_i32(A + hl_i32(A + p));
  }
  heap_drain();
  __syncthreads();
  auto start_of = [&](int s) -> int {
    int p = 0;
    for (int k = 0; k <= levels; ++k)
      if ((s >> k) & 1) p = hl_i32(J + (size_t)k * stride + p);
    return p;
  };
  int32_t* act = prio_act_now(f);
  for (int s = tid; s < n_slots; s += 1024) {
    const int p = start_of(s);
    bool bad = p + 4 > M;
    int t = 0, hp = 0, idx = -1;
    unsigned long long w3 = 0ULL;
    if (!bad) {
      const unsigned long long w0 = f.ring[(cur + (unsigned long long)p) & f.ring_mask];
      const unsigned long long w1 = f.ring[(cur + (unsigned long long)p + 1ULL) & f.ring_mask];
      const unsigned long long w2 = f.ring[(cur + (unsigned long long)p + 2ULL) & f.ring_mask];
      t = prio_lemire(w0, (unsigned long long)T);
      if (t < 0) bad = true;
      else {
        const int b0 = P.base[t], nh = P.base[t + 1] - b0;
        hp = nh > 0 ? prio_lemire(w1, (unsigned long long)nh) : -1;
        if (hp < 0) bad = true;
        else {
          atomicAdd(&s_cnt[b0 + hp], 1);
          if (!(uniform_real(w2, 0.0, 1.0) <= P.bias)) { idx = -2; w3 = f.ring[(cur + (unsigned long long)p + 3ULL) & f.ring_mask]; }
        }
      }
    }
    if (bad) s_bad = 1;
    P.slot_tree[s] = t; P.slot_heap[s] = hp; P.slot_idx[s] = idx; P.slot_word[s] = w3; act[s] = s;
  }
  __syncthreads();
  for (int h = tid; h < H; h += 1024) if (s_cnt[h] > s_sz[h]) s_bad = 1;
  __syncthreads();
  if (s_bad || tid != 0) return;                 // (not valid: k_prio_begin does the wave)
  const int used = start_of(n_slots);
  c->compact_from = 0;
  c->app_n = 0;
  c->cursor = cur + (unsigned long long)used;
  c->n_slots = n_slots;
  c->act_cnt = n_slots;
  c->use_closed = 0;
  c->round = 0;
  c->in_wave = 1;
  c->waves += 1;
  c->prio_wave = 1;
  c->prio_gen = (int32_t)c->waves;
  c->prio_n0 = c->n_nodes;
  prio_round_begin(f, c);
}

// ------------------------------------------------------------------ the pops of one heap, in slot order
__global__ __launch_bounds__(64) void k_prio_pops(DevForestView f) {
  const DevCtrl* c = f.ctrl;
  const PrioView& P = f.prio;
  const int h = blockIdx.x, lane = threadIdx.x;
  if (!c->prio_wave || !c->in_wave || P.gen[h] == c->prio_gen) return;      // (once per wave, right behind k_prio_begin)
  int t = 0;
  while (P.base[t + 1] <= h) ++t;
  const int hp = h - P.base[t];
  const int n_slots = c->n_slots;
  HeapRef hr = heap_of(P, h);
  for (int s0 = 0; s0 < n_slots; s0 += 64) {
    const int s = s0 + lane;
    const bool mine = s < n_slots && P.slot_tree[s] == t && P.slot_heap[s] == hp;
    const int idx = mine ? P.slot_idx[s] : 0;
    const unsigned long long wd = (mine && idx == -2) ? P.slot_word[s] : 0ULL;
    unsigned long long m = __ballot(mine);
    while (m) {
      const int l = __ffsll((long long)m) - 1;
      m &= m - 1;
      int id = __shfl(idx, l);
      if (id == -2) {   // (planned in parallel: the random entry's index is drawn here, with the heap's size at this turn)
        id = prio_lemire(__shfl(wd, l), (unsigned long long)hr.n);
        if (id < 0) { if (lane == 0) f.ctrl->fault = SFFK_FAULT_PRIO_REDRAW; id = 0; }
      }
      const int node = id < 0 ? heap_pop(hr) : heap_pop_at(hr, id);
      if (lane == 0) f.slot_node[s0 + l] = node;
    }
  }
  if (lane == 0) { *hr.size_p = hr.n; P.gen[h] = c->prio_gen; }
}

// ------------------------------------------------------------------ the end of the wave, per heap
__global__ __launch_bounds__(256) void k_prio_end(DevForestView f, NodeStoreView st) {
  __shared__ unsigned int s_fail[SFFK_DEV_MAX_GROUPS * 2];   // bit per slot: still failing (exhausted) at the wave's end
  __shared__ int s_id[256];
  __shared__ double s_key[256];
  __shared__ int s_cnt[4];
  DevCtrl* c = f.ctrl;
  const PrioView& P = f.prio;
  if (c->halt || !c->in_wave || c->n_act != 0) return;       // (the wave is not over: a fault stopped it)
  const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  int t = 0;
  while (P.base[t + 1] <= h) ++t;
  const int hp = h - P.base[t];
  const double* ref = P.ref + 6 * (size_t)h;
  HeapRef hr = heap_of(P, h);
  // ---- 1. the wave's new nodes of this tree, in creation order (src/forest.h:360-363)
  const int n0 = c->prio_n0, n1 = c->n_nodes;
  for (int b0 = n0; b0 < n1; b0 += 256) {
    const int id = b0 + tid;
    const bool mine = id < n1 && st.tree[id] == t;
    double k = 0.0;
    if (mine) {
      double p[6], r[6];
      for (int q = 0; q < 6; ++q) { p[q] = st.pos[6 * (size_t)id + q]; r[q] = ref[q]; }
      k = dist6(p, r);                                       // Distance(node, refPoint)
    }
    // ordered compaction of the batch into LDS
    const unsigned long long m = __ballot(mine);
    if (lane == 0) s_cnt[tid >> 6] = __popcll(m);
    __syncthreads();
    int before = 0, total = 0;
    for (int w = 0; w < 4; ++w) { if (w < (tid >> 6)) before += s_cnt[w]; total += s_cnt[w]; }
    if (mine) {
      const int at = before + __popcll(m & ((1ULL << lane) - 1ULL));
      s_id[at] = id; s_key[at] = k;
    }
    __syncthreads();
    if (tid < 64)
      for (int j = 0; j < total; ++j) heap_push(hr, s_id[j], s_key[j]);
    __syncthreads();
  }
  // ---- 2. the slots of this tree in slot order (src/forest.h:164-181); closed-list waves hold no heap nodes
  if (c->prio_wave) {
    const int n_slots = c->n_slots, n_fail = c->act_cnt;
    const int32_t* act = prio_act_now(f);
    for (int w = tid; w < (n_slots + 31) / 32; w += 256) s_fail[w] = 0u;
    __syncthreads();
    for (int e = tid; e < n_fail; e += 256) { const int s = act[e]; atomicOr(&s_fail[s >> 5], 1u << (s & 31)); }
    __syncthreads();
    if (tid < 64) {
      for (int s0 = 0; s0 < n_slots; s0 += 64) {
        const int s = s0 + lane;
        int op = 0;                                          // 1 = remove the node, 2 = put it back
        int node = 0;
        if (s < n_slots && P.slot_tree[s] == t) {
          const bool failing = (s_fail[s >> 5] >> (s & 31)) & 1u;
          const int sh = P.slot_heap[s];
          if (failing && sh != hp) op = 1;
          else if (!failing && sh == hp) op = 2;
          if (op) node = f.slot_node[s];
        }
        unsigned long long m = __ballot(op != 0);
        while (m) {
          const int l = __ffsll((long long)m) - 1;
          m &= m - 1;
          const int o = __shfl(op, l), nd = __shfl(node, l);
          if (o == 1) {
            const int at = hl_i32(hr.pos + nd);
            if (at >= 0) (void)heap_pop_at(hr, at);
          } else {
            double p[6], r[6];
            for (int q = 0; q < 6; ++q) { p[q] = st.pos[6 * (size_t)nd + q]; r[q] = ref[q]; }
            heap_push(hr, nd, dist6(p, r));
          }
        }
      }
    }
  }
  __syncthreads();
  // ---- 3. this heap's size; the last heap through: is every heap empty (src/forest.h:184-191)
  if (tid == 0) {
    *hr.size_p = hr.n;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (hr.n > 0) atomicAdd(&P.counters[1], 1);
    __threadfence();
    if (atomicAdd(&P.counters[0], 1) == P.n_heaps - 1) {
      const int ne = __hip_atomic_load(&P.counters[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      c->prio_all_empty = ne == 0 ? 1 : 0;
      P.counters[0] = 0;
      P.counters[1] = 0;
    }
  }
}

// position map of freshly uploaded heaps
__global__ __launch_bounds__(256) void k_prio_index(PrioView P) {
  const int h = blockIdx.y;
  const int n = P.size[h];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) P.pos[(size_t)h * P.cap + P.v[(size_t)h * P.cap + i]] = i;
}

void launch_prio_begin(hipStream_t s, const DevForestView& f) {
  hipLaunchKernelGGL(k_prio_plan, dim3(1), dim3(1024), 0, s, f);
  hipLaunchKernelGGL(k_prio_begin, dim3(1), dim3(64), 0, s, f);
  hipLaunchKernelGGL(k_prio_pops, dim3(f.prio.n_heaps), dim3(64), 0, s, f);
}
void launch_prio_end(hipStream_t s, const DevForestView& f, const NodeStoreView& st) {
  hipLaunchKernelGGL(k_prio_end, dim3(f.prio.n_heaps), dim3(256), 0, s, f, st);
}
void launch_prio_index(hipStream_t s, const PrioView& p) {
  if (p.n_heaps <= 0) return;
  hipLaunchKernelGGL(k_prio_index, dim3(64, p.n_heaps), dim3(256), 0, s, p);
}

}  // namespace sffk
