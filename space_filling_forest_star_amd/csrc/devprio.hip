// devprio.hip - the priority-frontier mode of SpaceForest::Solve on the device-resident engine (PrioView, kernels.h).
//
//   k_prio_plan   one workgroup of 1024 threads, everything in LDS: the wave's picks - tree, heap, "minimum or random
//                 entry" - exactly as src/forest.h:126-147 draws them (libstdc++ uniform_int = Lemire with rejection,
//                 uniform_real), every slot's start in the engine-word stream found by a scan over per-block functions
//   k_prio_begin  the same one slot after the other by one wavefront (heap sizes counted down as the slots take their
//                 nodes): only when the plan is not valid (a draw in the rejection zone, a heap asked for more than it
//                 holds) or SFFGPU_PRIO_SEQ is set; control block of the wave
//   k_prio_pops   one workgroup per heap: its slots found by all 256 threads, their pops in slot order on the first
//                 wavefront (src/heap.h:175-238: pop / pop at index)
//   k_prio_end    one workgroup per heap, behind the wave's rounds: pushes of the wave's new nodes of the heap's tree in
//                 creation order (src/forest.h:360-363), then per slot of the tree in slot order: an exhausted slot's node
//                 leaves the tree's OTHER heaps (:164-173), a slot that expanded its node puts it back onto the heap it
//                 came from (:178-180); the last heap through says whether every heap is empty (:184-191)
// The order of operations per heap is the reference's; heaps do not see each other, so one workgroup per heap is exact.
// A sift is ONE memory round trip (HeapWin / HeapAnc below) and its walk a ballot: see the comments there.
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

// ---- one heap in HBM, worked on by ONE wavefront (all 64 lanes call these together).
// A launch reads what it has written itself: loads go past the vector L1 (relaxed agent-scope = sc1, served by the L2
// where the stores land); a wavefront's accesses to one address reach the L2 in program order, so an operation sees
// the stores of the one before it without waiting for them.
// What makes a sift slow is one L2 round trip per level.  Both directions fetch their whole neighbourhood at once:
// the sift-down the 62 descendants of the next five levels (lane = position in that sub-heap), the sift-up every
// ancestor up to the root (lane = level) - the walk itself then runs on registers (round 4; 4 us -> ~1 us per pop).
#ifdef SFFK_PRIO_DEBUG
__device__ unsigned long long g_prio_dbg[16];
#define PDBG(i_, val_) do { h.dbg[i_] += (long long)(val_); } while (0)
#define PCLK() ((long long)wall_clock64())
#else
#define PDBG(i_, val_) do {} while (0)
#define PCLK() 0LL
#endif
struct HeapRef {
  int32_t* v; double* key; int32_t* pos; int32_t* size_p;
  int n;
#ifdef SFFK_PRIO_DEBUG
  long long dbg[16];
#endif
};
__device__ __forceinline__ int hl_i32(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double hl_f64(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void heap_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// Keys are distances (>= +0, never NaN): their bit patterns order like unsigned integers, so the walks below compare
// and carry them as 64-bit integers in SGPRs - every value of a walk is the same in all lanes, and written this way
// (readfirstlane / readlane) the compiler keeps the whole walk on the scalar unit with uniform branches; as "divergent"
// vector code a level cost ~400 cycles of exec-mask bookkeeping (1 us per five levels, measured).
typedef unsigned long long hkey_t;
__device__ __forceinline__ hkey_t hk_bits(double k) { return (hkey_t)__double_as_longlong(k); }
__device__ __forceinline__ int uni_i32(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ hkey_t uni_u64(hkey_t x) {
  const unsigned int lo = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(x & 0xffffffffULL));
  const unsigned int hi = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(x >> 32));
  return ((hkey_t)hi << 32) | (hkey_t)lo;
}
__device__ __forceinline__ int lane_i32(int x, int src) { return __builtin_amdgcn_readlane(x, src); }
__device__ __forceinline__ hkey_t lane_u64(hkey_t x, int src) {
  const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(x & 0xffffffffULL), src);
  const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(x >> 32), src);
  return ((hkey_t)hi << 32) | (hkey_t)lo;
}
// lane ^ 1 (DPP quad_perm [1,0,3,2]: a VALU move, not a trip through the LDS crossbar)
__device__ __forceinline__ int dpp_xor1(int x) { return __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true); }
__device__ __forceinline__ hkey_t hl_key(const double* p) {
  return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void heap_put(const HeapRef& h, int i, int node, hkey_t k) {
  h.v[i] = node; reinterpret_cast<unsigned long long*>(h.key)[i] = k; h.pos[node] = i;
}
// the same entry from every lane: one lane stores it (64 lanes on one address are not merged into one request)
__device__ __forceinline__ void heap_put1(const HeapRef& h, int i, int node, hkey_t k) { if ((threadIdx.x & 63) == 0) heap_put(h, i, node, k); }
// ---- the two sifts, each split into "ask for everything it can need" and "decide", so that an operation asks for its
// own entries, the ancestors and the sub-heap below in ONE round trip.
// One wavefront alone issues an instruction every 5-8 cycles, so a walk that compares level after level costs ~250
// cycles per level even as scalar code (measured: 1.9 us for 15 levels with all data in registers).  Instead every lane
// decides for its own entry, all at once.
struct HeapWin { hkey_t kq; int vq; int at; bool have; };    // lane q (2..63) = q-th entry (1-based, level order) below `index`
__device__ __forceinline__ HeapWin win_loads(const HeapRef& h, int index, int n) {
  const int lane = threadIdx.x & 63;
  const int d = 31 - __clz(lane | 1);
  const long long at64 = (((long long)index + 1) << d) - 1 + (lane - (1 << d));
  HeapWin w;
  w.have = lane >= 2 && at64 < n;
  w.at = (int)at64;
  w.kq = w.have ? hl_key(h.key + w.at) : 0ULL;
  w.vq = w.have ? hl_i32(h.v + w.at) : 0;
  return w;
}
// Heap::BubbleDown of (node, k) standing at `index` in a heap of n entries (src/heap.h:122-149: the smaller child, ties
// to the left), w = the window below index.  "I am the smaller child of my parent" (the sibling's key comes from the
// neighbouring lane) "and k is larger than my key" - the reference's two comparisons say exactly that: the element
// moves down to the smaller child, the left one on a tie, while k is larger than that child's key.  The ballot of these
// bits holds the whole path through the window; following it is a shift and a test per level, and the entries on the
// path move up with ONE store per array.
__device__ void down_finish(HeapRef& h, int index, int node, hkey_t k, int n, HeapWin w) {
  const int lane = threadIdx.x & 63;
  while (true) {
    if (2 * (long long)index + 1 >= n) break;
    const hkey_t ks = ((hkey_t)(unsigned int)dpp_xor1((int)(unsigned int)(w.kq >> 32)) << 32) | (hkey_t)(unsigned int)dpp_xor1((int)(unsigned int)(w.kq & 0xffffffffULL));
    const bool sib_have = dpp_xor1(w.have ? 1 : 0) != 0;
    const bool left = (lane & 1) == 0;
    const bool smaller = w.have && (left ? !(sib_have && ks < w.kq) : (w.kq < ks));
    const unsigned long long mv = __ballot(smaller && k > w.kq);   // the element would move down INTO this entry's place
    unsigned long long path = 0ULL;
    int q = 1;
    while (q < 32) {
      const unsigned int two = (unsigned int)(mv >> (2 * q)) & 3u;
      if (!two) break;
      q = 2 * q + (int)(two >> 1);
      path |= 1ULL << q;
    }
    if ((path >> lane) & 1ULL) heap_put(h, (w.at - 1) >> 1, w.vq, w.kq);   // every entry on the path: one level up
    if (q > 1) index = lane_i32(w.at, q);
    if (q < 32) break;
    w = win_loads(h, index, n);
  }
  heap_put1(h, index, node, k);
}
struct HeapAnc { hkey_t kp; int vp; bool have; };            // lane j = the (j + 1)-th ancestor of `index`
__device__ __forceinline__ HeapAnc anc_loads(const HeapRef& h, int index) {
  const int lane = threadIdx.x & 63;
  const unsigned int i1 = (unsigned int)index + 1u;        // 1-based: the j-th ancestor is (i1 >> j) - 1
  const unsigned int mine = lane < 31 ? (i1 >> (lane + 1)) : 0u;
  HeapAnc a;
  a.have = mine != 0u;
  a.kp = a.have ? hl_key(h.key + (mine - 1u)) : 0ULL;
  a.vp = a.have ? hl_i32(h.v + (mine - 1u)) : 0;
  return a;
}
// Heap::BubbleUp of (node, k) standing at `index` (src/heap.h:151-163)
__device__ void up_finish(HeapRef& h, int index, int node, hkey_t k, HeapAnc a) {
  const int lane = threadIdx.x & 63;
  const unsigned int i1 = (unsigned int)index + 1u;
  const unsigned long long stop = __ballot(!(a.have && a.kp > k));   // (lane 31 always stops)
  const int up = __ffsll((long long)stop) - 1;               // ancestors 1 .. up move one level down
  if (lane < up) heap_put(h, (int)(i1 >> lane) - 1, a.vp, a.kp);
  heap_put1(h, (int)(i1 >> up) - 1, node, k);
}
__device__ int heap_pop(HeapRef& h) {                       // Heap::pop()
  [[maybe_unused]] const long long t_a = PCLK();
  const int size = uni_i32(h.n);
  const int last_v = hl_i32(h.v + size - 1);
  const hkey_t last_k = hl_key(h.key + size - 1);
  const int root = hl_i32(h.v);
  const HeapWin w = win_loads(h, 0, size - 1);
  const int mn = uni_i32(root);
  if ((threadIdx.x & 63) == 0) h.pos[mn] = -1;
  h.n = size - 1;
  if (size > 1) down_finish(h, 0, uni_i32(last_v), uni_u64(last_k), size - 1, w);
  PDBG(0, 1); PDBG(1, PCLK() - t_a);
  return mn;
}
// Heap::pop(index), id < size
struct HeapAt { int val, last; hkey_t old_cost, new_cost; HeapAnc a; HeapWin w; };
__device__ __forceinline__ HeapAt pop_at_loads(const HeapRef& h, int id, int size) {
  HeapAt L;
  L.last = hl_i32(h.v + size - 1);
  L.new_cost = hl_key(h.key + size - 1);
  L.val = hl_i32(h.v + id);
  L.old_cost = hl_key(h.key + id);
  L.a = anc_loads(h, id);
  L.w = win_loads(h, id, size - 1);
  return L;
}
__device__ int pop_at_finish(HeapRef& h, int id, int size, const HeapAt& L) {
  const int val = uni_i32(L.val);
  if ((threadIdx.x & 63) == 0) h.pos[val] = -1;
  h.n = size - 1;
  if (id != size - 1) {
    const hkey_t nc = uni_u64(L.new_cost), oc = uni_u64(L.old_cost);
    const int last = uni_i32(L.last);
    if (nc < oc) up_finish(h, id, last, nc, L.a); else down_finish(h, id, last, nc, size - 1, L.w);
  }
  return val;
}
__device__ int heap_pop_at(HeapRef& h, int id) {
  id = uni_i32(id);
  const int size = uni_i32(h.n);
  if (id >= size) return -1;
  const HeapAt L = pop_at_loads(h, id, size);
  return pop_at_finish(h, id, size, L);
}
// `node` leaves the heap (src/forest.h:164-173).  hint = where the position map had it when the chunk was gathered: the
// map is asked again, together with everything a removal at `hint` needs - one round trip when the hint still holds
__device__ void heap_remove(HeapRef& h, int node, int hint) {
  node = uni_i32(node); hint = uni_i32(hint);
  const int size = uni_i32(h.n);
  const int now_v = hl_i32(h.pos + node);
  if (hint >= 0 && hint < size) {
    const HeapAt L = pop_at_loads(h, hint, size);
    const int now = uni_i32(now_v);
    if (now == hint) { (void)pop_at_finish(h, hint, size, L); return; }
    if (now >= 0) (void)heap_pop_at(h, now);
    return;
  }
  const int now = uni_i32(now_v);
  if (now >= 0) (void)heap_pop_at(h, now);
}
__device__ void heap_push(HeapRef& h, int node, hkey_t k) {  // Heap::push
  const int at = uni_i32(h.n);
  h.n = at + 1;
  const HeapAnc a = anc_loads(h, at);
  up_finish(h, at, uni_i32(node), uni_u64(k), a);
}
__device__ __forceinline__ int32_t* prio_act_now(const DevForestView& f) { return f.ctrl->act_sel ? f.act_slot2 : f.act_slot; }
__device__ __forceinline__ HeapRef heap_of(const PrioView& P, int h) {
  HeapRef r;
  r.v = P.v + (size_t)h * P.cap; r.key = P.key + (size_t)h * P.cap; r.pos = P.pos + (size_t)h * P.cap;
  r.size_p = P.size + h;
  r.n = uni_i32(P.size[h]);
#ifdef SFFK_PRIO_DEBUG
  for (int i = 0; i < 16; ++i) r.dbg[i] = 0;
#endif
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
//   coin(p) for EVERY stream position p (one bit in LDS), i.e. a slot that starts at p is 3 + coin(p + 2) words long;
//   the stream cut into 1024 blocks, one per thread: a slot never jumps over a block (blocks are >= 4 positions), it
//   enters a block at one of its first four positions - per block and entry: where the chain leaves it and how many slots
//   start inside; an inclusive scan over the blocks (composition of these 4-entry functions) gives every block the entry
//   and the index of its first slot of the chain that starts at position 0; every thread then walks its own block and
//   writes its slots (round 4; the first version built log2(wave) jump tables in global memory: 200 us per wave);
//   the slot's tree / heap / coin come from its words; the random entry's index is drawn by k_prio_pops, which knows the
//   heap's size at that slot's turn (it runs the heap's slots in order);
//   the plan is valid when no draw fell into Lemire's rejection zone and no heap is asked for more nodes than it holds
//   (then no tree or heap was found empty, so the reference would not have redrawn either) - otherwise the control block
//   is left alone and k_prio_begin, launched behind this kernel, does the wave one slot after the other.
#define PP_COIN_WORDS ((4 * 64 * SFFK_DEV_MAX_GROUPS + 4096) / 32)
__global__ __launch_bounds__(1024) void k_prio_plan(DevForestView f) {
  __shared__ int s_sz[SFFK_PRIO_MAX_HEAPS];
  __shared__ int s_cnt[SFFK_PRIO_MAX_HEAPS];
  __shared__ unsigned int s_coin[PP_COIN_WORDS];
  __shared__ int s_fn[1024][4];                  // (slots started << 2) | position in the next block, per entry 0..3
  __shared__ int s_bad, s_pool, s_used;
  DevCtrl* c = f.ctrl;
  const PrioView& P = f.prio;
  const int tid = threadIdx.x, lane = tid & 63;
  if (c->halt || c->in_wave || !P.plan) return;
  const int H = P.n_heaps, T = f.n_trees;
  for (int h = tid; h < H; h += 1024) { s_sz[h] = P.size[h]; s_cnt[h] = 0; }
  if (tid == 0) { s_bad = 0; s_pool = 0; s_used = -1; }
  __syncthreads();
  if (tid < T && P.base[tid + 1] > P.base[tid]) atomicAdd(&s_pool, s_sz[P.base[tid]]);   // src/forest.h:128-131
  __syncthreads();
  int n_slots = f.wave < s_pool ? f.wave : s_pool;
  if (n_slots < 1) n_slots = 1;
  const int M = 4 * n_slots + 8;                 // stream positions a wave without redraws can reach (+ the coins ahead)
  int bk = (M + 1023) / 1024;
  if (bk < 4) bk = 4;
  const int span = 1024 * bk + 64;               // coin bits the walks below may look at
  const unsigned long long cur = c->cursor;
  // ---- coins: position = thread (coalesced ring reads, eight passes' words asked for together), 64 bits per wavefront and pass
  for (int p0 = 0; p0 < span; p0 += 8 * 1024) {
    unsigned long long w[8];
    for (int j = 0; j < 8; ++j) {
      const int p = p0 + j * 1024 + tid;
      w[j] = f.ring[(cur + (unsigned long long)(p < M ? p : M - 1)) & f.ring_mask];
    }
    for (int j = 0; j < 8; ++j) {
      const int p = p0 + j * 1024 + tid;
      const bool coin = p < M && !(uniform_real(w[j], 0.0, 1.0) <= P.bias);
      const unsigned long long m = __ballot(coin);
      if (lane == 0 && p < span && (p >> 5) + 1 < PP_COIN_WORDS) { s_coin[p >> 5] = (unsigned int)m; s_coin[(p >> 5) + 1] = (unsigned int)(m >> 32); }
    }
  }
  __syncthreads();
  auto hop = [&](int p) -> int { return p + 3 + (int)((s_coin[(p + 2) >> 5] >> ((p + 2) & 31)) & 1u); };
  // ---- this block's function
  const int lo = tid * bk, hi = lo + bk;
  for (int e = 0; e < 4; ++e) {
    int p = lo + e, n = 0;
    while (p < hi) { p = hop(p); ++n; }
    s_fn[tid][e] = (n << 2) | (p - hi);
  }
  __syncthreads();
  // ---- inclusive scan: s_fn[.][b] = block b after blocks b-1, b-2, ... (Hillis-Steele, 10 steps)
  for (int d = 1; d < 1024; d <<= 1) {
    int g[4];
    for (int e = 0; e < 4; ++e) g[e] = s_fn[tid][e];
    if (tid >= d) {
      int r[4];
      for (int e = 0; e < 4; ++e) {
        const int first = s_fn[tid - d][e];                 // the lower blocks, entered at e
        const int then = g[first & 3];                      // this run of blocks, entered where they leave
        r[e] = (((first >> 2) + (then >> 2)) << 2) | (then & 3);
      }
      for (int e = 0; e < 4; ++e) g[e] = r[e];
    }
    __syncthreads();
    for (int e = 0; e < 4; ++e) s_fn[tid][e] = g[e];
    __syncthreads();
  }
  // ---- my slots (the chain enters block 0 at position 0)
  int p = lo, s = 0;
  if (tid > 0) { const int before = s_fn[tid - 1][0]; p = lo + (before & 3); s = before >> 2; }
  int32_t* act = prio_act_now(f);
  while (p < hi && s <= n_slots) {
    if (s == n_slots) { s_used = p; break; }
    const unsigned long long w0 = f.ring[(cur + (unsigned long long)p) & f.ring_mask];
    const unsigned long long w1 = f.ring[(cur + (unsigned long long)p + 1ULL) & f.ring_mask];
    const int nx = hop(p);
    bool bad = false;
    int t = prio_lemire(w0, (unsigned long long)T), hp = 0, idx = -1;
    unsigned long long w3 = 0ULL;
    if (t < 0) { bad = true; t = 0; }
    else {
      const int b0 = P.base[t], nh = P.base[t + 1] - b0;
      hp = nh > 0 ? prio_lemire(w1, (unsigned long long)nh) : -1;
      if (hp < 0) { bad = true; hp = 0; }
      else {
        atomicAdd(&s_cnt[b0 + hp], 1);
        if (nx - p == 4) { idx = -2; w3 = f.ring[(cur + (unsigned long long)p + 3ULL) & f.ring_mask]; }
      }
    }
    if (bad) s_bad = 1;
    P.slot_tree[s] = t; P.slot_heap[s] = hp; P.slot_idx[s] = idx; P.slot_word[s] = w3; act[s] = s;
    p = nx; ++s;
  }
  __syncthreads();
  for (int h = tid; h < H; h += 1024) if (s_cnt[h] > s_sz[h]) s_bad = 1;
  __syncthreads();
  if (s_bad || s_used < 0 || tid != 0) return;   // (not valid: k_prio_begin does the wave)
  c->compact_from = 0;
  c->app_n = 0;
  c->cursor = cur + (unsigned long long)s_used;
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

// ------------------------------------------------------------------ ordered gathers for the per-heap kernels
// The per-heap kernels run a heap's operations one after the other on one wavefront; FINDING them (which of the wave's
// slots / new nodes belong to this heap, their keys) is done by the whole workgroup, 2048 candidates at a time - eight
// consecutive ones per thread, so that every dependent load level is one round trip per chunk and not one per 64 slots.
#define PG_CHUNK 2048
// exclusive prefix of v over the 256 threads of the workgroup (s_w: 4 ints); *total = the sum
__device__ __forceinline__ int block_excl_scan_256(int v, int* s_w, int* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int x = v;
  for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d); if (lane >= d) x += y; }
  __syncthreads();                               // (s_w may still be read from the previous call)
  if (lane == 63) s_w[w] = x;
  __syncthreads();
  int before = 0, sum = 0;
  for (int i = 0; i < 4; ++i) { if (i < w) before += s_w[i]; sum += s_w[i]; }
  *total = sum;
  return before + x - v;
}

// ------------------------------------------------------------------ the pops of one heap, in slot order
__global__ __launch_bounds__(256) void k_prio_pops(DevForestView f) {
  __shared__ int s_slot[PG_CHUNK];
  __shared__ int s_idx[PG_CHUNK];
  __shared__ unsigned long long s_word[PG_CHUNK];
  __shared__ int s_w[4];
  const DevCtrl* c = f.ctrl;
  const PrioView& P = f.prio;
  const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  if (!c->prio_wave || !c->in_wave || P.gen[h] == c->prio_gen) return;      // (once per wave, right behind k_prio_begin)
  int t = 0;
  while (P.base[t + 1] <= h) ++t;
  const int hp = h - P.base[t];
  const int n_slots = c->n_slots;
  HeapRef hr = heap_of(P, h);
  for (int c0 = 0; c0 < n_slots; c0 += PG_CHUNK) {
    int st_[8], sh_[8];
    for (int j = 0; j < 8; ++j) {
      const int s = c0 + tid * 8 + j;
      const int sc = s < n_slots ? s : n_slots - 1;    // (clamped, not branched: the sixteen loads go out together)
      const int a_ = P.slot_tree[sc], b_ = P.slot_heap[sc];
      st_[j] = s < n_slots ? a_ : -1;
      sh_[j] = s < n_slots ? b_ : -1;
    }
    int cnt = 0;
    for (int j = 0; j < 8; ++j) cnt += (st_[j] == t && sh_[j] == hp) ? 1 : 0;
    int total;
    int at = block_excl_scan_256(cnt, s_w, &total);
    for (int j = 0; j < 8; ++j)
      if (st_[j] == t && sh_[j] == hp) {
        const int s = c0 + tid * 8 + j;
        const int idx = P.slot_idx[s];
        s_slot[at] = s; s_idx[at] = idx; s_word[at] = idx == -2 ? P.slot_word[s] : 0ULL;
        ++at;
      }
    __syncthreads();
    if (tid < 64) {
      for (int e = 0; e < total; ++e) {
        int id = uni_i32(s_idx[e]);
        if (id == -2) {   // (planned in parallel: the random entry's index is drawn here, with the heap's size at this turn)
          id = prio_lemire(uni_u64(s_word[e]), (unsigned long long)hr.n);
          if (id < 0) { if (lane == 0) f.ctrl->fault = SFFK_FAULT_PRIO_REDRAW; id = 0; }
        }
        const int node = id < 0 ? heap_pop(hr) : heap_pop_at(hr, id);
        if (lane == 0) f.slot_node[s_slot[e]] = node;
      }
    }
    __syncthreads();
  }
  if (tid == 0) { *hr.size_p = hr.n; P.gen[h] = c->prio_gen; }
#ifdef SFFK_PRIO_DEBUG
  if (tid == 0 && h == 0) for (int i = 0; i < 6; ++i) g_prio_dbg[i] += (unsigned long long)hr.dbg[i];
#endif
}

// ------------------------------------------------------------------ the end of the wave, per heap
__global__ __launch_bounds__(256) void k_prio_end(DevForestView f, NodeStoreView st) {
  __shared__ unsigned int s_fail[SFFK_DEV_MAX_GROUPS * 2];   // bit per slot: still failing (exhausted) at the wave's end
  __shared__ int s_op[PG_CHUNK];                 // node, or ~node: "remove it"
  __shared__ hkey_t s_key[PG_CHUNK];
  __shared__ int s_w[4];
  DevCtrl* c = f.ctrl;
  const PrioView& P = f.prio;
  if (c->halt || !c->in_wave || c->n_act != 0) return;       // (the wave is not over: a fault stopped it)
  const int h = blockIdx.x, tid = threadIdx.x;
  int t = 0;
  while (P.base[t + 1] <= h) ++t;
  const int hp = h - P.base[t];
  double ref[6];
  for (int q = 0; q < 6; ++q) ref[q] = P.ref[6 * (size_t)h + q];
  HeapRef hr = heap_of(P, h);
  auto key_of = [&](int node) -> double {                    // Distance(node, refPoint)
    double p[6];
    for (int q = 0; q < 6; ++q) p[q] = st.pos[6 * (size_t)node + q];
    return dist6(p, ref);
  };
  // ---- 1. the wave's new nodes of this tree, in creation order (src/forest.h:360-363)
  const int n0 = c->prio_n0, n1 = c->n_nodes;
#define h hr
  for (int c0 = n0; c0 < n1; c0 += PG_CHUNK) {
    [[maybe_unused]] const long long t_g = PCLK();
    bool mine[8];
    int cnt = 0;
    for (int j = 0; j < 8; ++j) {
      const int id = c0 + tid * 8 + j;
      mine[j] = id < n1 && st.tree[id] == t;
      cnt += mine[j] ? 1 : 0;
    }
    double k[8];
    for (int j = 0; j < 8; ++j) k[j] = mine[j] ? key_of(c0 + tid * 8 + j) : 0.0;
    int total;
    int at = block_excl_scan_256(cnt, s_w, &total);
    for (int j = 0; j < 8; ++j)
      if (mine[j]) { s_op[at] = c0 + tid * 8 + j; s_key[at] = hk_bits(k[j]); ++at; }
    __syncthreads();
    [[maybe_unused]] const long long t_p = PCLK();
    PDBG(6, total); PDBG(7, t_p - t_g);
    if (tid < 64)
      for (int e = 0; e < total; ++e) heap_push(hr, s_op[e], s_key[e]);
    PDBG(8, PCLK() - t_p);
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
    for (int c0 = 0; c0 < n_slots; c0 += PG_CHUNK) {
      [[maybe_unused]] const long long t_g = PCLK();
      int op[8];                                             // 1 = remove the node, 2 = put it back
      int cnt = 0;
      for (int j = 0; j < 8; ++j) {
        const int s = c0 + tid * 8 + j;
        op[j] = 0;
        if (s < n_slots && P.slot_tree[s] == t) {
          const bool failing = (s_fail[s >> 5] >> (s & 31)) & 1u;
          const int sh = P.slot_heap[s];
          if (failing && sh != hp) op[j] = 1;
          else if (!failing && sh == hp) op[j] = 2;
        }
        cnt += op[j] ? 1 : 0;
      }
      int node[8];
      for (int j = 0; j < 8; ++j) node[j] = op[j] ? f.slot_node[c0 + tid * 8 + j] : 0;
      hkey_t k[8];                                           // put back: the key; remove: where the node stands (a hint)
      for (int j = 0; j < 8; ++j) k[j] = op[j] == 2 ? hk_bits(key_of(node[j])) : op[j] == 1 ? (hkey_t)(unsigned int)hr.pos[node[j]] : 0ULL;
      int total;
      int at = block_excl_scan_256(cnt, s_w, &total);
      for (int j = 0; j < 8; ++j)
        if (op[j]) { s_op[at] = op[j] == 1 ? ~node[j] : node[j]; s_key[at] = k[j]; ++at; }
      __syncthreads();
      [[maybe_unused]] const long long t_p = PCLK();
      PDBG(9, total); PDBG(10, t_p - t_g);
      if (tid < 64) {
        for (int e = 0; e < total; ++e) {
          const int o = uni_i32(s_op[e]);
          if (o < 0) {
            PDBG(12, 1);
            heap_remove(hr, ~o, (int)(unsigned int)uni_u64(s_key[e]));
          } else {
            heap_push(hr, o, s_key[e]);
          }
        }
      }
      PDBG(11, PCLK() - t_p);
      __syncthreads();
    }
  }
#undef h
#ifdef SFFK_PRIO_DEBUG
  if (tid == 0 && blockIdx.x == 0) for (int i = 6; i < 16; ++i) g_prio_dbg[i] += (unsigned long long)hr.dbg[i];
#endif
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

#ifdef SFFK_PRIO_DEBUG
void debug_counters_prio(unsigned long long* out16) { (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_prio_dbg), sizeof(unsigned long long) * 16); }
#else
void debug_counters_prio(unsigned long long* out16) { for (int i = 0; i < 16; ++i) out16[i] = 0ULL; }
#endif
// position map of freshly uploaded heaps
__global__ __launch_bounds__(256) void k_prio_index(PrioView P) {
  const int h = blockIdx.y;
  const int n = P.size[h];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) P.pos[(size_t)h * P.cap + P.v[(size_t)h * P.cap + i]] = i;
}

void launch_prio_begin(hipStream_t s, const DevForestView& f) {
  hipLaunchKernelGGL(k_prio_plan, dim3(1), dim3(1024), 0, s, f);
  hipLaunchKernelGGL(k_prio_begin, dim3(1), dim3(64), 0, s, f);
  hipLaunchKernelGGL(k_prio_pops, dim3(f.prio.n_heaps), dim3(256), 0, s, f);
}
void launch_prio_end(hipStream_t s, const DevForestView& f, const NodeStoreView& st) {
  hipLaunchKernelGGL(k_prio_end, dim3(f.prio.n_heaps), dim3(256), 0, s, f, st);
}
void launch_prio_index(hipStream_t s, const PrioView& p) {
  if (p.n_heaps <= 0) return;
  hipLaunchKernelGGL(k_prio_index, dim3(64, p.n_heaps), dim3(256), 0, s, p);
}

}  // namespace sffk
