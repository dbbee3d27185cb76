// devstar.hip — SFF* (optimize = true) on the device-resident engine: choose-parent + rewire of
// SpaceForest::expandNode (src/forest.h:307-351) for the samples a round accepts (gfx950).
//
// What is sequential in the reference and how it is kept: sample i of a round
//   * takes k = floor(2e log10(#nodes)) with the nodes accepted before it counted in (:309),
//   * looks its k nearest up among the store AND those earlier samples (:317),
//   * reads each member's DistanceToRoot as the earlier samples' rewires left it (:322, :337),
//   * rewires members itself (:336-348) - a later sample of the round may rewire the same node again.
// Acceptance (who becomes a node, with which id) does not depend on any cost, so k_decide / k_resolve settle it first,
// exactly as for plain SFF.  The costs are then the unique fixed point of
//     view(i, x)  = proposal of the latest j < i whose rewire of x is active, else cost(x)            (per-node lists)
//     result(i)   = choose-parent / rewire of sample i evaluated on its views                          (one wavefront)
// and because j < i always, iterating "every sample recomputes" reaches it in (longest dependency chain + 1) passes
// whatever the order inside a pass; a pass that writes nothing proves it (kernel boundaries make a pass's writes visible
// to the next).  Everything is evaluated in the host engine's expression order (-ffp-contract=off): bit-identical.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"
#include "sff_geom.h"
#include "kernels_dev.h"

namespace sffk {

using namespace sffg;

#define STAR_INF __longlong_as_double(0x7ff0000000000000LL)
#define STAR_FAULT 4   // index in StarView::hdr

__device__ __forceinline__ bool star_accepted(const DevForestView& f, int i, int& rank) {
  const unsigned long long w = f.w_acc[i >> 6];
  rank = f.acc_pref[i >> 6] + __popcll(w & ((1ULL << (i & 63)) - 1ULL));
  return (w >> (i & 63)) & 1ULL;
}

// candidates of a group of up to 64 cells (lane = cell, m = its item count), flattened over the lanes.
// store = true: the node grid (permanent nodes of the sample's tree); false: the round's own grid - samples accepted
// EARLIER in the round (temporary id < self), of the same tree, not farther than `limit`
__device__ __forceinline__ void star_cells(const GridView& g, int m, int cell, int lane, const double* qp, int tree, bool store,
                                           int N0, int Tb, int self, double limit, const DevForestView& f, TopK& t, int k,
                                           int& have) {
  int inc = m;
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(inc, off);
    if (lane >= off) inc += o;
  }
  const int total = __shfl(inc, 63);
  for (int base = 0; base < total; base += 64) {
    const int j = base + lane;
    const int jj = j < total ? j : total - 1;
    int lo = 0, hi = 63;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (__shfl(inc, mid) > jj) hi = mid; else lo = mid + 1;
    }
    const int src_cell = __shfl(cell, lo);
    const int slot = jj - (__shfl(inc, lo) - __shfl(m, lo));
    bool cand = false;
    double d = 1.0e300;
    int id = 0x7fffffff;
    if (j < total) {
      const GridItem it = g.items[(size_t)src_cell * g.bk + slot];
      id = it.id;
      if (it.tree == tree) {
        if (store) cand = id < N0;
        else {
          int rk;
          cand = id >= Tb && id < self && star_accepted(f, id - Tb, rk);
        }
        if (cand) d = dist6(it.p, qp);
        if (cand && !store) cand = d <= limit;
      }
    }
    const double worst = topk_worst(t, k, have);
    cand = cand && (have < k || key_less(d, id, worst, 0x7fffffff));
    topk_insert(t, lane, k, have, __ballot(cand), d, id);
  }
}

// ------------------------------------------------------------------ k nearest + member edges + toucher lists
__global__ __launch_bounds__(256) void k_star_knn(ResolveArgs A, GridView g, GridView tg, NodeStoreView st, double cell_edge,
                                                  double slack) {
  const DevForestView& f = A.f;
  const StarView& S = A.S;
  const DevCtrl* c = f.ctrl;
  const int n = c->app_n;
  if (blockIdx.x == 0 && threadIdx.x < 32) S.ectrl[threadIdx.x] = 0;                       // the member-edge pipeline's block
  if (blockIdx.x == 0 && threadIdx.x < SFFK_STAR_PASSES) S.changed[threadIdx.x] = 0;
  if (n <= 0) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + wave;
  if (i >= n) return;
  int r;
  if (!star_accepted(f, i, r)) return;
  const int N0 = c->app_N0, Tb = f.temp_base;
  const unsigned ep = (unsigned)c->epoch;
  // k = (size_t)(2e log10(#nodes)) with the nodes accepted before this sample counted in (src/forest.h:309)
  const int Nn = N0 + r;
  const int k_ref = __popcll(__ballot(lane > 0 && lane <= SFFK_STAR_KMAX + 1 && S.ktab[lane] <= Nn));
  const int self = Tb + i;
  const int mine = st.tree[self];
  double qp[6];
  for (int q = 0; q < 6; ++q) qp[q] = A.newpos[6 * (size_t)i + q];
  const int tcnt = S.tree_cnt[16 * mine];
  if (k_ref > SFFK_STAR_KMAX) {   // (a node count beyond what the member slots are sized for: host path)
    if (lane == 0) atomicOr(S.hdr + STAR_FAULT, 1);
    return;
  }
  const int k = k_ref;
  const int k_store = k < tcnt ? k : tcnt;
  TopK t{1.0e300, 0x7fffffff};
  int have = 0;
  if (k > 0) {
    const int cx = grid_coord((float)qp[0], g.ox, g.inv_cell, g.nx), cy = grid_coord((float)qp[1], g.oy, g.inv_cell, g.ny),
              cz = grid_coord((float)qp[2], g.oz, g.inv_cell, g.nz);
    {   // the node grid's shared overflow list first (usually empty)
      int no = g.ovf_cnt[0];
      if (no > g.ovf_cap) no = g.ovf_cap;
      for (int base = 0; base < no; base += 64) {
        const int j = base + lane;
        bool cand = false;
        double d = 1.0e300;
        int id = 0x7fffffff;
        const double worst = topk_worst(t, k, have);
        if (j < no) {
          const GridItem it = g.ovf[j];
          id = it.id;
          if (id < N0 && it.tree == mine) {
            d = dist6(it.p, qp);
            cand = have < k || key_less(d, id, worst, 0x7fffffff);
          }
        }
        topk_insert(t, lane, k, have, __ballot(cand), d, id);
      }
    }
    // shells of cells around the sample's cell; after shell r every node within r * cell_edge (minus the fp32 slack
    // of the cell assignment) has been seen.  A tree with fewer than k nodes is complete as soon as all of them are in.
    const int rmax = max(max(g.nx, g.ny), g.nz);
    for (int rr = 0; rr <= rmax; ++rr) {
      if (have >= k_store && k_store == tcnt) break;      // the whole tree is in the list
      const int w = 2 * rr + 1;
      const int total = w * w * w;
      for (int c0 = 0; c0 < total; c0 += 64) {
        const int cc = c0 + lane;
        int cell = 0, m = 0;
        if (cc < total) {
          const int ox = cc % w - rr, oy = (cc / w) % w - rr, oz = cc / (w * w) - rr;
          const bool shell = ox == -rr || ox == rr || oy == -rr || oy == rr || oz == -rr || oz == rr;
          const int x = cx + ox, y = cy + oy, z = cz + oz;
          if (shell && x >= 0 && x < g.nx && y >= 0 && y < g.ny && z >= 0 && z < g.nz) {
            cell = (z * g.ny + y) * g.nx + x;
            m = g.cnt[cell];
            if (m > g.bk) m = g.bk;
          }
        }
        if (__any(m > 0)) star_cells(g, m, cell, lane, qp, mine, true, N0, Tb, self, 0.0, f, t, k, have);
      }
      const double covered = (double)rr * cell_edge - slack;
      if (have >= k && topk_worst(t, k, have) <= covered) break;
      if (cx - rr <= 0 && cy - rr <= 0 && cz - rr <= 0 && cx + rr >= g.nx - 1 && cy + rr >= g.ny - 1 && cz + rr >= g.nz - 1) break;
    }
    // the samples accepted earlier in this round (same tree): not farther than the k-th store node, or - while the
    // store holds fewer than k nodes of the tree - all of them (read straight from the temporary store entries)
    if (i > 0) {
      const bool all = have < k;
      if (all) {
        for (int base = 0; base < i; base += 64) {
          const int j = base + lane;
          bool cand = false;
          double d = 1.0e300;
          int rk;
          if (j < i && star_accepted(f, j, rk) && st.tree[Tb + j] == mine) {
            double mp[6];
            for (int q = 0; q < 6; ++q) mp[q] = st.pos[6 * (size_t)(Tb + j) + q];
            d = dist6(mp, qp);
            cand = true;
          }
          const double worst = topk_worst(t, k, have);
          cand = cand && (have < k || key_less(d, Tb + j, worst, 0x7fffffff));
          topk_insert(t, lane, k, have, __ballot(cand), d, Tb + j);
        }
      } else {
        const double limit = topk_worst(t, k, have);
        int rr = (int)((limit + slack) / cell_edge) + 1;
        if (rr > rmax) rr = rmax;
        const int w = 2 * rr + 1;
        const int total = w * w * w;
        for (int c0 = 0; c0 < total; c0 += 64) {
          const int cc = c0 + lane;
          int cell = 0, m = 0;
          if (cc < total) {
            const int x = cx + cc % w - rr, y = cy + (cc / w) % w - rr, z = cz + cc / (w * w) - rr;
            if (x >= 0 && x < g.nx && y >= 0 && y < g.ny && z >= 0 && z < g.nz) {
              cell = (z * g.ny + y) * g.nx + x;
              const bool maybe = tg.occ ? ((tg.occ[cell >> 5] >> (cell & 31)) & 1u) != 0 : true;
              if (maybe) { m = tg.cnt[cell]; if (m > tg.bk) m = tg.bk; }
            }
          }
          if (__any(m > 0)) star_cells(tg, m, cell, lane, qp, mine, false, N0, Tb, self, limit, f, t, k, have);
        }
        int no = tg.ovf_cnt[0];
        if (no > tg.ovf_cap) no = tg.ovf_cap;
        for (int base = 0; base < no; base += 64) {
          const int j = base + lane;
          bool cand = false;
          double d = 1.0e300;
          int id = 0x7fffffff;
          const double worst = topk_worst(t, k, have);
          if (j < no) {
            const GridItem it = tg.ovf[j];
            id = it.id;
            int rk;
            if (it.tree == mine && id >= Tb && id < self && star_accepted(f, id - Tb, rk)) {
              d = dist6(it.p, qp);
              cand = d <= limit && (have < k || key_less(d, id, worst, 0x7fffffff));
            }
          }
          topk_insert(t, lane, k, have, __ballot(cand), d, id);
        }
      }
    }
  }
  // ---- the members: ids, distances, toucher lists, the two edge tasks each
  const int cnt = have;
  const bool mem = lane < cnt;
  int node = -1;
  if (mem) {
    if (t.id < N0) node = t.id;
    else { int rk; star_accepted(f, t.id - Tb, rk); node = N0 + rk; }
  }
  const size_t p = (size_t)i * SFFK_STAR_KC + lane;
  S.prop[p] = STAR_INF;
  if (mem) {
    S.m_id[p] = node;
    S.m_d[p] = t.d;
    const unsigned long long mark = ((unsigned long long)ep << 32) | (unsigned long long)(p + 1);
    const unsigned long long old = atomicExch(&S.head[node], mark);
    S.next[p] = (unsigned)(old >> 32) == ep ? (int)(unsigned)(old & 0xffffffffULL) : 0;
  }
  {
    const size_t s0 = ((size_t)r * SFFK_STAR_KC + lane) * 2;
    if (mem) {
      double mp[6];
      for (int q = 0; q < 6; ++q) mp[q] = st.pos[6 * (size_t)t.id + q];
      double* fa = S.seg_a + 6 * s0;
      double* fb = S.seg_b + 6 * s0;
      for (int q = 0; q < 6; ++q) { fa[q] = qp[q]; fb[q] = mp[q]; fa[6 + q] = mp[q]; fb[6 + q] = qp[q]; }   // :323 / :336
      const int ns = edge_samples(edge_parts(qp, mp));
      S.seg_ns[s0] = ns; S.seg_ns[s0 + 1] = edge_samples(edge_parts(mp, qp));
    } else {
      S.seg_ns[s0] = -1; S.seg_ns[s0 + 1] = -1;
    }
    S.first_hit[s0] = 0x7fffffff; S.first_hit[s0 + 1] = 0x7fffffff;
    S.seg_ovf[s0] = 0; S.seg_ovf[s0 + 1] = 0;
  }
  if (lane == 0) {
    S.m_cnt[i] = cnt;
    S.acc_sample[r] = i;
    const int ex = A.parent[i];
    S.best[i] = A.pdist[i] + f.d_root[ex];   // (first guess: the plain SFF cost)
    S.psel[i] = ex;
    S.dcl[i] = A.pdist[i];
    S.cnt[2 * (size_t)i] = 0ULL; S.cnt[2 * (size_t)i + 1] = 0ULL;
  }
}

// DistanceToRoot of node x as sample `i` finds it: the proposal of the latest accepted sample before i whose rewire
// of x is active, else the node's own cost (a node created by this round: its sample's chosen cost)
__device__ __forceinline__ double star_view(const DevForestView& f, const StarView& S, int x, int i, int N0, unsigned ep) {
  const unsigned long long h = S.head[x];
  int q = (unsigned)(h >> 32) == ep ? (int)(unsigned)(h & 0xffffffffULL) : 0;
  int bs = -1;
  double bv = 0;
  for (int guard = 0; q && guard < (1 << 17); ++guard) {   // (a list holds at most one pair per accepted sample)
    const int p = q - 1;
    const int s = p / SFFK_STAR_KC;
    if (s < i && s > bs) {
      const double pr = S.prop[p];
      if (pr < STAR_INF) { bs = s; bv = pr; }
    }
    q = S.next[p];
  }
  if (bs >= 0) return bv;
  return x < N0 ? f.d_root[x] : S.best[S.acc_sample[x - N0]];
}

// ------------------------------------------------------------------ one pass of the fixed point
__global__ __launch_bounds__(256) void k_star_pass(ResolveArgs A, int pass, int sample_blocks) {
  const DevForestView& f = A.f;
  const StarView& S = A.S;
  const DevCtrl* c = f.ctrl;
  if (c->app_n <= 0 || S.hdr[1] || S.hdr[STAR_FAULT]) return;
  if (pass > 0 && S.changed[pass - 1] == 0) return;        // the pass before wrote nothing: fixed point reached
  const int N0 = c->app_N0;
  const unsigned ep = (unsigned)c->epoch;
  if ((int)blockIdx.x >= sample_blocks) {
    // border entries of the round (src/forest.h:288-294): d = cost(neighbour) + cost(expanded) + their distance, the
    // costs as the rejected sample's turn finds them
    const int e = ((int)blockIdx.x - sample_blocks) * 256 + threadIdx.x;
    if (e >= S.hdr[2]) return;
    const int s = S.ev_sample[e];
    const double vn = star_view(f, S, S.ev_nb[e], s, N0, ep), ve = star_view(f, S, S.ev_ex[e], s, N0, ep);
    f.b_dist[S.hdr[3] + e] = vn + ve + S.ev_dist[e];
    return;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + wave;
  if (r >= S.hdr[0]) return;
  const int i = S.acc_sample[r];
  const int cnt = S.m_cnt[i];
  const size_t p = (size_t)i * SFFK_STAR_KC + lane;
  const bool mem = lane < cnt;
  const int ex = A.parent[i];
  const int x = mem ? S.m_id[p] : (lane == cnt ? ex : -1);
  const double d = mem ? S.m_d[p] : 0.0;
  // the member edges' answers: free, and the Collide calls isPathFree makes (early exit at the first hit)
  const size_t s0 = ((size_t)r * SFFK_STAR_KC + lane) * 2;
  int fh_f = 0x7fffffff, fh_b = 0x7fffffff, ns_f = 0, ns_b = 0;
  if (mem) { fh_f = S.first_hit[s0]; fh_b = S.first_hit[s0 + 1]; ns_f = S.seg_ns[s0]; ns_b = S.seg_ns[s0 + 1]; }
  if (__any(mem && (fh_f == 0 || fh_b == 0))) {   // an edge's triangle candidate list ran over: host path
    if (lane == 0) atomicOr(S.hdr + STAR_FAULT, 1);
    return;
  }
  const bool free_f = fh_f == 0x7fffffff, free_b = fh_b == 0x7fffffff;
  const unsigned long long calls_f = free_f ? (unsigned long long)ns_f : (unsigned long long)fh_f;
  const unsigned long long calls_b = free_b ? (unsigned long long)ns_b : (unsigned long long)fh_b;
  const double v = x >= 0 ? star_view(f, S, x, i, N0, ep) : 0.0;
  // ---- choose parent (:320-327): the members in (distance, id) order against the running best
  const double pd = A.pdist[i];
  double best = pd + __shfl(v, cnt);           // dist(new, expanded) + expanded->DistanceToRoot (:308)
  int psel = ex;
  double dcl = pd;
  const double nd = d + v;
  unsigned long long cc = 0, pf = 0;
  int cur = 0;
  while (true) {
    const unsigned long long m = __ballot(mem && lane >= cur && nd < best - SFFG_TOL);
    if (!m) break;
    const int b = __ffsll((long long)m) - 1;
    pf += 1;
    cc += __shfl(calls_f, b);
    if (__shfl((int)free_f, b)) { best = __shfl(nd, b); psel = __shfl(x, b); dcl = __shfl(d, b); }
    cur = b + 1;
  }
  // ---- rewire (:332-350)
  const double proposed = best + d;
  const bool test = mem && proposed < v - SFFG_TOL;
  const bool act = test && free_b;
  pf += (unsigned long long)__popcll(__ballot(test));
  unsigned long long cb = test ? calls_b : 0ULL;
  for (int off = 32; off > 0; off >>= 1) cb += __shfl_xor(cb, off);
  cc += cb;
  const double np = act ? proposed : STAR_INF;
  // ---- write what changed
  bool diff = mem && __double_as_longlong(S.prop[p]) != __double_as_longlong(np);
  if (lane == 0)
    diff |= __double_as_longlong(S.best[i]) != __double_as_longlong(best) || S.psel[i] != psel ||
            __double_as_longlong(S.dcl[i]) != __double_as_longlong(dcl) || S.cnt[2 * (size_t)i] != cc || S.cnt[2 * (size_t)i + 1] != pf;
  if (diff && mem) S.prop[p] = np;
  if (__any(diff)) {
    if (lane == 0) {
      S.best[i] = best; S.psel[i] = psel; S.dcl[i] = dcl;
      S.cnt[2 * (size_t)i] = cc; S.cnt[2 * (size_t)i + 1] = pf;
      S.changed[pass] = 1;
    }
  }
}

// ------------------------------------------------------------------ apply: nodes, rewires, housekeeping
__global__ __launch_bounds__(256) void k_star_apply(ResolveArgs A, GridView tg, int n_bound, int max_passes) {
  const DevForestView& f = A.f;
  const StarView& S = A.S;
  DevCtrl* c = f.ctrl;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + wave;
  // the round's own grid has been read (query kernel, k_star_knn): empty the cells this launch bound's samples used
  if (tg.cnt && i < n_bound && lane == 0) {
    const size_t t = (size_t)f.temp_base + i;
    const float x = A.st.x[t];
    if (x == x) {
      const size_t cell = grid_cell_of(tg, x, A.st.y[t], A.st.z[t]);
      tg.cnt[cell] = 0;
      if (tg.occ) tg.occ[cell >> 5] = 0u;   // (every set bit of the word belongs to a sample of this round)
    }
    if (i == 0) tg.ovf_cnt[0] = 0;
  }
  const int n = c->app_n;
  if (n <= 0 || S.hdr[1]) return;
  // fixed point reached?  (the first pass that wrote nothing; none = not converged within the launches of a round)
  int passes = 0;
  bool conv = false;
  for (int t = 0; t < max_passes; ++t) {
    ++passes;
    if (S.changed[t] == 0) { conv = true; break; }
  }
  if (!conv || S.hdr[STAR_FAULT]) {
    // nothing of this round is kept: the control block goes back to where the round began and the host redoes the
    // round on its unbounded path (k_append finds app_n = 0)
    if (blockIdx.x == 0) {
      __syncthreads();
      for (int w = threadIdx.x; w < (int)(sizeof(DevCtrl) / 4); w += 256)
        reinterpret_cast<int32_t*>(c)[w] = reinterpret_cast<const int32_t*>(S.backup)[w];
    }
    return;
  }
  if (i >= n) return;
  int r;
  if (!star_accepted(f, i, r)) return;
  const int N0 = c->app_N0, fn0 = c->app_fn0;
  const unsigned ep = (unsigned)c->epoch;
  const int id = N0 + r;
  const int cnt = S.m_cnt[i];
  const size_t p = (size_t)i * SFFK_STAR_KC + lane;
  // ---- the members this sample rewires for good: its proposal is active and no later sample's is
  int rewired = 0;
  if (lane < cnt) {
    const double pr = S.prop[p];
    const int x = S.m_id[p];
    if (pr < STAR_INF && x < N0) {
      const unsigned long long h = S.head[x];
      int q = (unsigned)(h >> 32) == ep ? (int)(unsigned)(h & 0xffffffffULL) : 0;
      bool last = true;
      for (int guard = 0; q && guard < (1 << 17); ++guard) {
        const int pq = q - 1;
        if (pq / SFFK_STAR_KC > i && S.prop[pq] < STAR_INF) { last = false; break; }
        q = S.next[pq];
      }
      if (last) {
        f.parent[x] = id;                       // neighbor.Closest = newNode (:344)
        f.d_closest[x] = S.m_d[p];
        f.d_root[x] = pr;                       // descendants keep their costs (:346)
        rewired = 1;
      }
    }
  }
  const int n_rew = __popcll(__ballot(rewired != 0));
  if (lane == 0) {
    // ---- the new node (:329, :353-367); a later sample of the round may already have rewired it
    const int ex = A.parent[i];
    const double* np = A.newpos + 6 * (size_t)i;
    const size_t o = (size_t)id;
    GridItem it;
    for (int k = 0; k < 6; ++k) it.p[k] = np[k];
    it.id = id;
    it.tree = A.st.tree[ex];
    it.pad[0] = it.pad[1] = 0;
    A.st.x[o] = (float)np[0]; A.st.y[o] = (float)np[1]; A.st.z[o] = (float)np[2];
    A.st.yaw[o] = (float)np[3]; A.st.pitch[o] = (float)np[4]; A.st.roll[o] = (float)np[5];
    for (int k = 0; k < 6; ++k) A.st.pos[6 * o + k] = np[k];
    A.st.tree[o] = it.tree;
    int par = S.psel[i];
    double dc = S.dcl[i], dr = S.best[i];
    int more = 0;
    {
      const unsigned long long h = S.head[id];
      int q = (unsigned)(h >> 32) == ep ? (int)(unsigned)(h & 0xffffffffULL) : 0;
      int bs = -1;
      for (int guard = 0; q && guard < (1 << 17); ++guard) {
        const int pq = q - 1;
        const int s = pq / SFFK_STAR_KC;
        if (s > bs) {
          const double pr = S.prop[pq];
          if (pr < STAR_INF) { bs = s; dr = pr; dc = S.m_d[pq]; }
        }
        q = S.next[pq];
      }
      if (bs >= 0) { int rk; star_accepted(f, bs, rk); par = N0 + rk; more = 1; }
    }
    f.parent[o] = par;
    f.d_closest[o] = dc;
    f.d_root[o] = dr;
    f.iter[o] = (uint32_t)(c->iter0_app + i + 1);
    f.nflag[o] = 2;
    (c->front_sel ? f.frontier2 : f.frontier)[fn0 + r] = id;   // :365
    grid_put(A.g, it);                                          // flannIndex->addPoints, :367
    atomicAdd(S.tree_cnt + 16 * it.tree, 1);
    unsigned long long* acc = S.acc + (size_t)(blockIdx.x & 63) * SFFK_STAR_ACC;
    atomicAdd(acc + 0, S.cnt[2 * (size_t)i]);
    atomicAdd(acc + 1, S.cnt[2 * (size_t)i + 1]);
    atomicAdd(acc + 4, (unsigned long long)cnt);
    if (n_rew + more) atomicAdd(acc + 5, (unsigned long long)(n_rew + more));
    if (r == 0) { atomicAdd(acc + 2, 1ULL); atomicAdd(acc + 3, (unsigned long long)passes); }
  }
}

void launch_star_stage(hipStream_t s, const ResolveArgs& a, int n_bound, const StarLaunch& L) {
  if (n_bound <= 0) return;
  const int sample_blocks = (n_bound + 3) / 4;
  hipLaunchKernelGGL(k_star_knn, dim3(sample_blocks), dim3(256), 0, s, a, L.g, L.tg, L.st, L.cell_edge, L.slack);
  // the member edges: compact -> clearance cull -> exact, sized by the header the commit wrote ({accepted samples, skip})
  launch_round_collide(s, L.env, L.rob, nullptr, 0, nullptr, nullptr, a.S.seg_a, a.S.seg_b, a.S.seg_ns,
                       n_bound * SFFK_STAR_KC * 2, a.S.ectrl, L.list, L.list_cap, L.masks, a.S.first_hit, a.S.seg_ovf, nullptr,
                       a.S.hdr, SFFK_STAR_KC * 2);
  const int event_blocks = (n_bound + 255) / 256;
  const int passes = L.passes > 0 && L.passes < SFFK_STAR_PASSES ? L.passes : SFFK_STAR_PASSES;
  for (int pass = 0; pass < passes; ++pass)
    hipLaunchKernelGGL(k_star_pass, dim3(sample_blocks + event_blocks), dim3(256), 0, s, a, pass, sample_blocks);
  hipLaunchKernelGGL(k_star_apply, dim3(sample_blocks), dim3(256), 0, s, a, L.tg, n_bound, passes);
}

}  // namespace sffk
