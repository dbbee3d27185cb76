// devstar.hip — SFF* (optimize = true) on the device-resident engine: choose-parent + rewire of
// SpaceForest::expandNode (src/forest.h:307-351) for the samples a round accepts (gfx950).
//
// What is sequential in the reference and how it is kept: sample i of a round
//   * takes k = floor(2e log10(#nodes)) with the nodes accepted before it counted in (:309),
//   * looks its k nearest up among the store AND those earlier samples (:317),
//   * reads each member's DistanceToRoot as the earlier samples' rewires left it (:322, :337),
//   * rewires members itself (:336-348) - a later sample of the round may rewire the same node again.
// Acceptance (who becomes a node, with which id) does not depend on any cost, so k_commit settles it first,
// exactly as for plain SFF.  The costs are then the unique fixed point of
//     view(i, x)  = proposal of the latest j < i whose rewire of x is active, else cost(x)            (per-node lists)
//     result(i)   = choose-parent / rewire of sample i evaluated on its views                          (one wavefront)
// and because j < i always, iterating "every sample recomputes" reaches it in (longest dependency chain + 1) passes
// whatever the order inside a pass; a pass that writes nothing proves it (kernel boundaries make a pass's writes visible
// to the next; inside k_star_tail - kernels.hip, all passes after the first in one launch - the exchanged words are written
// through and read from memory, star_pass_dev.h).  Everything is evaluated in the host engine's expression order
// (-ffp-contract=off): bit-identical.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"
#include "sff_geom.h"
#include "kernels_dev.h"
#include "star_pass_dev.h"
#include <cmath>
#include <cstring>
#include <cstdlib>

namespace sffk {

using namespace sffg;



// candidates of a group of up to 64 cells of the node grid (lane = cell, m = its item count), flattened over the lanes:
// permanent nodes (id < N0) of the sample's tree
__device__ __forceinline__ void star_cells(const GridView& g, int m, int cell, int lane, const double* qp, int tree, int N0,
                                           TopK& t, int k, int& have) {
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
      if (it.tree == tree && id < N0) { d = dist6(it.p, qp); cand = true; }
    }
    const double worst = topk_worst(t, k, have);
    cand = cand && (have < k || key_less(d, id, worst, 0x7fffffff));
    topk_insert(t, lane, k, have, __ballot(cand), d, id);
  }
}

// ------------------------------------------------------------------ k nearest + toucher lists
// half-width R0 (cells) of the cube gathered in one go: 2 .. STAR_R0_MAX, chosen by the host so that the cube covers about
// two sampling distances whatever the grid's cell size has become (the grid re-cells itself as the forest gets denser)
#define STAR_R0_MAX 4
#define STAR_U (((2 * STAR_R0_MAX + 1) * (2 * STAR_R0_MAX + 1) * (2 * STAR_R0_MAX + 1) + 63) / 64)
#define STAR_MATE_U 16                              // x 64 accepted samples of a round whose list is scanned from registers
#define STAR_POOL 10                                // batches of 64 cube candidates selected from at once
#define STAR_INF_BITS 0x7ff0000000000000ULL
__global__ __launch_bounds__(256) void k_star_knn(ResolveArgs A, GridView g, GridView tg, NodeStoreView st, double cell_edge,
                                                  double slack, int R0) {
  __shared__ double s_sel_d[4][64], s_srt_d[4][64];
  __shared__ int s_sel_i[4][64], s_srt_i[4][64];
  __shared__ int s_mate[4][64 * STAR_MATE_U];
  const DevForestView& f = A.f;
  const StarView& S = A.S;
  const DevCtrl* c = f.ctrl;
  const int n = c->app_n;
  if (blockIdx.x == 0 && threadIdx.x < SFFK_STAR_PASSES) S.changed[threadIdx.x] = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) { S.changed[SFFK_STAR_BAR] = 0; S.hdr[STAR_PASSES_RUN] = 0; S.hdr[STAR_CONVERGED] = 0; }   // (k_star_tail)
  for (int t = blockIdx.x * 256 + threadIdx.x; t < SFFK_STAR_PASSES * SFFK_SUBLISTS * SFFK_STAR_SUB; t += gridDim.x * 256) S.sub[t] = 0;
  if (n <= 0) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + wave;
  if (i >= n) return;
  int r;
  if (!star_accepted(f, i, r)) return;
  const int N0 = c->app_N0, Tb = f.temp_base;
  const unsigned ep = (unsigned)c->epoch;
  const unsigned long long dbg_t0 = S.dbg ? wall_clock64() : 0ULL;
  unsigned long long dbg_t1 = 0, dbg_t2 = 0, dbg_t3 = 0, dbg_c1 = 0, dbg_c2 = 0, dbg_c3 = 0;
  int dbg_shells = 0, dbg_total = 0;
  // k = (size_t)(2e log10(#nodes)) with the nodes accepted before this sample counted in (src/forest.h:309)
  const int Nn = N0 + r;
  const int k_ref = __popcll(__ballot(lane > 0 && lane <= SFFK_STAR_KMAX + 1 && S.ktab[lane] <= Nn));
  const int self = Tb + i;
  const int mine = st.tree[self];
  double qp[6];
  for (int q = 0; q < 6; ++q) qp[q] = A.newpos[6 * (size_t)i + q];
  // (round 5) what the END of this kernel needs is asked for now, beside the cube's own loads: the earlier accepted samples
  // of the round (rank -> sample -> tree: two dependent trips that used to follow the cube and the shells, 12.5 us per
  // sample) and lane 0's first guess (expanded node -> its cost)
  int sidv[STAR_MATE_U];
  const bool mates_small = r > 0 && r <= 64 * STAR_MATE_U;
#pragma unroll
  for (int u = 0; u < STAR_MATE_U; ++u) { const int rq = 64 * u + lane; sidv[u] = (mates_small && rq < r) ? Tb + S.acc_sample[rq] : -1; }
  const int ex0 = A.parent[i];
  const double pd0 = A.pdist[i];
  const int no_raw = g.ovf_cnt[0];
  const int tcnt = S.tree_cnt[16 * mine];
  {   // the sample's edge slots: nothing asked for yet
    const size_t s0 = ((size_t)i * SFFK_STAR_KC + lane) * 2;
    S.ew[s0] = 0; S.ew[s0 + 1] = 0;
  }
  if (k_ref > SFFK_STAR_KMAX) {   // (a node count beyond what the member slots are sized for: host path)
    if (lane == 0) atomicOr(S.hdr + STAR_FAULT, 1);
    return;
  }
  const int k = k_ref;
  const int k_store = k < tcnt ? k : tcnt;
  TopK t{1.0e300, 0x7fffffff};
  int have = 0;
  int trv[STAR_MATE_U];
#pragma unroll
  for (int u = 0; u < STAR_MATE_U; ++u) trv[u] = -1;
  double dr0 = 0.0;
  if (k <= 0) dr0 = f.d_root[ex0];
  if (k > 0) {
    const int cx = grid_coord((float)qp[0], g.ox, g.inv_cell, g.nx), cy = grid_coord((float)qp[1], g.oy, g.inv_cell, g.ny),
              cz = grid_coord((float)qp[2], g.oz, g.inv_cell, g.nz);
    // ---- the cube of half-width R0 in one go: every cell's count is requested before anything is looked at (one
    // trip to memory instead of one per shell), then the cube's candidates are flattened over the lanes
    int cellv[STAR_U], mv[STAR_U];
    GridItem ovf_pre[2];
    const int W5 = 2 * R0 + 1, cube = W5 * W5 * W5;
#pragma unroll
    for (int u = 0; u < STAR_U; ++u) {
      const int cc = u * 64 + lane;
      cellv[u] = 0; mv[u] = 0;
      if (cc < cube) {
        const int x = cx + cc % W5 - R0, y = cy + (cc / W5) % W5 - R0, z = cz + cc / (W5 * W5) - R0;
        if (x >= 0 && x < g.nx && y >= 0 && y < g.ny && z >= 0 && z < g.nz) {
          cellv[u] = (z * g.ny + y) * g.nx + x;
          mv[u] = g.cnt[cellv[u]];
        }
      }
    }
    // (the second trips of the loads asked for at the top go out while the cube's counts are on their way; so do the first
    // 128 entries of the node grid's shared overflow list, which every sample scans after its cube)
#pragma unroll
    for (int u = 0; u < STAR_MATE_U; ++u) trv[u] = sidv[u] >= 0 ? st.tree[sidv[u]] : -1;
    dr0 = f.d_root[ex0];
    const int no_pre = no_raw < g.ovf_cap ? no_raw : g.ovf_cap;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      ovf_pre[h].id = 0x7fffffff; ovf_pre[h].tree = -1;
      if (64 * h + lane < no_pre) ovf_pre[h] = g.ovf[64 * h + lane];
    }
#pragma unroll
    for (int u = 0; u < STAR_U; ++u) if (mv[u] > g.bk) mv[u] = g.bk;
    {
      int mine_sum = 0;
#pragma unroll
      for (int u = 0; u < STAR_U; ++u) mine_sum += mv[u];
      int inc = mine_sum;
      for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
      }
      const int total = __shfl(inc, 63);
      dbg_total = total;
      if (S.dbg) dbg_c1 = wall_clock64();
      // candidate j of the flattened cube -> its grid item (two batches of 64 per step: their loads are in flight together)
      auto fetch2 = [&](int base, GridItem* it, bool* val) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int j = base + 64 * h + lane;
          const int jj = j < total ? j : total - 1;
          int lo = 0, hi = 63;
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (__shfl(inc, mid) > jj) hi = mid; else lo = mid + 1;
          }
          int rest = jj - (__shfl(inc, lo) - __shfl(mine_sum, lo));   // index among the owner lane's items
          int src_cell = 0;
#pragma unroll
          for (int u = 0; u < STAR_U; ++u) {
            if (u * 64 >= cube) break;
            const int mu = __shfl(mv[u], lo), cu = __shfl(cellv[u], lo);
            if (rest >= 0 && rest < mu) { src_cell = cu; rest += 1 << 20; }   // (found: park the index out of every later range)
            else if (rest >= 0 && rest < (1 << 20)) rest -= mu;
          }
          val[h] = j < total;
          it[h].id = 0x7fffffff; it[h].tree = -1;
          if (val[h]) it[h] = g.items[(size_t)src_cell * g.bk + (rest - (1 << 20))];
        }
      };
      // The first STAR_POOL x 64 candidates are only measured, not ranked one by one: their k smallest keys are found
      // by bisection on the distance bits (ballots and scalar counts), and only those k are sorted into the list - the
      // list's rank-insertion costs a dozen cross-lane moves per newcomer, and while it fills nearly every candidate is one
      unsigned long long kb[STAR_POOL];
      int kid[STAR_POOL];
#pragma unroll
      for (int b = 0; b < STAR_POOL; ++b) { kb[b] = STAR_INF_BITS; kid[b] = 0x7fffffff; }
#pragma unroll
      for (int sp = 0; sp < STAR_POOL / 2; ++sp) {
        if (sp * 128 < total) {
          GridItem it[2];
          bool val[2];
          fetch2(sp * 128, it, val);
#pragma unroll
          for (int h = 0; h < 2; ++h)
            if (val[h] && it[h].tree == mine && it[h].id < N0) {
              kb[2 * sp + h] = (unsigned long long)__double_as_longlong(dist6(it[h].p, qp));
              kid[2 * sp + h] = it[h].id;
            }
        }
      }
      {
        if (S.dbg) dbg_c2 = wall_clock64();
        // Selection of the k smallest keys (distance bits, then id) among the pool, round 5: the bisection runs on the HIGH
        // 32 bits of the keys first (a 32-bit compare per batch and round instead of a 64-bit one, 31 rounds instead of 63,
        // only the batches the cube filled); everything below the boundary value is in, the (usually single) keys that
        // share the boundary's high word are told apart by their low words, equal distances by their ids - the same set as
        // the 63-round bisection on the whole key picked.
        const int nbat = (total + 63) >> 6;           // batches of the pool that hold candidates
        unsigned hi[STAR_POOL], lo[STAR_POOL];
        int n_valid = 0;
#pragma unroll
        for (int b = 0; b < STAR_POOL; ++b) {
          hi[b] = (unsigned)(kb[b] >> 32); lo[b] = (unsigned)(kb[b] & 0xffffffffULL);
          if (b < nbat) n_valid += __popcll(__ballot(kb[b] < STAR_INF_BITS));
        }
        const unsigned INF_HI = (unsigned)(STAR_INF_BITS >> 32);
        unsigned Vh = INF_HI, Vl = 0xffffffffu;       // winners: hi < Vh, or hi == Vh with (lo < Vl, or lo == Vl with id <= I)
        int I = 0x7fffffff;
        if (n_valid > k) {
          Vh = 0u;
          for (int bit = 30; bit >= 0; --bit) {       // Vh = high word of the k-th smallest key: the largest x with count(hi < x) < k
            const unsigned trial = Vh | (1u << bit);
            int cnt_less = 0;
#pragma unroll
            for (int b = 0; b < STAR_POOL; ++b) if (b < nbat) cnt_less += __popcll(__ballot(hi[b] < trial));
            if (cnt_less < k) Vh = trial;
          }
          int cA = 0, tB = 0;
#pragma unroll
          for (int b = 0; b < STAR_POOL; ++b)
            if (b < nbat) { cA += __popcll(__ballot(hi[b] < Vh)); tB += __popcll(__ballot(hi[b] == Vh)); }
          const int needB = k - cA;
          if (tB > needB) {                            // several keys share the boundary's high word: their low words
            Vl = 0u;
            for (int bit = 31; bit >= 0; --bit) {
              const unsigned trial = Vl | (1u << bit);
              int cl = 0;
#pragma unroll
              for (int b = 0; b < STAR_POOL; ++b) if (b < nbat) cl += __popcll(__ballot(hi[b] == Vh && lo[b] < trial));
              if (cl < needB) Vl = trial;
            }
            int cB = 0, tC = 0;
#pragma unroll
            for (int b = 0; b < STAR_POOL; ++b)
              if (b < nbat) { cB += __popcll(__ballot(hi[b] == Vh && lo[b] < Vl)); tC += __popcll(__ballot(hi[b] == Vh && lo[b] == Vl)); }
            const int needC = needB - cB;
            if (tC > needC) {                          // equal distances: the smallest ids among them
              unsigned int Iu = 0u;
              for (int bit = 30; bit >= 0; --bit) {
                const unsigned int trial = Iu | (1u << bit);
                int cl = 0;
#pragma unroll
                for (int b = 0; b < STAR_POOL; ++b)
                  if (b < nbat) cl += __popcll(__ballot(hi[b] == Vh && lo[b] == Vl && (unsigned int)kid[b] < trial));
                if (cl < needC) Iu = trial;
              }
              I = (int)Iu;
            }
          }
        }
        if (S.dbg) dbg_c3 = wall_clock64();
        // the winners, compacted into LDS, then ranked among themselves (k reads each, all lanes the same address)
        double* ud = s_sel_d[wave];
        int* ui = s_sel_i[wave];
        int nsel = 0;
#pragma unroll
        for (int b = 0; b < STAR_POOL; ++b) {
          const bool win = b < nbat && kb[b] < STAR_INF_BITS &&
                           (hi[b] < Vh || (hi[b] == Vh && (lo[b] < Vl || (lo[b] == Vl && kid[b] <= I))));
          const unsigned long long m = __ballot(win);
          if (win) {
            const int at = nsel + __popcll(m & ((1ULL << lane) - 1ULL));
            ud[at] = __longlong_as_double((long long)kb[b]);
            ui[at] = kid[b];
          }
          nsel += __popcll(m);
        }
        __builtin_amdgcn_wave_barrier();
        double md = 1.0e300;
        int mi = 0x7fffffff, rank = 0;
        if (lane < nsel) { md = ud[lane]; mi = ui[lane]; }
        for (int q = 0; q < nsel; ++q) {
          const double od = ud[q];
          const int oi = ui[q];
          if (key_less(od, oi, md, mi)) ++rank;
        }
        __builtin_amdgcn_wave_barrier();
        double* sd = s_srt_d[wave];
        int* si = s_srt_i[wave];
        if (lane < nsel) { sd[rank] = md; si[rank] = mi; }
        __builtin_amdgcn_wave_barrier();
        if (lane < nsel) { t.d = sd[lane]; t.id = si[lane]; }
        have = nsel;
      }
      // whatever the pool did not hold (a cube with more than STAR_POOL x 64 nodes), one by one
      for (int base = STAR_POOL * 64; base < total; base += 128) {
        GridItem it[2];
        bool val[2];
        fetch2(base, it, val);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (base + 64 * h >= total) break;
          bool cand = false;
          double d = 1.0e300;
          const int id = it[h].id;
          if (val[h] && it[h].tree == mine && id < N0) { d = dist6(it[h].p, qp); cand = true; }
          const double worst = topk_worst(t, k, have);
          cand = cand && (have < k || key_less(d, id, worst, 0x7fffffff));
          topk_insert(t, lane, k, have, __ballot(cand), d, id);
        }
      }
    }
    {   // the node grid's shared overflow list (usually empty)
      int no = no_raw;
      if (no > g.ovf_cap) no = g.ovf_cap;
      for (int base = 0; base < no; base += 64) {
        const int j = base + lane;
        bool cand = false;
        double d = 1.0e300;
        int id = 0x7fffffff;
        const double worst = topk_worst(t, k, have);
        if (j < no) {
          const GridItem it = base == 0 ? ovf_pre[0] : (base == 64 ? ovf_pre[1] : g.ovf[j]);
          id = it.id;
          if (id < N0 && it.tree == mine) {
            d = dist6(it.p, qp);
            cand = have < k || key_less(d, id, worst, 0x7fffffff);
          }
        }
        topk_insert(t, lane, k, have, __ballot(cand), d, id);
      }
    }
    // ---- beyond the cube (rare: young, sparse trees): shells of cells until the k-th distance lies inside the covered
    // ball; a tree with fewer than k nodes is complete as soon as all of them are in
    const int rmax = max(max(g.nx, g.ny), g.nz);
    if (S.dbg) dbg_t1 = wall_clock64();
    for (int rr = R0; rr <= rmax; ++rr) {
      if (have >= k_store && k_store == tcnt) break;      // the whole tree is in the list
      if (rr > R0) {
        ++dbg_shells;
        // only the shell's own cells (two caps of w x w, w - 2 rings of 8 rr: 24 rr^2 + 2 of the cube's (2 rr + 1)^3 - the
        // loop used to run over the whole cube and mask its inside), four cells per lane and trip so that the counts of
        // 256 cells are on their way together (round 5, like k_knn_grid)
        const int w = 2 * rr + 1;
        const int ww = w * w, ring = 8 * rr;
        const int total = 2 * ww + (w - 2) * ring;
        for (int c0 = 0; c0 < total; c0 += 256) {
          int cellq[4], mq[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int cq = c0 + u * 64 + lane;
            cellq[u] = 0; mq[u] = 0;
            if (cq < total) {
              int ox, oy, oz;
              if (cq < 2 * ww) {
                const int face = cq >= ww ? 1 : 0, ii = cq - face * ww;
                ox = ii % w - rr; oy = ii / w - rr; oz = face ? rr : -rr;
              } else {
                const int cc = cq - 2 * ww;
                const int layer = cc / ring, pp = cc - layer * ring;
                const int side = pp / (2 * rr), t_ = pp - side * 2 * rr;
                oz = -rr + 1 + layer;
                ox = side == 0 ? -rr + t_ : side == 1 ? rr : side == 2 ? rr - t_ : -rr;
                oy = side == 0 ? -rr : side == 1 ? -rr + t_ : side == 2 ? rr : rr - t_;
              }
              const int x = cx + ox, y = cy + oy, z = cz + oz;
              if (x >= 0 && x < g.nx && y >= 0 && y < g.ny && z >= 0 && z < g.nz) {
                cellq[u] = (z * g.ny + y) * g.nx + x;
                mq[u] = g.cnt[cellq[u]];
                if (mq[u] > g.bk) mq[u] = g.bk;
              }
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (__any(mq[u] > 0)) star_cells(g, mq[u], cellq[u], lane, qp, mine, N0, t, k, have);
        }
      }
      const double covered = (double)rr * cell_edge - slack;
      if (have >= k && topk_worst(t, k, have) <= covered) break;
      if (cx - rr <= 0 && cy - rr <= 0 && cz - rr <= 0 && cx + rr >= g.nx - 1 && cy + rr >= g.ny - 1 && cz + rr >= g.nz - 1) break;
    }
    // ---- the samples accepted earlier in this round (ranks below this one's, k_commit's list) of the same tree: not
    // farther than the k-th store node - or, while the store holds fewer than k nodes of the tree, all of them
    if (S.dbg) dbg_t2 = wall_clock64();
    if (r > 0) {
      const bool all = have < k;
      const double limit = all ? 1.0e300 : topk_worst(t, k, have);
      if (r <= 64 * STAR_MATE_U) {
        // k_commit's list of the accepted samples: every rank's sample and tree are requested up front (two trips to
        // memory whatever the length), the few of the same tree are compacted in LDS and measured a batch at a time
        int* ml = s_mate[wave];
        int nm = 0;
#pragma unroll
        for (int u = 0; u < STAR_MATE_U; ++u) {
          if (64 * u >= r) break;
          const bool mt = sidv[u] >= 0 && trv[u] == mine;
          const unsigned long long mm = __ballot(mt);
          if (mt) ml[nm + __popcll(mm & ((1ULL << lane) - 1ULL))] = sidv[u];
          nm += __popcll(mm);
        }
        __builtin_amdgcn_wave_barrier();
        for (int base = 0; base < nm; base += 64) {
          const int j = base + lane;
          bool cand = false;
          double d = 1.0e300;
          int sid = 0x7fffffff;
          if (j < nm) {
            sid = ml[j];
            double mp[6];
            for (int q = 0; q < 6; ++q) mp[q] = st.pos[6 * (size_t)sid + q];
            d = dist6(mp, qp);
            cand = d <= limit;
          }
          const double worst = topk_worst(t, k, have);
          cand = cand && (have < k || key_less(d, sid, worst, 0x7fffffff));
          topk_insert(t, lane, k, have, __ballot(cand), d, sid);
        }
      } else if (all) {
        for (int base = 0; base < r; base += 64) {   // (a tree wanted whole in a huge round: rare)
          const int rq = base + lane;
          bool cand = false;
          double d = 1.0e300;
          int sid = 0x7fffffff;
          if (rq < r) {
            sid = Tb + S.acc_sample[rq];
            if (st.tree[sid] == mine) {
              double mp[6];
              for (int q = 0; q < 6; ++q) mp[q] = st.pos[6 * (size_t)sid + q];
              d = dist6(mp, qp);
              cand = true;
            }
          }
          const double worst = topk_worst(t, k, have);
          cand = cand && (have < k || key_less(d, sid, worst, 0x7fffffff));
          topk_insert(t, lane, k, have, __ballot(cand), d, sid);
        }
      } else {
        // a large round: the round's own grid, cells of the cube around the ball of the k-th store node
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
          if (__any(m > 0)) star_cells_mates(tg, m, cell, lane, qp, mine, Tb, self, limit, f, t, k, have);
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
  if (S.dbg) dbg_t3 = wall_clock64();
  // ---- the members: ids, distances, toucher lists
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
  if (lane == 0) {
    S.m_cnt[i] = cnt;
    S.best[i] = pd0 + dr0;   // (first guess: the plain SFF cost)
    S.psel[i] = ex0;
    S.dcl[i] = pd0;
    S.cnt[2 * (size_t)i] = 0ULL; S.cnt[2 * (size_t)i + 1] = 0ULL;
    if (S.dbg && k > 0) {   // waves | ticks: cube, shells, mates, lists | shells walked | cube candidates | longest wave
      const unsigned long long t4 = wall_clock64();
      atomicAdd(S.dbg + 0, 1ULL); atomicAdd(S.dbg + 1, dbg_t1 - dbg_t0); atomicAdd(S.dbg + 2, dbg_t2 - dbg_t1);
      atomicAdd(S.dbg + 3, dbg_t3 - dbg_t2); atomicAdd(S.dbg + 4, t4 - dbg_t3); atomicAdd(S.dbg + 5, (unsigned long long)dbg_shells);
      atomicAdd(S.dbg + 6, (unsigned long long)dbg_total); atomicMax(S.dbg + 7, t4 - dbg_t0);
      if (dbg_shells) atomicAdd(S.dbg + 8, 1ULL);
      atomicAdd(S.dbg + 9, dbg_c1 - dbg_t0); atomicAdd(S.dbg + 10, dbg_c2 - dbg_c1); atomicAdd(S.dbg + 11, dbg_c3 - dbg_c2);
      atomicAdd(S.dbg + 12, dbg_t1 - dbg_c3);
    }
  }
}

// ------------------------------------------------------------------ one pass of the fixed point (bodies: star_pass_dev.h)
__global__ __launch_bounds__(256) void k_star_pass(ResolveArgs A, EnvView env, NodeStoreView st, int pass, int sample_blocks) {
  __shared__ StarPassLds L;
  const DevForestView& f = A.f;
  const StarView& S = A.S;
  const DevCtrl* c = f.ctrl;
  // (every word the early returns look at is asked for before the first of them is tested: a launch that finds nothing to
  //  do - most of a round's 15 pass / exact launches - costs one trip to memory instead of three one after the other)
  const int app_n = c->app_n, N0 = c->app_N0;
  const unsigned ep = (unsigned)c->epoch;
  const int h_n = S.hdr[0], h_skip = S.hdr[1], h_fault = S.hdr[STAR_FAULT];
  const int chg = pass > 0 ? S.changed[pass - 1] : 1;
  // (the rank's sample with them; all but the last sample workgroup lie within the list whatever the round's count is)
  const int i_pre = (int)blockIdx.x < sample_blocks - 1 ? S.acc_sample[blockIdx.x * 4 + (threadIdx.x >> 6)] : -1;
  if (app_n <= 0 || h_skip || h_fault) return;
  if (chg == 0) return;                                     // the pass before wrote nothing: fixed point reached
  const int Tb = f.temp_base;
  if ((int)blockIdx.x >= sample_blocks) {
    const int e = ((int)blockIdx.x - sample_blocks) * 256 + threadIdx.x;
    if (e < S.hdr[2]) star_pass_event<false>(f, S, e, S.hdr[3], N0, ep);
    return;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + wave;
  if (r >= h_n) return;
  const int i = i_pre >= 0 ? i_pre : S.acc_sample[r];
  star_pass_sample<false>(A, env, st, pass, blockIdx.x & (SFFK_SUBLISTS - 1), i, N0, Tb, ep, L, wave, lane);
}

// ------------------------------------------------------------------ apply: nodes, rewires, housekeeping
__global__ __launch_bounds__(256) void k_star_apply(ResolveArgs A, GridView tg, int n_bound, int max_passes, int tail) {
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
  const int h_skip = S.hdr[1], h_fault = S.hdr[STAR_FAULT];
  int chg[SFFK_STAR_PASSES];                  // (asked for together: one trip to memory, not one per pass)
#pragma unroll
  for (int t = 0; t < SFFK_STAR_PASSES; ++t) chg[t] = S.changed[t];
  const int t_passes = S.hdr[STAR_PASSES_RUN], t_conv = S.hdr[STAR_CONVERGED];
  if (n <= 0 || h_skip) return;
  // fixed point reached?  (the first pass that wrote nothing; none = not converged within the launches of a round)
  int passes = 0;
  bool conv = false;
  if (tail) {                                   // (the passes after the first ran in k_star_tail: its count, its verdict)
    passes = t_passes;
    conv = t_conv != 0;
  } else {
#pragma unroll
    for (int t = 0; t < SFFK_STAR_PASSES; ++t) {
      if (t < max_passes && !conv) {
        ++passes;
        if (chg[t] == 0) conv = true;
      }
    }
  }
  if (!conv || h_fault) {
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
  if (S.hist) {
    // the parent history (exact per-iteration tree dumps): this sample's node with the parent it chose, then every member
    // its proposal rewires at its turn - whether or not a later sample of the round rewires that node again
    const bool actv = lane < cnt && S.prop[p] < STAR_INF;
    const unsigned long long am = __ballot(actv);
    const int nrec = 1 + __popcll(am);
    int base = 0;
    if (lane == 0) base = atomicAdd(S.hist_ctl, nrec);
    base = __shfl(base, 0);
    const int it_i = c->iter0_app + i + 1;
    if (base + nrec <= S.hist_cap) {
      if (lane == 0) { S.hist[3 * (size_t)base] = id; S.hist[3 * (size_t)base + 1] = S.psel[i]; S.hist[3 * (size_t)base + 2] = it_i; }
      if (actv) {
        const size_t at = (size_t)base + 1 + __popcll(am & ((1ULL << lane) - 1ULL));
        S.hist[3 * at] = S.m_id[p]; S.hist[3 * at + 1] = id; S.hist[3 * at + 2] = it_i;
      }
    } else if (lane == 0) S.hist_ctl[1] = 1;
  }
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
    f.nflag[o] = f.prio.n_heaps ? 0 : 2;
    if (!f.prio.n_heaps) (c->front_sel ? f.frontier2 : f.frontier)[fn0 + r] = id;   // :365 (priority mode: k_prio_end)
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
  int R0 = (int)std::ceil(L.cube_reach / L.cell_edge);
  R0 = R0 < 2 ? 2 : (R0 > STAR_R0_MAX ? STAR_R0_MAX : R0);
  static const bool lone = getenv("SFFGPU_STAR_KNN") && !strcmp(getenv("SFFGPU_STAR_KNN"), "lone");
  if (lone) hipLaunchKernelGGL(k_star_knn, dim3(sample_blocks), dim3(256), 0, s, a, L.g, L.tg, L.st, L.cell_edge, L.slack, R0);
  else launch_star_knn_wg(s, a, L.g, L.tg, L.st, L.cell_edge, L.slack, n_bound, R0);
  const int event_blocks = (n_bound + 255) / 256;
  // the passes after the first: one launch (k_star_tail, as many passes as the round needs, up to SFFK_STAR_TAIL_PASSES) or -
  // SFFGPU_STAR_TAIL=0 when the forest was created - the fixed chain of up to SFFK_STAR_PASSES pass / exact launches
  if (L.tail) {
    const int passes = L.passes > 0 ? L.passes : SFFK_STAR_TAIL_PASSES;
    hipLaunchKernelGGL(k_star_pass, dim3(sample_blocks + event_blocks), dim3(256), 0, s, a, L.env, L.st, 0, sample_blocks);
    if (passes > 1) launch_star_exact(s, L.env, L.rob, L.st.pos, a.S, 0);
    launch_star_tail(s, a, L.env, L.rob, L.st, n_bound, passes, L.tail_wgs, L.tail_stall);
    hipLaunchKernelGGL(k_star_apply, dim3(sample_blocks), dim3(256), 0, s, a, L.tg, n_bound, passes, 1);
    return;
  }
  const int passes = L.passes > 0 && L.passes < SFFK_STAR_PASSES ? L.passes : SFFK_STAR_PASSES;
  for (int pass = 0; pass < passes; ++pass) {
    hipLaunchKernelGGL(k_star_pass, dim3(sample_blocks + event_blocks), dim3(256), 0, s, a, L.env, L.st, pass, sample_blocks);
    // (what the pass could not answer from the clearance bits; nothing to do = the launch returns at once)
    if (pass + 1 < passes) launch_star_exact(s, L.env, L.rob, L.st.pos, a.S, pass);
  }
  hipLaunchKernelGGL(k_star_apply, dim3(sample_blocks), dim3(256), 0, s, a, L.tg, n_bound, passes, 0);
}

}  // namespace sffk
