// kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the SFF hot path.
//
//   k_sample_steer   : RandGen::randomPointInDistance + Point::getStateInDistance (+ limits test)
//                      for a whole wave of frontier slots in one launch; also emits the fp32
//                      neighbour-sweep query of each sample.
//   k_sweep          : linear neighbour sweep over the SoA fp32 node store (replaces FLANN
//                      radiusSearch / knnSearch).  HBM-streaming kernel: 16 B/lane coalesced
//                      column loads, queries read through the scalar cache (wave-uniform),
//                      fp32 superset filter, exact fp64 re-test of the rare survivors.
//   k_collide_poses  : Environment::Collide — one wave per pose, wave-cooperative traversal of a
//                      64-ary box hierarchy (one child box per lane, __ballot compaction), robot
//                      triangles staged in LDS, exact fp64 triangle contact at the leaves.
//   k_collide_segments: Solver::isPathFree — one wave per (edge, 64-sample chunk), lane = sample;
//                      broad phase with the chunk's swept robot box, plane + box culls per
//                      (sample, triangle), exact first-hit index via atomicMin.
//
// Compiled with -ffp-contract=off: every fp64 value that feeds a decision has the same bits as
// the host evaluation of sff_geom.h.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "kernels.h"
#include <algorithm>
#include "sff_geom.h"
#include "kernels_dev.h"
#include "star_pass_dev.h"

namespace sffk {

using namespace sffg;


// ------------------------------------------------------------------ sample + steer
__global__ __launch_bounds__(256) void k_sample_steer(SampleLaunch P) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  sample_steer_one(i, i, -1, P);
}

// ------------------------------------------------------------------ neighbour sweep
__device__ __forceinline__ float wrapf(float d) {
  const float PI_F = 3.14159274f;
  if (d < -PI_F) return d + 2.0f * PI_F;
  if (d >= PI_F) return d - 2.0f * PI_F;
  return d;
}

// One thread owns 4 consecutive nodes (float4 per column = 16 B/lane, fully coalesced) and loops
// over the wave-uniform query list; a query's parameters are fetched with scalar loads.
__global__ __launch_bounds__(256) void k_sweep(NodeStoreView st, int first, int n_nodes,
                                               const SweepQuery* __restrict__ queries,
                                               const double* __restrict__ qpos,  // nq x 6 exact query positions
                                               int nq, int q_per_block, int32_t* __restrict__ cnt,
                                               int32_t* __restrict__ hit_idx, double* __restrict__ hit_dist, int cap) {
  const int n4 = (n_nodes + 3) >> 2;
  // blockIdx.y selects a slice of the query list: small stores still fill the chip
  const int q_begin = blockIdx.y * q_per_block;
  const int q_end = q_begin + q_per_block < nq ? q_begin + q_per_block : nq;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n4; t += gridDim.x * blockDim.x) {
    const int base = first + (t << 2);
    const int t4 = (first >> 2) + t;
    float4 X = reinterpret_cast<const float4*>(st.x)[t4];
    float4 Y = reinterpret_cast<const float4*>(st.y)[t4];
    float4 Z = reinterpret_cast<const float4*>(st.z)[t4];
    float4 A = reinterpret_cast<const float4*>(st.yaw)[t4];
    float4 B = reinterpret_cast<const float4*>(st.pitch)[t4];
    float4 C = reinterpret_cast<const float4*>(st.roll)[t4];
    const float xs[4] = {X.x, X.y, X.z, X.w}, ys[4] = {Y.x, Y.y, Y.z, Y.w}, zs[4] = {Z.x, Z.y, Z.z, Z.w};
    const float as[4] = {A.x, A.y, A.z, A.w}, bs[4] = {B.x, B.y, B.z, B.w}, cs[4] = {C.x, C.y, C.z, C.w};
    for (int q = q_begin; q < q_end; ++q) {
      const SweepQuery Q = queries[q];  // wave-uniform address -> scalar loads
      if (!Q.active) continue;
      float d3[4];
      bool any = false;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float dx = xs[j] - Q.x, dy = ys[j] - Q.y, dz = zs[j] - Q.z;
        d3[j] = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
        any |= d3[j] <= Q.r2f;
      }
      if (!any) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int id = base + j;
        if (!(d3[j] <= Q.r2f) || id >= first + n_nodes || id >= Q.max_id) continue;  // NaN placeholders fail here
        float da = wrapf(as[j] - Q.yaw), db = wrapf(bs[j] - Q.pitch), dc = wrapf(cs[j] - Q.roll);
        float d6 = fmaf(dc, dc, fmaf(db, db, fmaf(da, da, d3[j])));
        if (!(d6 <= Q.r2f)) continue;
        if (Q.tree >= 0 ? st.tree[id] != Q.tree : (Q.tree < -1 && st.tree[id] == -2 - Q.tree)) continue;   // (-2 - t: every tree but t)
        // exact re-test in fp64 on the authoritative positions (reference: realDist, src/forest.h:274)
        double np[6], qp[6];
        for (int k = 0; k < 6; ++k) { np[k] = st.pos[6 * (size_t)id + k]; qp[k] = qpos[6 * (size_t)q + k]; }
        double d = dist6(np, qp);
        if (d < Q.r) {
          int slot = atomicAdd(&cnt[q], 1);
          if (slot < cap) {
            hit_idx[(size_t)q * cap + slot] = id;
            hit_dist[(size_t)q * cap + slot] = d;
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------ grid neighbour query
__global__ __launch_bounds__(256) void k_grid_insert(GridView g, NodeStoreView st, int first, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int id = first + i;
  if (!(st.x[id] == st.x[id])) return;  // NaN placeholder
  GridItem it;
  for (int k = 0; k < 6; ++k) it.p[k] = st.pos[6 * (size_t)id + k];
  it.id = id;
  it.tree = st.tree[id];
  it.pad[0] = it.pad[1] = 0;
  grid_put(g, it);
}

__device__ __forceinline__ void grid_test(const GridItem& it, const SweepQuery& Q, int q, const NodeStoreView& st,
                                          const double* __restrict__ qpos, int32_t* __restrict__ cnt,
                                          int32_t* __restrict__ hit_idx, double* __restrict__ hit_dist, int cap) {
  if (it.id >= Q.max_id) return;
  if (Q.tree >= 0 ? it.tree != Q.tree : (Q.tree < -1 && it.tree == -2 - Q.tree)) return;
  double qp[6];
  for (int k = 0; k < 6; ++k) qp[k] = qpos[6 * (size_t)q + k];
  const double d = dist6(it.p, qp);
  if (d < Q.r) {
    int slot = atomicAdd(&cnt[q], 1);
    if (slot < cap) {
      hit_idx[(size_t)q * cap + slot] = it.id;
      hit_dist[(size_t)q * cap + slot] = d;
    }
  }
}

// One wavefront per query: the cells touched by the query ball's bounding box are dealt to the
// lanes (27 cells for the planner's radius), each lane walks its cell's bucket; all lanes then
// share the overflow list.  Same fp32 superset filter + exact fp64 re-test as the linear sweep.
template <bool SPECULATE>
__device__ __forceinline__ void grid_walk(const GridView& g, const SweepQuery& Q, int q, int lane, float rf,
                                          const NodeStoreView& st, const double* __restrict__ qpos,
                                          int32_t* __restrict__ cnt, int32_t* __restrict__ hit_idx,
                                          double* __restrict__ hit_dist, int cap) {
  const int lx = grid_coord(Q.x - rf, g.ox, g.inv_cell, g.nx), hx = grid_coord(Q.x + rf, g.ox, g.inv_cell, g.nx);
  const int ly = grid_coord(Q.y - rf, g.oy, g.inv_cell, g.ny), hy = grid_coord(Q.y + rf, g.oy, g.inv_cell, g.ny);
  const int lz = grid_coord(Q.z - rf, g.oz, g.inv_cell, g.nz), hz = grid_coord(Q.z + rf, g.oz, g.inv_cell, g.nz);
  const int wx = hx - lx + 1, wy = hy - ly + 1, wz = hz - lz + 1;
  const int total = wx * wy * wz;
  for (int c0 = 0; c0 < total; c0 += 32) {   // (lane = index within the query's half-wave)
    const int c = c0 + lane;
    if (c < total) {
      const int cx = lx + c % wx, cy = ly + (c / wx) % wy, cz = lz + c / (wx * wy);
      const size_t cell = ((size_t)cz * g.ny + cy) * g.nx + cx;
      // the first two items of the bucket are fetched together with its fill count (three independent loads
      // in flight; most cells hold 0-2 nodes), the rest only when the count says they exist
      const GridItem* items = g.items + cell * g.bk;
      GridItem i0, i1;
      int m;
      if (SPECULATE) {   // node grid: most cells near a query hold a node or two
        i0 = items[0];
        i1 = g.bk > 1 ? items[1] : i0;
        m = g.cnt[cell];
      } else {           // the round's own grid is nearly empty: look at the count first
        m = g.cnt[cell];
        if (m > 0) i0 = items[0];
        if (m > 1) i1 = items[1];
      }
      if (m > g.bk) m = g.bk;
      if (m > 0) grid_test(i0, Q, q, st, qpos, cnt, hit_idx, hit_dist, cap);
      if (m > 1) grid_test(i1, Q, q, st, qpos, cnt, hit_idx, hit_dist, cap);
      if (m > 2) {
        GridItem it[6];
#pragma unroll
        for (int j = 0; j < 6; ++j)
          if (j + 2 < m) it[j] = items[j + 2];
#pragma unroll
        for (int j = 0; j < 6; ++j)
          if (j + 2 < m) grid_test(it[j], Q, q, st, qpos, cnt, hit_idx, hit_dist, cap);
      }
    }
  }
  int no = g.ovf_cnt[0];
  if (no > g.ovf_cap) no = g.ovf_cap;
  for (int j = lane; j < no; j += 32) grid_test(g.ovf[j], Q, q, st, qpos, cnt, hit_idx, hit_dist, cap);
}

// One half-wavefront (32 lanes) per query: the cells touched by the query ball's bounding box are dealt to the
// lanes (27 cells for the planner's radius), each lane walks its cell's bucket; all lanes then
// share the overflow list.  Same fp32 superset filter + exact fp64 re-test as the linear sweep.  With tg the
// same walk is repeated over the grid of the round's own samples (query i keeps the ids below its max_id).
__global__ __launch_bounds__(256) void k_grid_query(GridView g, GridView tg, NodeStoreView st,
                                                    const SweepQuery* __restrict__ queries,
                                                    const double* __restrict__ qpos, int nq, int32_t* __restrict__ cnt,
                                                    int32_t* __restrict__ hit_idx, double* __restrict__ hit_dist,
                                                    int cap, const int32_t* __restrict__ dev_n) {
  if (dev_n) {
    if (dev_n[1]) return;
    nq = dev_n[0];
  }
  const int q = blockIdx.x * 8 + (threadIdx.x >> 5);
  const int lane = threadIdx.x & 31;
  if (q >= nq) return;
  const SweepQuery Q = queries[q];
  if (!Q.active) return;
  const float rf = sqrtf(Q.r2f) * 1.000001f;
  grid_walk<true>(g, Q, q, lane, rf, st, qpos, cnt, hit_idx, hit_dist, cap);
  if (tg.cnt) grid_walk<false>(tg, Q, q, lane, rf, st, qpos, cnt, hit_idx, hit_dist, cap);
}

// ------------------------------------------------------------------ exact k nearest
// (TopK / topk_insert / topk_worst: kernels_dev.h)
__global__ __launch_bounds__(256) void k_knn_linear(NodeStoreView st, int n_store, const KnnQuery* __restrict__ queries, int nq,
                                                    int kcap, int32_t* __restrict__ idx, double* __restrict__ dist,
                                                    int32_t* __restrict__ cnt, double abs_eps) {
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (q >= nq) return;
  const KnnQuery Q = queries[q];
  const int k = Q.k < 64 ? Q.k : 64;
  const float qx = (float)Q.pos[0], qy = (float)Q.pos[1], qz = (float)Q.pos[2];
  const float qa = (float)Q.pos[3], qb = (float)Q.pos[4], qc = (float)Q.pos[5];
  TopK t{1.0e300, 0x7fffffff};
  int have = 0;
  const int lim = n_store < Q.max_id ? n_store : Q.max_id;
  for (int base = 0; base < lim; base += 64) {
    const int id = base + lane;
    // fp32 superset filter against the current k-th distance (slack: a few fp32 ulps of the coordinates)
    const double worst = topk_worst(t, k, have);
    bool cand = false;
    if (id < lim) {
      const float dx = st.x[id] - qx, dy = st.y[id] - qy, dz = st.z[id] - qz;
      const float da = wrapf(st.yaw[id] - qa), db = wrapf(st.pitch[id] - qb), dc = wrapf(st.roll[id] - qc);
      const float d6 = fmaf(dc, dc, fmaf(db, db, fmaf(da, da, fmaf(dz, dz, fmaf(dy, dy, dx * dx)))));
      if (d6 == d6) {                                 // (NaN placeholders never match)
        const double wi = (worst + abs_eps) * (1.0 + 1e-5);
        cand = worst >= 1.0e299 || (double)d6 <= wi * wi * 1.000001;
        if (cand && Q.tree >= 0 && st.tree[id] != Q.tree) cand = false;
      }
    }
    if (!__any(cand)) continue;
    double d = 1.0e300;
    if (cand) {
      double np[6];
      for (int c = 0; c < 6; ++c) np[c] = st.pos[6 * (size_t)id + c];
      d = dist6(np, Q.pos);
      cand = have < k || key_less(d, id, worst, 0x7fffffff);
    }
    topk_insert(t, lane, k, have, __ballot(cand), d, id);
  }
  if (lane == 0) cnt[q] = have;
  if (lane < have && lane < kcap) {
    idx[(size_t)q * kcap + lane] = t.id;
    dist[(size_t)q * kcap + lane] = t.d;
  }
}

// candidates of one group of up to 64 cells (lane = cell, m = its item count), flattened over the lanes
__device__ __forceinline__ void knn_cells(const GridView& g, int m, int cell, int lane, const KnnQuery& Q, const NodeStoreView& st,
                                          TopK& t, int k, int& have, bool mates, double mate_limit, int32_t* mate_out, int& n_mates, int mate_cap,
                                          double bound_d = 1.0e300, int bound_id = 0x7fffffff) {
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
      if (id < Q.max_id && (Q.tree < 0 || it.tree == Q.tree) && (mates ? id >= Q.mate_base : id < Q.mate_base)) {
        d = dist6(it.p, Q.pos);
        cand = true;
      }
    }
    if (mates) {
      cand = cand && d <= mate_limit;
      const unsigned long long mm = __ballot(cand);
      if (cand) {
        const int at = n_mates + __popcll(mm & ((1ULL << lane) - 1ULL));
        if (at < mate_cap) mate_out[at] = id;
      }
      n_mates += __popcll(mm);
    } else {
      const double worst = topk_worst(t, k, have);
      cand = cand && key_less(d, id, bound_d, bound_id) && (have < k || key_less(d, id, worst, 0x7fffffff));
      topk_insert(t, lane, k, have, __ballot(cand), d, id);
    }
  }
}

__global__ __launch_bounds__(256) void k_knn_grid(GridView g, GridView tg, NodeStoreView st, const KnnQuery* __restrict__ queries,
                                                  int nq, int kcap, int32_t* __restrict__ idx, double* __restrict__ dist,
                                                  int32_t* __restrict__ cnt, int32_t* __restrict__ mate_idx,
                                                  int32_t* __restrict__ mate_cnt, double cell_edge, double slack, int mate_cap,
                                                  int n_store) {
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (q >= nq) return;
  const KnnQuery Q = queries[q];
  const int k = Q.k < 64 ? Q.k : 64;
  TopK t{1.0e300, 0x7fffffff};
  int have = 0, n_mates = 0;
  const int cx = grid_coord((float)Q.pos[0], g.ox, g.inv_cell, g.nx), cy = grid_coord((float)Q.pos[1], g.oy, g.inv_cell, g.ny),
            cz = grid_coord((float)Q.pos[2], g.oz, g.inv_cell, g.nz);
  // shared overflow list first (usually empty)
  {
    int no = g.ovf_cnt[0];
    if (no > g.ovf_cap) no = g.ovf_cap;
    for (int base = 0; base < no; base += 64) {
      const int j = base + lane;
      bool cand = false;
      double d = 1.0e300;
      int id = 0x7fffffff;
      const double worst = topk_worst(t, k, have);   // (a shuffle: outside the divergent code below)
      if (j < no) {
        const GridItem it = g.ovf[j];
        id = it.id;
        if (id < Q.max_id && id < Q.mate_base && (Q.tree < 0 || it.tree == Q.tree)) {
          d = dist6(it.p, Q.pos);
          cand = have < k || key_less(d, id, worst, 0x7fffffff);
        }
      }
      topk_insert(t, lane, k, have, __ballot(cand), d, id);
    }
  }
  // shells of cells around the query's cell: shell r = the cube of half-width r minus the cube of half-width r-1.
  // After shell r every node within r * cell_edge (minus fp32 slack) of the query has been seen.  Only the shell's own
  // cells are enumerated (two caps of w x w cells, w - 2 rings of 8 r in between; until round 4 the loop ran over the
  // whole cube and masked its inside: sum of (2r+1)^3 instead of (2R+1)^3 - ten times the work at 40 shells), four
  // cells per lane and trip.  A query far from every node (RRT's random steering targets while the tree is small and
  // its cells already fine) would still visit the whole grid: once the cells asked for outnumber the store four to
  // one the query is answered by a sweep of the store instead (same keys, so the same k nearest).
  const int rmax = max(max(g.nx, g.ny), g.nz);
  long long scanned = 0;
  bool sweep = false;
  for (int r = 0; r <= rmax; ++r) {
    const int w = 2 * r + 1;
    const int ww = w * w, ring = 8 * r;
    const int total = r > 0 ? 2 * ww + (w - 2) * ring : 1;
    if (n_store > 0 && scanned + total > 4LL * n_store + 4096) { sweep = true; break; }
    scanned += total;
    for (int c0 = 0; c0 < total; c0 += 256) {
      int cell[4], m[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + u * 64 + lane;
        cell[u] = 0; m[u] = 0;
        if (c < total) {
          int ox, oy, oz;
          if (c < 2 * ww) {
            const int face = c >= ww ? 1 : 0, i = c - face * ww;
            ox = i % w - r; oy = i / w - r; oz = face ? r : -r;
          } else {
            const int cc = c - 2 * ww;
            const int layer = cc / ring, pp = cc - layer * ring;
            const int side = pp / (2 * r), t_ = pp - side * 2 * r;
            oz = -r + 1 + layer;
            ox = side == 0 ? -r + t_ : side == 1 ? r : side == 2 ? r - t_ : -r;
            oy = side == 0 ? -r : side == 1 ? -r + t_ : side == 2 ? r : r - t_;
          }
          const int x = cx + ox, y = cy + oy, z = cz + oz;
          if (x >= 0 && x < g.nx && y >= 0 && y < g.ny && z >= 0 && z < g.nz) {
            cell[u] = (z * g.ny + y) * g.nx + x;
            m[u] = g.cnt[cell[u]];
            if (m[u] > g.bk) m[u] = g.bk;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (__any(m[u] > 0)) knn_cells(g, m[u], cell[u], lane, Q, st, t, k, have, false, 0.0, nullptr, n_mates, 0);
    }
    const double covered = (double)r * cell_edge - slack;   // (cells are assigned from fp32 coordinates)
    if (have >= k && topk_worst(t, k, have) <= covered) break;
    if (cx - r <= 0 && cy - r <= 0 && cz - r <= 0 && cx + r >= g.nx - 1 && cy + r >= g.ny - 1 && cz + r >= g.nz - 1) break;
  }
  if (sweep) {   // (k_knn_linear's loop; what the shells found is found again)
    t = TopK{1.0e300, 0x7fffffff};
    have = 0;
    const float qx = (float)Q.pos[0], qy = (float)Q.pos[1], qz = (float)Q.pos[2];
    const float qa = (float)Q.pos[3], qb = (float)Q.pos[4], qc = (float)Q.pos[5];
    int lim = n_store < Q.max_id ? n_store : Q.max_id;
    if (Q.mate_base < lim) lim = Q.mate_base;
    for (int base = 0; base < lim; base += 64) {
      const int id = base + lane;
      const double worst = topk_worst(t, k, have);
      bool cand = false;
      if (id < lim) {
        const float dx = st.x[id] - qx, dy = st.y[id] - qy, dz = st.z[id] - qz;
        const float da = wrapf(st.yaw[id] - qa), db = wrapf(st.pitch[id] - qb), dc = wrapf(st.roll[id] - qc);
        const float d6 = fmaf(dc, dc, fmaf(db, db, fmaf(da, da, fmaf(dz, dz, fmaf(dy, dy, dx * dx)))));
        if (d6 == d6) {                                 // (NaN placeholders never match)
          const double wi = (worst + slack) * (1.0 + 1e-5);
          cand = worst >= 1.0e299 || (double)d6 <= wi * wi * 1.000001;
          if (cand && Q.tree >= 0 && st.tree[id] != Q.tree) cand = false;
        }
      }
      if (!__any(cand)) continue;
      double d = 1.0e300;
      if (cand) {
        double np[6];
        for (int c = 0; c < 6; ++c) np[c] = st.pos[6 * (size_t)id + c];
        d = dist6(np, Q.pos);
        cand = have < k || key_less(d, id, worst, 0x7fffffff);
      }
      topk_insert(t, lane, k, have, __ballot(cand), d, id);
    }
  }
  // the round's temporaries not farther than the k-th store node - all of the tree's while the k nearest are the
  // whole tree (then they are read straight from the temporary store entries: coalesced, no cube over the grid)
  if (tg.cnt && Q.mate_base < Q.max_id) {
    const bool all = Q.whole_tree != 0 || have < k;
    const double limit = all ? 1.0e300 : topk_worst(t, k, have);
    if (all) {
      for (int base = Q.mate_base; base < Q.max_id; base += 64) {
        const int id = base + lane;
        bool cand = false;
        if (id < Q.max_id) {
          const float x = st.x[id];
          cand = x == x && (Q.tree < 0 || st.tree[id] == Q.tree);     // (NaN: the sample fell outside the limits)
        }
        const unsigned long long mm = __ballot(cand);
        if (cand) {
          const int at = n_mates + __popcll(mm & ((1ULL << lane) - 1ULL));
          if (at < mate_cap) mate_idx[(size_t)q * mate_cap + at] = id;
        }
        n_mates += __popcll(mm);
      }
    } else {
      int rr = (int)((limit + slack) / cell_edge) + 1;
      if (rr > rmax) rr = rmax;
      const int w = 2 * rr + 1;
      const int total = w * w * w;
      for (int c0 = 0; c0 < total; c0 += 64) {
        const int c = c0 + lane;
        int cell = 0, m = 0;
        if (c < total) {
          const int x = cx + c % w - rr, y = cy + (c / w) % w - rr, z = cz + c / (w * w) - rr;
          if (x >= 0 && x < g.nx && y >= 0 && y < g.ny && z >= 0 && z < g.nz) {
            cell = (z * g.ny + y) * g.nx + x;
            const bool maybe = tg.occ ? ((tg.occ[cell >> 5] >> (cell & 31)) & 1u) != 0 : true;
            if (maybe) { m = tg.cnt[cell]; if (m > tg.bk) m = tg.bk; }
          }
        }
        if (__any(m > 0)) knn_cells(tg, m, cell, lane, Q, st, t, k, have, true, limit, mate_idx + (size_t)q * mate_cap, n_mates, mate_cap);
      }
      int no = tg.ovf_cnt[0];
      if (no > tg.ovf_cap) no = tg.ovf_cap;
      for (int base = 0; base < no; base += 64) {
        const int j = base + lane;
        bool cand = false;
        int id = 0x7fffffff;
        if (j < no) {
          const GridItem it = tg.ovf[j];
          id = it.id;
          if (id < Q.max_id && id >= Q.mate_base && (Q.tree < 0 || it.tree == Q.tree)) cand = dist6(it.p, Q.pos) <= limit;
        }
        const unsigned long long mm = __ballot(cand);
        if (cand) {
          const int at = n_mates + __popcll(mm & ((1ULL << lane) - 1ULL));
          if (at < mate_cap) mate_idx[(size_t)q * mate_cap + at] = id;
        }
        n_mates += __popcll(mm);
      }
    }
  }
  if (lane == 0) { cnt[q] = have; if (mate_cnt) mate_cnt[q] = n_mates; }
  if (lane < have && lane < kcap) {
    idx[(size_t)q * kcap + lane] = t.id;
    dist[(size_t)q * kcap + lane] = t.d;
  }
}

// The same question without a round grid (the RRT session's k-nearest queries, sffgpu_knn over an indexed store):
// ONE WORKGROUP per query.  A launch holds a few dozen queries - a wavefront each left the chip idle and the launch as
// slow as its slowest query (RRT*: 72 % of the GPU time).  The four wavefronts take the shell's batches of 64 cells
// in turn, each keeps what IT found in a list of its own; after every shell the lists are merged (rank = own index +
// entries of the other lists that sort before, by bisection in LDS), wave 0 keeps the merged k best, the others start
// empty again and only take candidates that beat the merged k-th key.  Same keys, same order: the same k nearest.
// The search of k_knn_grid_wg as a function of the whole workgroup (barriers inside; every wavefront must call it with the
// same arguments): the k nearest of Q among the grid's nodes, merged list in wave 0's lanes (t, G entries).  tree_nodes:
// nodes the queried tree holds (the search stops once a tree of no more than k nodes is in; INT_MAX: unknown).  Also the
// store part of SFF*'s k-nearest sets (k_star_knn_wg).
__device__ void knn_wg_search(const GridView& g, const NodeStoreView& st, const KnnQuery& Q, int k, double cell_edge, double slack,
                              int n_store, int sweep_only, int tree_nodes, TopK& t, int& G_out, int r_first = 0) {
  __shared__ double s_d[4][64];
  __shared__ int s_id[4][64];
  __shared__ int s_have[4];
  __shared__ double m_d[64];
  __shared__ int m_id[64];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  t = TopK{1.0e300, 0x7fffffff};
  int have = 0, n_mates = 0;
  double bd = 1.0e300;       // the merged k-th key (inf while fewer than k are known)
  int bi = 0x7fffffff, G = 0;
  auto merge = [&]() {
    s_d[wv][lane] = lane < have ? t.d : 1.0e300;
    s_id[wv][lane] = lane < have ? t.id : 0x7fffffff;
    if (lane == 0) s_have[wv] = have;
    __syncthreads();
    int rank = lane, total = 0;
    for (int o = 0; o < 4; ++o) {
      const int ho = s_have[o];
      total += ho;
      if (o != wv && lane < have) {
        int lo = 0, hi = ho;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (key_less(s_d[o][mid], s_id[o][mid], t.d, t.id)) lo = mid + 1; else hi = mid;
        }
        rank += lo;
      }
    }
    if (lane < have && rank < k) { m_d[rank] = t.d; m_id[rank] = t.id; }
    __syncthreads();
    G = total < k ? total : k;
    if (wv == 0) {
      t.d = lane < G ? m_d[lane] : 1.0e300;
      t.id = lane < G ? m_id[lane] : 0x7fffffff;
      have = G;
    } else {
      t = TopK{1.0e300, 0x7fffffff};
      have = 0;
    }
    if (G >= k) { bd = m_d[k - 1]; bi = m_id[k - 1]; }
    __syncthreads();
  };
  const int cx = grid_coord((float)Q.pos[0], g.ox, g.inv_cell, g.nx), cy = grid_coord((float)Q.pos[1], g.oy, g.inv_cell, g.ny),
            cz = grid_coord((float)Q.pos[2], g.oz, g.inv_cell, g.nz);
  if (!sweep_only) {   // shared overflow list (usually empty)
    int no = g.ovf_cnt[0];
    if (no > g.ovf_cap) no = g.ovf_cap;
    for (int base = wv * 64; base < no; base += 256) {
      const int j = base + lane;
      bool cand = false;
      double d = 1.0e300;
      int id = 0x7fffffff;
      const double worst = topk_worst(t, k, have);
      if (j < no) {
        const GridItem it = g.ovf[j];
        id = it.id;
        if (id < Q.max_id && id < Q.mate_base && (Q.tree < 0 || it.tree == Q.tree)) {
          d = dist6(it.p, Q.pos);
          cand = have < k || key_less(d, id, worst, 0x7fffffff);
        }
      }
      topk_insert(t, lane, k, have, __ballot(cand), d, id);
    }
  }
  const int rmax = sweep_only ? -1 : max(max(g.nx, g.ny), g.nz);
  long long scanned = 0;
  bool sweep = sweep_only != 0;
  int r_start = 0;
  bool done = false;
  if (r_first > 0 && !sweep_only) {
    // the cube of half-width r_first in one go (SFF*'s sets lie within about two steps of the sample): every cell's count
    // is on its way before anything is looked at, one merge instead of one per shell
    const int w = 2 * r_first + 1, total = w * w * w;
    scanned += total;
    for (int b0 = 0; b0 * 64 < total; b0 += 16) {
      int cell[4], m[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = (b0 + 4 * u + wv) * 64 + lane;
        cell[u] = 0; m[u] = 0;
        if (c < total) {
          const int x = cx + c % w - r_first, y = cy + (c / w) % w - r_first, z = cz + c / (w * w) - r_first;
          if (x >= 0 && x < g.nx && y >= 0 && y < g.ny && z >= 0 && z < g.nz) {
            cell[u] = (z * g.ny + y) * g.nx + x;
            m[u] = g.cnt[cell[u]];
            if (m[u] > g.bk) m[u] = g.bk;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (__any(m[u] > 0)) knn_cells(g, m[u], cell[u], lane, Q, st, t, k, have, false, 0.0, nullptr, n_mates, 0, bd, bi);
    }
    merge();
    const int r = r_first;
    const double covered = (double)r * cell_edge - slack;
    done = (tree_nodes <= k && G >= tree_nodes) || (G >= k && bd <= covered) ||
           (cx - r <= 0 && cy - r <= 0 && cz - r <= 0 && cx + r >= g.nx - 1 && cy + r >= g.ny - 1 && cz + r >= g.nz - 1);
    r_start = r_first + 1;
  }
  for (int r = r_start; r <= rmax && !done; ++r) {
    const int w = 2 * r + 1;
    const int ww = w * w, ring = 8 * r;
    const int total = r > 0 ? 2 * ww + (w - 2) * ring : 1;
    if (n_store > 0 && scanned + total > 4LL * n_store + 4096) { sweep = true; break; }
    scanned += total;
    for (int b0 = 0; b0 * 64 < total; b0 += 16) {
      int cell[4], m[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = (b0 + 4 * u + wv) * 64 + lane;
        cell[u] = 0; m[u] = 0;
        if (c < total) {
          int ox, oy, oz;
          if (c < 2 * ww) {
            const int face = c >= ww ? 1 : 0, i = c - face * ww;
            ox = i % w - r; oy = i / w - r; oz = face ? r : -r;
          } else {
            const int cc = c - 2 * ww;
            const int layer = cc / ring, pp = cc - layer * ring;
            const int side = pp / (2 * r), t_ = pp - side * 2 * r;
            oz = -r + 1 + layer;
            ox = side == 0 ? -r + t_ : side == 1 ? r : side == 2 ? r - t_ : -r;
            oy = side == 0 ? -r : side == 1 ? -r + t_ : side == 2 ? r : r - t_;
          }
          const int x = cx + ox, y = cy + oy, z = cz + oz;
          if (x >= 0 && x < g.nx && y >= 0 && y < g.ny && z >= 0 && z < g.nz) {
            cell[u] = (z * g.ny + y) * g.nx + x;
            m[u] = g.cnt[cell[u]];
            if (m[u] > g.bk) m[u] = g.bk;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (__any(m[u] > 0)) knn_cells(g, m[u], cell[u], lane, Q, st, t, k, have, false, 0.0, nullptr, n_mates, 0, bd, bi);
    }
    merge();
    if (tree_nodes <= k && G >= tree_nodes) break;          // (a tree with no more than k nodes: all of them are in)
    const double covered = (double)r * cell_edge - slack;   // (cells are assigned from fp32 coordinates)
    if (G >= k && bd <= covered) break;
    if (cx - r <= 0 && cy - r <= 0 && cz - r <= 0 && cx + r >= g.nx - 1 && cy + r >= g.ny - 1 && cz + r >= g.nz - 1) break;
  }
  if (sweep) {   // (k_knn_linear's loop, 256 nodes per wavefront and trip; what the shells found is found again)
    t = TopK{1.0e300, 0x7fffffff};
    have = 0; G = 0; bd = 1.0e300; bi = 0x7fffffff;
    const float qx = (float)Q.pos[0], qy = (float)Q.pos[1], qz = (float)Q.pos[2];
    const float qa = (float)Q.pos[3], qb = (float)Q.pos[4], qc = (float)Q.pos[5];
    int lim = n_store < Q.max_id ? n_store : Q.max_id;
    if (Q.mate_base < lim) lim = Q.mate_base;
    for (int base = wv * 256; base < lim; base += 1024) {
      float d6[4];
      bool in[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int id = base + u * 64 + lane;
        in[u] = id < lim;
        const int ic = in[u] ? id : lim - 1;
        const float dx = st.x[ic] - qx, dy = st.y[ic] - qy, dz = st.z[ic] - qz;
        const float da = wrapf(st.yaw[ic] - qa), db = wrapf(st.pitch[ic] - qb), dc = wrapf(st.roll[ic] - qc);
        d6[u] = fmaf(dc, dc, fmaf(db, db, fmaf(da, da, fmaf(dz, dz, fmaf(dy, dy, dx * dx)))));
        if (in[u] && Q.tree >= 0 && st.tree[ic] != Q.tree) in[u] = false;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int id = base + u * 64 + lane;
        const double worst = topk_worst(t, k, have);
        bool cand = false;
        if (in[u] && d6[u] == d6[u]) {                  // (NaN placeholders never match)
          const double wi = (worst + slack) * (1.0 + 1e-5);
          cand = worst >= 1.0e299 || (double)d6[u] <= wi * wi * 1.000001;
        }
        if (!__any(cand)) continue;
        double d = 1.0e300;
        if (cand) {
          double np[6];
          for (int c = 0; c < 6; ++c) np[c] = st.pos[6 * (size_t)id + c];
          d = dist6(np, Q.pos);
          cand = have < k || key_less(d, id, worst, 0x7fffffff);
        }
        topk_insert(t, lane, k, have, __ballot(cand), d, id);
      }
    }
    merge();
  }
  G_out = G;
}

__global__ __launch_bounds__(256) void k_knn_grid_wg(GridView g, NodeStoreView st, const KnnQuery* __restrict__ queries, int nq, int kcap,
                                                     int32_t* __restrict__ idx, double* __restrict__ dist, int32_t* __restrict__ cnt,
                                                     double cell_edge, double slack, int n_store, int sweep_only) {
  const int q = blockIdx.x, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (q >= nq) return;
  const KnnQuery Q = queries[q];
  const int k = Q.k < 64 ? Q.k : 64;
  TopK t{1.0e300, 0x7fffffff};
  int G = 0;
  knn_wg_search(g, st, Q, k, cell_edge, slack, n_store, sweep_only, 0x7fffffff, t, G);
  if (wv == 0) {
    if (lane == 0) cnt[q] = G;
    if (lane < G && lane < kcap) {
      idx[(size_t)q * kcap + lane] = t.id;
      dist[(size_t)q * kcap + lane] = t.d;
    }
  }
}


// ------------------------------------------------------------------ SFF*: k nearest of an accepted sample, one WORKGROUP per sample
// k_star_knn (devstar.hip) with the store part done by the whole workgroup (round 5): the four wavefronts take a shell's
// batches of cells in turn and merge their lists after every shell (knn_wg_search, the RRT session's k-nearest search) -
// one lone wavefront per accepted sample was 44 us per sample on configs[4] and 80 for the tenth of them that walk shells,
// and the launch is as long as its slowest sample.  Wave 0 then adds the round's earlier accepted samples and writes the
// members and their toucher lists exactly as k_star_knn does.  Same keys: the same sets.
#define SKW_MATE_U 16
#ifndef SKW_FIRST
#define SKW_FIRST 1
#endif
__global__ __launch_bounds__(256) void k_star_knn_wg(ResolveArgs A, GridView g, GridView tg, NodeStoreView st, double cell_edge,
                                                     double slack, int R0) {
  __shared__ int s_mate[64 * SKW_MATE_U];
  const DevForestView& f = A.f;
  const StarView& S = A.S;
  const DevCtrl* c = f.ctrl;
  const int n = c->app_n;
  const int n_acc = S.hdr[0];                      // (k_commit: the accepted samples, S.acc_sample[rank])
  const int N0 = c->app_N0, Tb = f.temp_base;
  const unsigned ep = (unsigned)c->epoch;
  const int i_first = S.acc_sample[blockIdx.x];    // (asked for with the header words; the grid never exceeds the list's size)
  if (blockIdx.x == 0 && threadIdx.x < SFFK_STAR_PASSES) S.changed[threadIdx.x] = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) { S.changed[SFFK_STAR_BAR] = 0; S.hdr[STAR_PASSES_RUN] = 0; S.hdr[STAR_CONVERGED] = 0; }   // (k_star_tail)
  for (int t = blockIdx.x * 256 + threadIdx.x; t < SFFK_STAR_PASSES * SFFK_SUBLISTS * SFFK_STAR_SUB; t += gridDim.x * 256) S.sub[t] = 0;
  if (n <= 0) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int r = blockIdx.x; r < n_acc; r += gridDim.x) {
    const int i = r == (int)blockIdx.x ? i_first : S.acc_sample[r];
    const unsigned long long dbg_t0 = S.dbg ? wall_clock64() : 0ULL;
    // k = (size_t)(2e log10(#nodes)) with the nodes accepted before this sample counted in (src/forest.h:309)
    const int Nn = N0 + r;
    const int k_ref = __popcll(__ballot(lane > 0 && lane <= SFFK_STAR_KMAX + 1 && S.ktab[lane] <= Nn));
    const int self = Tb + i;
    const int mine = st.tree[self];
    double qp[6];
    for (int q = 0; q < 6; ++q) qp[q] = A.newpos[6 * (size_t)i + q];
    // (wave 0: what the end needs, asked for beside the search - see k_star_knn)
    int sidv[SKW_MATE_U], trv[SKW_MATE_U];
    const bool mates_small = wave == 0 && r > 0 && r <= 64 * SKW_MATE_U;
#pragma unroll
    for (int u = 0; u < SKW_MATE_U; ++u) { const int rq = 64 * u + lane; sidv[u] = (mates_small && rq < r) ? Tb + S.acc_sample[rq] : -1; }
    const int ex0 = A.parent[i];
    const double pd0 = A.pdist[i];
    const int tcnt = S.tree_cnt[16 * mine];
    if (wave == 0) {   // the sample's edge slots: nothing asked for yet
      const size_t s0 = ((size_t)i * SFFK_STAR_KC + lane) * 2;
      S.ew[s0] = 0; S.ew[s0 + 1] = 0;
    }
    if (k_ref > SFFK_STAR_KMAX) {   // (a node count beyond what the member slots are sized for: host path)
      if (threadIdx.x == 0) atomicOr(S.hdr + 4, 1);
      continue;
    }
    const int k = k_ref;
#pragma unroll
    for (int u = 0; u < SKW_MATE_U; ++u) trv[u] = sidv[u] >= 0 ? st.tree[sidv[u]] : -1;
    const double dr0 = f.d_root[ex0];
    TopK t{1.0e300, 0x7fffffff};
    int have = 0;
    if (k > 0) {
      KnnQuery Q;
      for (int q = 0; q < 6; ++q) Q.pos[q] = qp[q];
      Q.tree = mine; Q.max_id = N0; Q.k = k; Q.mate_base = 0x7fffffff; Q.whole_tree = 0; Q.pad_ = 0;
      // (the first shells in one go - up to SKW_FIRST: beyond that the search usually ends before the cube does - then shell by shell)
      knn_wg_search(g, st, Q, k, cell_edge, slack, N0, 0, tcnt, t, have, R0 < SKW_FIRST ? R0 : SKW_FIRST);
    }
    const unsigned long long dbg_t1 = S.dbg ? wall_clock64() : 0ULL;
    unsigned long long dbg_t2 = 0ULL;
    if (wave == 0) {
      const int cx = grid_coord((float)qp[0], g.ox, g.inv_cell, g.nx), cy = grid_coord((float)qp[1], g.oy, g.inv_cell, g.ny),
                cz = grid_coord((float)qp[2], g.oz, g.inv_cell, g.nz);
      const int rmax = max(max(g.nx, g.ny), g.nz);
      if (k > 0) {
    // ---- the samples accepted earlier in this round (ranks below this one's, k_commit's list) of the same tree: not
    // farther than the k-th store node - or, while the store holds fewer than k nodes of the tree, all of them
    if (r > 0) {
      const bool all = have < k;
      const double limit = all ? 1.0e300 : topk_worst(t, k, have);
      if (r <= 64 * SKW_MATE_U) {
        // k_commit's list of the accepted samples: every rank's sample and tree are requested up front (two trips to
        // memory whatever the length), the few of the same tree are compacted in LDS and measured a batch at a time
        int* ml = s_mate;
        int nm = 0;
#pragma unroll
        for (int u = 0; u < SKW_MATE_U; ++u) {
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
  if (S.dbg) dbg_t2 = wall_clock64();
  // ---- the members: ids, distances, toucher lists
  const int cnt = have;
  const bool mem = lane < cnt;
  int node = -1;
  if (mem) {
    if (t.id < N0) node = t.id;
    else { int rk; star_accepted(f, t.id - Tb, rk); node = N0 + rk; }
  }
  const size_t p = (size_t)i * SFFK_STAR_KC + lane;
  S.prop[p] = __longlong_as_double(0x7ff0000000000000LL);
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
    if (S.dbg && k > 0) {   // SFFGPU_PROFILE: samples | ticks: search (four wavefronts), earlier samples, lists | longest sample
      const unsigned long long t3 = wall_clock64();
      atomicAdd(S.dbg + 0, 1ULL); atomicAdd(S.dbg + 1, dbg_t1 - dbg_t0); atomicAdd(S.dbg + 3, dbg_t2 - dbg_t1); atomicAdd(S.dbg + 4, t3 - dbg_t2);
      atomicMax(S.dbg + 7, t3 - dbg_t0);
    }
    }
    }
    __syncthreads();   // (the search's LDS and s_mate are the next sample's)
  }
}
void launch_star_knn_wg(hipStream_t s, const ResolveArgs& a, const GridView& g, const GridView& tg, const NodeStoreView& st,
                        double cell_edge, double slack, int n_bound, int R0) {
  int blocks = n_bound < 2048 ? n_bound : 2048;   // (a workgroup per accepted sample; more than 2 048 of them loop)
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_star_knn_wg, dim3(blocks), dim3(256), 0, s, a, g, tg, st, cell_edge, slack, R0);
}

__global__ __launch_bounds__(256) void k_set_tree(int32_t* __restrict__ tree_col, const int32_t* __restrict__ ids, int n,
                                                  int32_t value) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) tree_col[ids[i]] = value;
}

// ------------------------------------------------------------------ collision: shared pieces
struct WaveStack {
  int32_t* s;  // LDS, STACK_CAP entries per wave
  int sp;      // wave-uniform
  int32_t* hash;   // LDS, TG_HASH entries per wave (triangle-grid path)
};
#define STACK_CAP 256
// candidate triangles of an item are staged in LDS 64 at a time: box (6) + plane (5) + vertices (9) + the robot's
// extent along the normal (2) doubles each.
// One lane fetches one candidate (20 independent loads in flight) instead of every lane waiting on each candidate's
// data in turn: the exact kernel is bound by exactly that latency (83 % of its wave cycles are parked on memory,
// profiles/r2g_sq_summary.json).
#define STAGE_TRI 22
#ifndef STAGE_N
#define STAGE_N 64      // candidates per staging step (32 with three waves per SIMD measured slower: 59 vs 53 us)
#endif
#define STAGE_DOUBLES (STAGE_N * STAGE_TRI)
#define TG_HASH 512   // per-wave LDS hash set: a triangle listed by several cells of a query enters the candidates once
__device__ __forceinline__ void stage_candidates(const EnvView& env, const int32_t* cand, int k0, int kc, int lane, double* stage) {
  if (env.cand) {
    // packed records: lane = one 16-byte unit of one candidate's record, consecutive lanes read consecutive units
    const int units = kc * (STAGE_TRI / 2);
    for (int u = lane; u < units; u += 64) {
      const int k = u / (STAGE_TRI / 2), part = u - k * (STAGE_TRI / 2);
      const double2 v = reinterpret_cast<const double2*>(env.cand + (size_t)STAGE_TRI * (size_t)cand[k0 + k])[part];
      stage[k * STAGE_TRI + 2 * part] = v.x;
      stage[k * STAGE_TRI + 2 * part + 1] = v.y;
    }
    __builtin_amdgcn_wave_barrier();
    return;
  }
  if (lane < kc) {
    const int t = cand[k0 + lane];
    double* o = stage + lane * STAGE_TRI;
    const double* b = env.tri_box + 6 * (size_t)t;
    const double* pl = env.tri_plane + 5 * (size_t)t;
    const double* tr = env.tri + 9 * (size_t)t;
    for (int q = 0; q < 6; ++q) o[q] = b[q];
    for (int q = 0; q < 5; ++q) o[6 + q] = pl[q];
    for (int q = 0; q < 9; ++q) o[11 + q] = tr[q];
    o[20] = env.tri_ext ? env.tri_ext[2 * (size_t)t] : -1e300;      // (no extents: the plane-side cull never fires)
    o[21] = env.tri_ext ? env.tri_ext[2 * (size_t)t + 1] : 1e300;
  }
  __builtin_amdgcn_wave_barrier();
}

// push the set lanes of `mask`: entry = code_base + lane
__device__ __forceinline__ void push_mask(WaveStack& st, unsigned long long mask, int code_base, int lane) {
  int before = __popcll(mask & ((1ULL << lane) - 1ULL));
  if ((mask >> lane) & 1ULL) {
    int at = st.sp + before;
    if (at < STACK_CAP) st.s[at] = code_base + lane;
  }
  st.sp += __popcll(mask);
  if (st.sp > STACK_CAP) st.sp = STACK_CAP;  // overflow is reported by the caller's check
}

__device__ __forceinline__ bool box_hit(const double* lo, const double* hi, const double* qlo, const double* qhi) {
  return !(lo[0] > qhi[0] || qlo[0] > hi[0] || lo[1] > qhi[1] || qlo[1] > hi[1] || lo[2] > qhi[2] || qlo[2] > hi[2]);
}

// Traverse the 64-ary hierarchy with query box [qlo,qhi]; collects overlapping leaf triangles
// (indices into the leaf-ordered env arrays) into cand[] (LDS, cap entries).  Returns the
// number found; *overflow set when a stack or the list ran over.
// candidates of a query box from the triangle grid: lanes = cells of the box (64 at a time), their CSR ranges are
// flattened over the lanes (lane = list entry), every triangle id goes through the wave's LDS hash set so that it
// enters cand[] once however many cells list it
__device__ int collect_from_grid(const EnvView& env, const double* qlo, const double* qhi, int lane, int32_t* hash,
                                 int32_t* cand, int cap, bool* overflow) {
  for (int k = lane; k < TG_HASH; k += 64) hash[k] = 0;
  __builtin_amdgcn_wave_barrier();
  int c_lo[3], c_n[3];
  for (int a = 0; a < 3; ++a) {
    double flo = (qlo[a] - env.tg_org[a]) * env.tg_inv, fhi = (qhi[a] - env.tg_org[a]) * env.tg_inv;
    int lo = flo < 0 ? 0 : (flo >= env.tg_n[a] ? env.tg_n[a] - 1 : (int)flo);
    int hi = fhi < 0 ? 0 : (fhi >= env.tg_n[a] ? env.tg_n[a] - 1 : (int)fhi);
    if (!(flo == flo)) lo = 0;
    if (!(fhi == fhi)) hi = env.tg_n[a] - 1;
    c_lo[a] = lo;
    c_n[a] = hi - lo + 1;
  }
  const int total_cells = c_n[0] * c_n[1] * c_n[2];
  int n_cand = 0;
  *overflow = false;
  for (int c0 = 0; c0 < total_cells; c0 += 64) {
    const int c = c0 + lane;
    int start = 0, m = 0;
    if (c < total_cells) {
      const int x = c_lo[0] + c % c_n[0], y = c_lo[1] + (c / c_n[0]) % c_n[1], z = c_lo[2] + c / (c_n[0] * c_n[1]);
      const size_t cell = ((size_t)z * env.tg_n[1] + y) * env.tg_n[0] + x;
      start = env.tg_start[cell];
      m = env.tg_start[cell + 1] - start;
    }
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
      const int src_start = __shfl(start, lo);
      const int slot = jj - (__shfl(inc, lo) - __shfl(m, lo));
      bool fresh = false;
      int t = 0;
      if (j < total) {
        t = env.tg_list[src_start + slot];
        // the triangle's own box against the query box (it touches the cell, not necessarily the query)
        const double* b = env.tri_box + 6 * (size_t)t;
        if (box_hit(b, b + 3, qlo, qhi)) {
          unsigned h = ((unsigned)t * 2654435761u) >> 23;
          while (true) {
            h &= TG_HASH - 1;
            const int old = atomicCAS(&hash[h], 0, t + 1);
            if (old == 0) { fresh = true; break; }
            if (old == t + 1) break;
            ++h;
          }
        }
      }
      const unsigned long long fm = __ballot(fresh);
      if (fresh) {
        const int at = n_cand + __popcll(fm & ((1ULL << lane) - 1ULL));
        if (at < cap) cand[at] = t;
      }
      n_cand += __popcll(fm);
      if (n_cand > cap || n_cand > TG_HASH / 2) { *overflow = true; return n_cand > cap ? cap : n_cand; }
    }
  }
  return n_cand;
}

__device__ int collect_candidates(const EnvView& env, const double* qlo, const double* qhi, int lane, WaveStack& st,
                                  int32_t* cand, int cap, bool* overflow) {
  if (env.tg_start) return collect_from_grid(env, qlo, qhi, lane, st.hash, cand, cap, overflow);
  int n_cand = 0;
  st.sp = 0;
  *overflow = false;
  if (env.n_levels == 1) {
    // small environments (<= 4096 triangles): the top level IS the list of leaf groups.  Walk the hit groups
    // straight off the ballot mask and fetch the next group's triangle boxes while the current ones are tested,
    // so the dependent loads of consecutive groups overlap (the general stack walk below pays them one by one).
    const int cnt = env.level_count[0];
    bool h = false;
    if (lane < cnt) {
      const double* b = env.level_box[0] + 6 * (size_t)lane;
      h = box_hit(b, b + 3, qlo, qhi);
    }
    unsigned long long groups = __ballot(h);
    double cur[6] = {0, 0, 0, 0, 0, 0}, nxt[6] = {0, 0, 0, 0, 0, 0};
    auto fetch = [&](int g, double* out) -> bool {
      const int t = g * 64 + lane;
      if (t >= env.n_tri) return false;
      const double* b = env.tri_box + 6 * (size_t)t;
      for (int k = 0; k < 6; ++k) out[k] = b[k];
      return true;
    };
    bool have_cur = false;
    int g = -1;
    if (groups) {
      g = __ffsll((long long)groups) - 1;
      groups &= groups - 1;
      have_cur = fetch(g, cur);
    }
    while (g >= 0) {
      int gn = -1;
      bool have_nxt = false;
      if (groups) {
        gn = __ffsll((long long)groups) - 1;
        groups &= groups - 1;
        have_nxt = fetch(gn, nxt);
      }
      const bool hit = have_cur && box_hit(cur, cur + 3, qlo, qhi);
      const unsigned long long m = __ballot(hit);
      if (m) {
        const int before = __popcll(m & ((1ULL << lane) - 1ULL));
        if (hit && n_cand + before < cap) cand[n_cand + before] = g * 64 + lane;
        n_cand += __popcll(m);
        if (n_cand > cap) { *overflow = true; n_cand = cap; }
      }
      for (int k = 0; k < 6; ++k) cur[k] = nxt[k];
      have_cur = have_nxt;
      g = gn;
    }
    return n_cand;
  }
  // top level: up to 64 boxes
  {
    const int L = env.n_levels - 1;
    const int cnt = env.level_count[L];
    bool h = false;
    if (lane < cnt) {
      const double* b = env.level_box[L] + 6 * (size_t)lane;
      h = box_hit(b, b + 3, qlo, qhi);
    }
    unsigned long long m = __ballot(h);
    push_mask(st, m, (L << 24), lane);
  }
  while (st.sp > 0) {
    int code = st.s[st.sp - 1];
    st.sp -= 1;
    const int L = code >> 24, idx = code & 0xFFFFFF;
    if (L == 0) {
      // leaf group: triangles [64 idx, 64 idx + 64)
      const int t = idx * 64 + lane;
      bool h = false;
      if (t < env.n_tri) {
        const double* b = env.tri_box + 6 * (size_t)t;
        h = box_hit(b, b + 3, qlo, qhi);
      }
      unsigned long long m = __ballot(h);
      if (m) {
        int before = __popcll(m & ((1ULL << lane) - 1ULL));
        if (h && n_cand + before < cap) cand[n_cand + before] = t;
        n_cand += __popcll(m);
        if (n_cand > cap) { *overflow = true; n_cand = cap; }
      }
    } else {
      const int child = idx * 64 + lane;
      const int cnt = env.level_count[L - 1];
      bool h = false;
      if (child < cnt) {
        const double* b = env.level_box[L - 1] + 6 * (size_t)child;
        h = box_hit(b, b + 3, qlo, qhi);
      }
      unsigned long long m = __ballot(h);
      if (st.sp + __popcll(m) > STACK_CAP) *overflow = true;
      push_mask(st, m, ((L - 1) << 24) + idx * 64, lane);
    }
  }
  return n_cand;
}

// conservative plane test: the robot's bounding sphere (centre c, radius rr) lies strictly on one
// side of the triangle's plane, by far more than the rounding noise of the exact test's normal
// axis, so the 17-axis test would separate the pair on that axis anyway
__device__ __forceinline__ bool plane_clear(const double* pl, const double* c, double rr) {
  double s = (pl[0] * c[0] + pl[1] * c[1]) + pl[2] * c[2] - pl[3];
  double as = s < 0 ? -s : s;
  double slack = 1e-9 * (pl[4] * (fabs(c[0]) + fabs(c[1]) + fabs(c[2]) + 1.0) + fabs(pl[3]));
  return as > rr * pl[4] * (1.0 + 1e-9) + slack;
}

// conservative sphere / triangle test: true when the closest point of triangle T to the centre c is
// farther than rr (plus slack).  The robot lies inside that sphere, so no robot triangle can touch T
// and the exact 17-axis test would find a separating axis; the slack (1e-9 relative) dwarfs rounding.
__device__ __forceinline__ bool tri_far(const double* T, const double* c, double rr) {
  double ab[3], ac[3], ap[3];
  for (int i = 0; i < 3; ++i) { ab[i] = T[3 + i] - T[i]; ac[i] = T[6 + i] - T[i]; ap[i] = c[i] - T[i]; }
  double d1 = dot(ab, ap), d2 = dot(ac, ap);
  double q[3];
  bool done = false;
  if (d1 <= 0 && d2 <= 0) { for (int i = 0; i < 3; ++i) q[i] = T[i]; done = true; }
  double bp[3], cp[3], d3 = 0, d4 = 0, d5 = 0, d6 = 0;
  if (!done) {
    for (int i = 0; i < 3; ++i) bp[i] = c[i] - T[3 + i];
    d3 = dot(ab, bp); d4 = dot(ac, bp);
    if (d3 >= 0 && d4 <= d3) { for (int i = 0; i < 3; ++i) q[i] = T[3 + i]; done = true; }
  }
  if (!done) {
    double vc = d1 * d4 - d3 * d2;
    if (vc <= 0 && d1 >= 0 && d3 <= 0) {
      double v = d1 / (d1 - d3);
      for (int i = 0; i < 3; ++i) q[i] = T[i] + v * ab[i];
      done = true;
    }
  }
  if (!done) {
    for (int i = 0; i < 3; ++i) cp[i] = c[i] - T[6 + i];
    d5 = dot(ab, cp); d6 = dot(ac, cp);
    if (d6 >= 0 && d5 <= d6) { for (int i = 0; i < 3; ++i) q[i] = T[6 + i]; done = true; }
  }
  if (!done) {
    double vb = d5 * d2 - d1 * d6;
    if (vb <= 0 && d2 >= 0 && d6 <= 0) {
      double w = d2 / (d2 - d6);
      for (int i = 0; i < 3; ++i) q[i] = T[i] + w * ac[i];
      done = true;
    }
  }
  if (!done) {
    double va = d3 * d6 - d5 * d4;
    if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) {
      double w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
      for (int i = 0; i < 3; ++i) q[i] = T[3 + i] + w * (T[6 + i] - T[3 + i]);
      done = true;
    }
  }
  if (!done) {
    double vc = d1 * d4 - d3 * d2, vb = d5 * d2 - d1 * d6, va = d3 * d6 - d5 * d4;
    double den = va + vb + vc;
    if (!(den != 0)) return false;          // degenerate triangle: do not cull
    double v = vb / den, w = vc / den;
    for (int i = 0; i < 3; ++i) q[i] = T[i] + ab[i] * v + ac[i] * w;
  }
  double dd = 0, mag = 0;
  for (int i = 0; i < 3; ++i) { double e = c[i] - q[i]; dd += e * e; mag += fabs(c[i]) + fabs(q[i]); }
  if (!(dd == dd)) return false;            // NaN guard: never cull on garbage
  double lim = rr * (1.0 + 1e-9) + 1e-9 * (mag + 1.0);
  return dd > lim * lim;
}
// squared distance of c to the closest point of triangle T and the magnitude of the operands (tri_far's own computation,
// for callers that compare ONE distance with several radii); false: degenerate triangle or NaN - never cull
__device__ __forceinline__ bool tri_dist2(const double* T, const double* c, double& dd_out, double& mag_out) {
  double ab[3], ac[3], ap[3];
  for (int i = 0; i < 3; ++i) { ab[i] = T[3 + i] - T[i]; ac[i] = T[6 + i] - T[i]; ap[i] = c[i] - T[i]; }
  double d1 = dot(ab, ap), d2 = dot(ac, ap);
  double q[3];
  bool done = false;
  if (d1 <= 0 && d2 <= 0) { for (int i = 0; i < 3; ++i) q[i] = T[i]; done = true; }
  double bp[3], cp[3], d3 = 0, d4 = 0, d5 = 0, d6 = 0;
  if (!done) {
    for (int i = 0; i < 3; ++i) bp[i] = c[i] - T[3 + i];
    d3 = dot(ab, bp); d4 = dot(ac, bp);
    if (d3 >= 0 && d4 <= d3) { for (int i = 0; i < 3; ++i) q[i] = T[3 + i]; done = true; }
  }
  if (!done) {
    double vc = d1 * d4 - d3 * d2;
    if (vc <= 0 && d1 >= 0 && d3 <= 0) {
      double v = d1 / (d1 - d3);
      for (int i = 0; i < 3; ++i) q[i] = T[i] + v * ab[i];
      done = true;
    }
  }
  if (!done) {
    for (int i = 0; i < 3; ++i) cp[i] = c[i] - T[6 + i];
    d5 = dot(ab, cp); d6 = dot(ac, cp);
    if (d6 >= 0 && d5 <= d6) { for (int i = 0; i < 3; ++i) q[i] = T[6 + i]; done = true; }
  }
  if (!done) {
    double vb = d5 * d2 - d1 * d6;
    if (vb <= 0 && d2 >= 0 && d6 <= 0) {
      double w = d2 / (d2 - d6);
      for (int i = 0; i < 3; ++i) q[i] = T[i] + w * ac[i];
      done = true;
    }
  }
  if (!done) {
    double va = d3 * d6 - d5 * d4;
    if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) {
      double w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
      for (int i = 0; i < 3; ++i) q[i] = T[3 + i] + w * (T[6 + i] - T[3 + i]);
      done = true;
    }
  }
  if (!done) {
    double vc = d1 * d4 - d3 * d2, vb = d5 * d2 - d1 * d6, va = d3 * d6 - d5 * d4;
    double den = va + vb + vc;
    if (!(den != 0)) return false;          // degenerate triangle
    double v = vb / den, w = vc / den;
    for (int i = 0; i < 3; ++i) q[i] = T[i] + ab[i] * v + ac[i] * w;
  }
  double dd = 0, mag = 0;
  for (int i = 0; i < 3; ++i) { double e = c[i] - q[i]; dd += e * e; mag += fabs(c[i]) + fabs(q[i]); }
  if (!(dd == dd)) return false;            // NaN guard
  dd_out = dd; mag_out = mag;
  return true;
}
__device__ __forceinline__ bool dist2_far(double dd, double mag, double rr) {   // tri_far's verdict from tri_dist2's outputs
  const double lim = rr * (1.0 + 1e-9) + 1e-9 * (mag + 1.0);
  return dd > lim * lim;
}

// Clearance bits: true when a robot whose MODEL ORIGIN is at c provably touches nothing in any rotation (the
// exact test would find a separating axis for every pair), so the traversal and the exact tests can be skipped.
// Points outside the grid are farther than the build threshold from the environment's box.
__device__ __forceinline__ bool surely_clear(const EnvView& env, const double* c) {
  if (!env.clear_bits) return false;
  const double fx = (c[0] - env.clear_org[0]) * env.clear_inv, fy = (c[1] - env.clear_org[1]) * env.clear_inv,
               fz = (c[2] - env.clear_org[2]) * env.clear_inv;
  if (!(fx == fx && fy == fy && fz == fz)) return false;
  if (fx < 0 || fy < 0 || fz < 0 || fx >= env.clear_n[0] || fy >= env.clear_n[1] || fz >= env.clear_n[2]) return true;
  const long long idx = ((long long)(int)fz * env.clear_n[1] + (int)fy) * env.clear_n[0] + (int)fx;
  return (env.clear_bits[idx >> 5] >> (idx & 31)) & 1u;
}
// the same for an UN-ROTATED robot (edge samples, src/problemStruct.h:157-165): the edge plane of the bits
__device__ __forceinline__ bool surely_clear_edge(const EnvView& env, const double* c) {
  if (!env.clear_bits_edge) return false;
  const double fx = (c[0] - env.clear_org[0]) * env.clear_inv, fy = (c[1] - env.clear_org[1]) * env.clear_inv,
               fz = (c[2] - env.clear_org[2]) * env.clear_inv;
  if (!(fx == fx && fy == fy && fz == fz)) return false;
  if (fx < 0 || fy < 0 || fz < 0 || fx >= env.clear_n[0] || fy >= env.clear_n[1] || fz >= env.clear_n[2]) return true;
  const long long idx = ((long long)(int)fz * env.clear_n[1] + (int)fy) * env.clear_n[0] + (int)fx;
  return (env.clear_bits_edge[idx >> 5] >> (idx & 31)) & 1u;
}

// Conservative triangle / axis-aligned box test (box = centre c + [blo, bhi], blo <= 0 <= bhi per axis): false only when a
// separating axis exists by more than the rounding noise (box axes, the triangle's normal, the nine edge x axis
// directions); NaNs compare false, so garbage never separates.
__device__ __forceinline__ bool tri_box_maybe(const double* T, const double* c, const double* blo, const double* bhi) {
  double bc[3], bh[3], v[3][3];
  for (int a = 0; a < 3; ++a) {
    bc[a] = c[a] + 0.5 * (blo[a] + bhi[a]);
    bh[a] = 0.5 * (bhi[a] - blo[a]);
    bh[a] += 1e-9 * (fabs(bc[a]) + bh[a] + 1.0);
  }
  for (int k = 0; k < 3; ++k)
    for (int a = 0; a < 3; ++a) v[k][a] = T[3 * k + a] - bc[a];
  for (int a = 0; a < 3; ++a) {
    const double lo = min3(v[0][a], v[1][a], v[2][a]), hi = max3(v[0][a], v[1][a], v[2][a]);
    if (lo > bh[a] || hi < -bh[a]) return false;
  }
  const double e[3][3] = {{v[1][0] - v[0][0], v[1][1] - v[0][1], v[1][2] - v[0][2]},
                          {v[2][0] - v[1][0], v[2][1] - v[1][1], v[2][2] - v[1][2]},
                          {v[0][0] - v[2][0], v[0][1] - v[2][1], v[0][2] - v[2][2]}};
  double vmax[3];
  for (int a = 0; a < 3; ++a) vmax[a] = max3(fabs(v[0][a]), fabs(v[1][a]), fabs(v[2][a]));
  auto separated = [&](const double* ax) -> bool {
    const double p0 = dot(ax, v[0]), p1 = dot(ax, v[1]), p2 = dot(ax, v[2]);
    const double aa[3] = {fabs(ax[0]), fabs(ax[1]), fabs(ax[2])};
    const double r = aa[0] * bh[0] + aa[1] * bh[1] + aa[2] * bh[2];
    const double slack = 1e-9 * (aa[0] * (vmax[0] + bh[0]) + aa[1] * (vmax[1] + bh[1]) + aa[2] * (vmax[2] + bh[2]));
    return min3(p0, p1, p2) > r + slack || max3(p0, p1, p2) < -(r + slack);
  };
  double n[3];
  cross(e[0], e[1], n);
  if (separated(n)) return false;
  for (int k = 0; k < 3; ++k) {
    const double a0[3] = {0.0, -e[k][2], e[k][1]}, a1[3] = {e[k][2], 0.0, -e[k][0]}, a2[3] = {-e[k][1], e[k][0], 0.0};
    if (separated(a0) || separated(a1) || separated(a2)) return false;
  }
  return true;
}

// Clearance bits by scatter from the triangles (round 5; the gather - every cell over every triangle group - took 57 ms
// for dense_3D's 134 M cells): one workgroup per triangle walks the cells whose centre lies within the sphere radius of the
// triangle's box, a thread takes one 32-cell word of one row, tests its cells and clears the blocked ones with one
// atomicAnd per plane.  TWO planes over the same cells:
//   pose plane  a robot whose model origin lies in the cell cannot touch the triangle in ANY rotation: the triangle is
//               farther from the cell centre than the bounding-sphere radius + half the cell diagonal (the old test), OR it
//               misses the cube of that radius around the cell (a cell is a box, not a ball: the sphere over-covers its faces);
//   edge plane  the same for the UN-ROTATED robot (edge samples carry no rotation, src/problemStruct.h:157-165): the
//               triangle misses cell + [robot box], inflated by the reach of a group of eight samples.
// A cell stays clear in a plane unless some triangle fails BOTH tests of that plane.
__global__ __launch_bounds__(256) void k_clear_scatter(EnvView env, ClearBuildArgs P, uint32_t* __restrict__ bits_pose,
                                                       uint32_t* __restrict__ bits_edge) {
  const int t = blockIdx.x;
  const double* T = env.tri + 9 * (size_t)t;
  const double* tb = env.tri_box + 6 * (size_t)t;
  const double h = 1.0 / env.clear_inv;
  const double reach = P.thr_edge > P.thr_pose ? P.thr_edge : P.thr_pose;
  int i0[3], i1[3];
  for (int a = 0; a < 3; ++a) {
    const double lo = (tb[a] - reach - env.clear_org[a]) * env.clear_inv - 0.5, hi = (tb[3 + a] + reach - env.clear_org[a]) * env.clear_inv - 0.5;
    if (!(lo == lo) || !(hi == hi)) { i0[a] = 0; i1[a] = env.clear_n[a] - 1; continue; }
    i0[a] = lo < 1.0 ? 0 : (lo >= (double)env.clear_n[a] ? env.clear_n[a] : (int)lo - 1);
    i1[a] = hi < -1.0 ? -1 : (hi + 1.0 >= (double)env.clear_n[a] ? env.clear_n[a] - 1 : (int)hi + 1);
    if (i1[a] < i0[a]) return;
  }
  const int nyr = i1[1] - i0[1] + 1;
  const long long rows = (long long)nyr * (i1[2] - i0[2] + 1);
  const int nwmax = ((i1[0] - i0[0]) >> 5) + 2;
  const long long tasks = rows * nwmax;
  // (blockIdx.y: the triangle's tasks dealt out to several workgroups - a few large triangles were the whole launch).
  // Round 6: a thread still owns one 32-cell word of a row, but it only FILTERS its cells (six flops each: farther from the
  // triangle's plane than the reach = untouched); the cells that pass are then dealt out over the wavefront's lanes, 64 at a
  // time, for the closest-point computation and the two box tests (hundreds of fp64 operations), and their verdicts come
  // back to the word's owner through LDS.  Before, every thread walked its 32 cells one after the other and a wavefront was
  // as slow as its busiest lane.
  __shared__ uint32_t s_mp[256], s_me[256];
  const int lane = threadIdx.x & 63, wbase = threadIdx.x & ~63;
  const double* pl = env.tri_plane ? env.tri_plane + 5 * (size_t)t : nullptr;
  const long long stride = 256LL * gridDim.y;
  for (long long task0 = (long long)blockIdx.y * 256; task0 < tasks; task0 += stride) {
    const long long task = task0 + threadIdx.x;
    long long w = 0, base = 0, c0 = 0, c1 = -1;
    int iy = 0, iz = 0;
    uint32_t cand = 0u;
    if (task < tasks) {
      const long long row = task / nwmax;
      const int k = (int)(task - row * nwmax);
      iy = i0[1] + (int)(row % nyr); iz = i0[2] + (int)(row / nyr);
      base = ((long long)iz * env.clear_n[1] + iy) * env.clear_n[0];
      const long long l0 = base + i0[0], l1 = base + i1[0];
      w = (l0 >> 5) + k;
      if (w <= (l1 >> 5)) {
        c0 = w * 32 > l0 ? w * 32 : l0; c1 = w * 32 + 31 < l1 ? w * 32 + 31 : l1;
        const double cy = env.clear_org[1] + ((double)iy + 0.5) * h, cz = env.clear_org[2] + ((double)iz + 0.5) * h;
        for (long long l = c0; l <= c1; ++l) {
          const double c[3] = {env.clear_org[0] + ((double)(l - base) + 0.5) * h, cy, cz};
          if (!(pl && plane_clear(pl, c, reach))) cand |= 1u << (int)(l & 31);
        }
      }
    }
    s_mp[threadIdx.x] = 0u; s_me[threadIdx.x] = 0u;
    // the wavefront's candidates, 64 at a time (wavefront-local: no barrier - the LDS words are this wavefront's own)
    int cnt = __popc(cand), inc = cnt;
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(inc, off);
      if (lane >= off) inc += o;
    }
    const int total = __shfl(inc, 63);
    __builtin_amdgcn_wave_barrier();
    for (int b0 = 0; b0 < total; b0 += 64) {
      const int j = b0 + lane;
      const int jj = j < total ? j : total - 1;
      int lo = 0, hi = 63;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (__shfl(inc, mid) > jj) hi = mid; else lo = mid + 1;
      }
      const int nth = jj - (__shfl(inc, lo) - __shfl(cnt, lo));   // which of the owner's candidates
      uint32_t m = (uint32_t)__shfl((int)cand, lo);
      for (int q = 0; q < nth; ++q) m &= m - 1u;
      const int bit = __ffs((int)m) - 1;
      const long long ow = __shfl((long long)w, lo), ob = __shfl((long long)base, lo);
      const int oy = __shfl(iy, lo), oz = __shfl(iz, lo);
      if (j < total) {
        const long long ix = ow * 32 + bit - ob;
        const double c[3] = {env.clear_org[0] + ((double)ix + 0.5) * h, env.clear_org[1] + ((double)oy + 0.5) * h,
                             env.clear_org[2] + ((double)oz + 0.5) * h};
        // (one closest-point computation, compared with both radii: tri_far's arithmetic)
        double dd = 0.0, mag = 0.0;
        const bool known = tri_dist2(T, c, dd, mag);
        if (!(known && dist2_far(dd, mag, reach))) {
          if (!(known && dist2_far(dd, mag, P.thr_pose)) && tri_box_maybe(T, c, P.pose_lo, P.pose_hi)) atomicOr(&s_mp[wbase + lo], 1u << bit);
          if (!(known && dist2_far(dd, mag, P.thr_edge)) && tri_box_maybe(T, c, P.edge_lo, P.edge_hi)) atomicOr(&s_me[wbase + lo], 1u << bit);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t mp = __hip_atomic_load(&s_mp[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const uint32_t me = __hip_atomic_load(&s_me[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (mp) atomicAnd(bits_pose + w, ~mp);
    if (me) atomicAnd(bits_edge + w, ~me);
  }
}

// ------------------------------------------------------------------ triangle grid build
// One thread per cell: the triangles whose (slightly inflated) cell box they touch - group boxes first, then the
// triangle's box, then the exact triangle / box test.  Pass 1 counts, pass 2 (after the host's prefix sum) writes ids.
__global__ __launch_bounds__(256) void k_tgrid_build(EnvView env, int32_t* __restrict__ cnt_or_start,
                                                     int32_t* __restrict__ list, int fill) {
  const long long n_cells = (long long)env.tg_n[0] * env.tg_n[1] * env.tg_n[2];
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n_cells) return;
  const int nx = env.tg_n[0], ny = env.tg_n[1];
  const long long iz = idx / ((long long)nx * ny), rem = idx - iz * (long long)nx * ny;
  const long long iy = rem / nx, ix = rem - iy * nx;
  const double h = 1.0 / env.tg_inv, eps = 1e-6 * h;
  const double lo[3] = {env.tg_org[0] + (double)ix * h - eps, env.tg_org[1] + (double)iy * h - eps, env.tg_org[2] + (double)iz * h - eps};
  const double hi[3] = {env.tg_org[0] + (double)(ix + 1) * h + eps, env.tg_org[1] + (double)(iy + 1) * h + eps,
                        env.tg_org[2] + (double)(iz + 1) * h + eps};
  int n = 0;
  int at = fill ? cnt_or_start[idx] : 0;
  const int n_groups = env.level_count[0];
  for (int g = 0; g < n_groups; ++g) {
    const double* gb = env.level_box[0] + 6 * (size_t)g;
    if (!box_hit(gb, gb + 3, lo, hi)) continue;
    const int t1 = g * 64 + 64 < env.n_tri ? g * 64 + 64 : env.n_tri;
    for (int t = g * 64; t < t1; ++t) {
      const double* b = env.tri_box + 6 * (size_t)t;
      if (!box_hit(b, b + 3, lo, hi)) continue;
      if (!tri_box_overlap(lo, hi, env.tri + 9 * (size_t)t)) continue;
      if (fill) list[at++] = t;
      ++n;
    }
  }
  if (!fill) cnt_or_start[idx] = n;
}
void launch_tgrid_build(hipStream_t s, const EnvView& env, int32_t* cnt_or_start, int32_t* list, bool fill) {
  const long long n_cells = (long long)env.tg_n[0] * env.tg_n[1] * env.tg_n[2];
  hipLaunchKernelGGL(k_tgrid_build, dim3((unsigned)((n_cells + 255) / 256)), dim3(256), 0, s, env, cnt_or_start, list, fill ? 1 : 0);
}

#ifdef SFFK_DEBUG_COUNTERS
__device__ unsigned long long g_dbg[16];
__device__ unsigned long long g_dbg_q[16];   // k_query_classify: sampled waves | ticks: scan, classify, cull, flushes | pairs, survivors, live
#define QDBG(i, x) do { if (qdbg_on) atomicAdd(&g_dbg_q[i], (unsigned long long)(x)); } while (0)
struct DbgAcc { unsigned long long v[16]; };
#define DBG_DECL DbgAcc dbg_acc = {{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}};
#define DBG_ARG , DbgAcc& dbg_acc
#define DBG_PASS , dbg_acc
#define DBG_T() wall_clock64()
#define DBG_ADD(i, x) dbg_acc.v[i] += (unsigned long long)(x)
#define DBG_FLUSH() do { if (lane == 0) for (int q_ = 0; q_ < 16; ++q_) if (dbg_acc.v[q_]) atomicAdd(&g_dbg[q_], dbg_acc.v[q_]); } while (0)
#else
#define DBG_DECL
#define DBG_ARG
#define DBG_PASS
#define DBG_T() 0ULL
#define DBG_ADD(i, x) do { } while (0)
#define DBG_FLUSH() do { } while (0)
#define QDBG(i, x) do { } while (0)
#endif

#ifdef SFFK_CI_TRACE
__device__ unsigned long long g_ci_trace[4096 * 8];   // one launch of k_collide_items: per wave {t0, item start, broad done, end, nc, samples, items, -}
__device__ unsigned int g_ci_launch;
__device__ unsigned long long g_ci_tmp[4096 * 2];     // per wave: broad phase done (clock), candidates of its last chunk
#endif
// ------------------------------------------------------------------ pose kernel
#define POSE_WAVES 4
#define CAND_CAP 256

__device__ __forceinline__ void pose_frame(const RobotView& rob, const double* __restrict__ pos6, int pose, double* p,
                                           double* R, double* c) {
  for (int k = 0; k < 6; ++k) p[k] = pos6[6 * (size_t)pose + k];
  if (p[3] == 0 && p[4] == 0 && p[5] == 0) {
    R[0] = R[4] = R[8] = 1; R[1] = R[2] = R[3] = R[5] = R[6] = R[7] = 0;
  } else {
    rotation(p, R);
  }
  xform(R, p, rob.center, c);
}

// ---- an item with many candidate triangles is shared by the wavefronts of its workgroup (round 5).  The narrow phase
// costs ~2.5 us per CANDIDATE whatever the number of samples (a serial loop of fp64 culls on one wavefront), and on
// building.obj a chunk near the walls has 35 and more of them: the exact kernels' launches were as long as their one or
// two worst items (k_collide_items 93 us on average / 317 at most per round, k_star_exact up to 403 us) while most
// wavefronts had long finished.  The owner publishes the item in LDS (ShareSlot) and the candidate list stays where its
// broad phase left it; owner and idle siblings take blocks of SHARE_BLK candidates off one 64-bit word (sequence << 32 |
// next candidate: a grab names the item it belongs to), run narrow_block on their own stage / queue buffers and
// atomicMin the first hit; the owner waits until every candidate is accounted for.  A wavefront helps only once its own
// items are done, and leaves when all wavefronts of the workgroup are (no barrier anywhere).
#ifndef SEG_WAVES
#define SEG_WAVES 4
#endif
// when an item is shared, and in blocks of how many candidates.  A chunk's narrow phase costs per (candidate, touching
// sample): on building.obj (a robot as wide as the corridors) 8-12 candidates x 35 samples ran 80-130 us, on dense_3D
// 6 candidates x 64 samples 15 us - shared from `SHARE_WORK` (candidates x masked samples) on; a pose from 8 candidates on
#ifndef SHARE_WORK
#define SHARE_WORK 192
#endif
#ifndef SHARE_POSE_MIN
#define SHARE_POSE_MIN 8
#endif
#define SHARE_IDLE (1 << 30)
__device__ __forceinline__ int share_block(int nc) { return nc >= 64 ? 4 : (nc >= 24 ? 2 : 1); }
struct ShareSlot {
  unsigned long long sk;     // (item sequence << 32) | next candidate to hand out (>= SHARE_IDLE: nothing to take)
  int nc, done_k, minhit, seg, chunk, kind;   // kind 0: chunk of an edge (a, b = its end points, mask); 1: pose (a = pose, b = centre, R)
  unsigned long long mask;
  double a[6], b[6], R[9];
};
struct ShareArea {
  ShareSlot slot[SEG_WAVES];
  int done_waves;            // wavefronts whose own items are finished
};
__device__ __forceinline__ unsigned long long lds_load64(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ int lds_load32(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ void share_init(ShareArea& sh) {   // (every thread of the workgroup, before its first barrier)
  if (threadIdx.x < SEG_WAVES) { sh.slot[threadIdx.x].sk = (unsigned long long)SHARE_IDLE; sh.slot[threadIdx.x].nc = 0; }
  if (threadIdx.x == 0) sh.done_waves = 0;
}

// a block of a pose item's candidate triangles (cand[k0 .. k0 + kc)) against the posed robot: lane = robot triangle
__device__ bool pose_block(const EnvView& env, const RobotView& rob, const double* rtri, double* stage, const int32_t* cand,
                           int k0, int kc, const double* p, const double* R, const double* c, double rr, int lane) {
  bool hit = false;
  stage_candidates(env, cand, k0, kc, lane, stage);
  for (int r0 = 0; r0 < rob.n_tri && !hit; r0 += 64) {
    const int r = r0 + lane;
    double Q[9];
    const bool have = r < rob.n_tri;
    if (have)
      for (int v = 0; v < 3; ++v) xform(R, p, rtri + 9 * r + 3 * v, Q + 3 * v);
    bool lane_hit = false;
    for (int k = 0; k < kc; ++k) {
      const double* sb = stage + k * STAGE_TRI;
      if (plane_clear(sb + 6, c, rr)) continue;  // wave-uniform
      if (have && !lane_hit) {
        if (tri_box_overlap(sb, sb + 3, Q)) lane_hit = sat17(sb + 11, Q);
      }
    }
    hit = __any(lane_hit);
  }
  __builtin_amdgcn_wave_barrier();
  return hit;
}

// exact test of one posed robot (pose p, rotation R, bounding-sphere centre c) by one wavefront
__device__ bool pose_exact(const EnvView& env, const RobotView& rob, const double* rtri, int32_t* stack, int32_t* cand,
                           double* stage, const double* p, const double* R, const double* c, int lane, ShareArea* share = nullptr) {
  // conservative query box around the posed bounding sphere
  double qlo[3], qhi[3];
  double rr = rob.radius * (1 + 1e-9) + 1e-9 * (fabs(c[0]) + fabs(c[1]) + fabs(c[2]) + 1);
  for (int k = 0; k < 3; ++k) { qlo[k] = c[k] - rr; qhi[k] = c[k] + rr; }

  WaveStack st{stack, 0, stack + STACK_CAP};
  bool overflow;
  int nc = collect_candidates(env, qlo, qhi, lane, st, cand, CAND_CAP, &overflow);
  bool hit = false;
  if (overflow) {
    // list overflowed: fall back to every env triangle (correct, slow, practically unreachable)
    for (int r0 = 0; r0 < rob.n_tri && !hit; r0 += 64) {
      const int r = r0 + lane;
      double Q[9];
      const bool have = r < rob.n_tri;
      if (have)
        for (int v = 0; v < 3; ++v) xform(R, p, rtri + 9 * r + 3 * v, Q + 3 * v);
      bool lane_hit = false;
      for (int t = 0; t < env.n_tri; ++t) {
        if (have && !lane_hit) {
          const double* b = env.tri_box + 6 * (size_t)t;
          if (tri_box_overlap(b, b + 3, Q)) lane_hit = sat17(env.tri + 9 * (size_t)t, Q);
        }
      }
      hit = __any(lane_hit);
    }
  } else {
    // candidates staged 64 at a time; every lane poses up to ceil(n_tri/64) robot triangles and walks the stage - or, with
    // many candidates, blocks of them shared with the workgroup's idle wavefronts (ShareSlot, kind 1)
    const bool shared = share && nc >= SHARE_POSE_MIN;
    ShareSlot* const hs = shared ? &share->slot[threadIdx.x >> 6] : nullptr;
    unsigned seq = 0;
    if (shared && lane == 0) {
      seq = (unsigned)(hs->sk >> 32) + 1u;
      hs->nc = nc; hs->done_k = 0; hs->minhit = 0x7fffffff; hs->kind = 1;
      for (int k = 0; k < 6; ++k) hs->a[k] = p[k];
      for (int k = 0; k < 3; ++k) hs->b[k] = c[k];
      for (int k = 0; k < 9; ++k) hs->R[k] = R[k];
      __threadfence_block();
      __hip_atomic_store(&hs->sk, (unsigned long long)seq << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    const int blk = shared ? share_block(nc) : STAGE_N;
    for (int knext = 0; !hit; knext += STAGE_N) {
      int k0 = knext;
      if (shared) {
        int seen = 0x7fffffff;
        if (lane == 0) {
          k0 = (int)(unsigned)(atomicAdd(&hs->sk, (unsigned long long)blk) & 0xffffffffULL);
          seen = lds_load32(&hs->minhit);
        }
        k0 = __builtin_amdgcn_readfirstlane(k0);
        seen = __builtin_amdgcn_readfirstlane(seen);
        if (k0 >= nc) break;
        const int kc = nc - k0 < blk ? nc - k0 : blk;
        const bool h = seen == 0 ? true : pose_block(env, rob, rtri, stage, cand, k0, kc, p, R, c, rr, lane);
        if (lane == 0) {
          if (h) atomicMin(&hs->minhit, 0);
          atomicAdd(&hs->done_k, kc);
        }
        continue;   // (the owner keeps taking blocks until none is left: the word says when)
      }
      if (k0 >= nc) break;
      const int kc = nc - k0 < blk ? nc - k0 : blk;
      hit = pose_block(env, rob, rtri, stage, cand, k0, kc, p, R, c, rr, lane);
    }
    if (shared) {
      int out = 0;
      if (lane == 0) {
        while (lds_load32(&hs->done_k) < nc) __builtin_amdgcn_s_sleep(2);
        out = lds_load32(&hs->minhit);
        __hip_atomic_store(&hs->sk, ((unsigned long long)seq << 32) | (unsigned long long)SHARE_IDLE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      hit = __builtin_amdgcn_readfirstlane(out) == 0;
    }
  }
  return hit;
}

__global__ __launch_bounds__(64 * POSE_WAVES) void k_collide_poses(EnvView env, RobotView rob,
                                                                   const double* __restrict__ pos6, int n,
                                                                   const int32_t* __restrict__ live_flags,
                                                                   uint8_t* __restrict__ hit_out, int explicit_rt) {
  extern __shared__ double lds_d[];
  // layout: robot triangles (n_tri*9 doubles) | per-wave candidate stages | per-wave stacks | per-wave candidate lists
  double* rtri = lds_d;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double* stage = rtri + (size_t)rob.n_tri * 9 + (size_t)wave * STAGE_DOUBLES;
  int32_t* ibase = reinterpret_cast<int32_t*>(rtri + (size_t)rob.n_tri * 9 + (size_t)POSE_WAVES * STAGE_DOUBLES);
  int32_t* stack = ibase + wave * (STACK_CAP + TG_HASH);   // (+ the wave's triangle-grid hash set behind its stack)
  int32_t* cand = ibase + POSE_WAVES * (STACK_CAP + TG_HASH) + wave * CAND_CAP;
  // which poses of this workgroup need the exact test at all (most do not: clearance bits)
  const int pose = blockIdx.x * POSE_WAVES + wave;
  bool need = false;
  double p[6], R[9], c[3] = {0, 0, 0};
  if (pose < n) {
    const bool run = !live_flags || (live_flags[pose] & 3) == 1;   // else not owned / out of limits / host path
    if (run && env.n_tri != 0) {                                     // (HasMap == false: src/environment.h:307-309)
      if (explicit_rt) {   // pos6 holds n x 12 doubles: the 3x3 rotation (row-major) and the translation
        const double* rt = pos6 + 12 * (size_t)pose;
        for (int k = 0; k < 9; ++k) R[k] = rt[k];
        for (int k = 0; k < 3; ++k) { p[k] = rt[9 + k]; p[3 + k] = 0; }
        need = !surely_clear(env, p);
        if (need) xform(R, p, rob.center, c);
      } else {
        for (int k = 0; k < 6; ++k) p[k] = pos6[6 * (size_t)pose + k];
        need = !surely_clear(env, p);
        if (need) pose_frame(rob, pos6, pose, p, R, c);
      }
    }
    if (!need && lane == 0) hit_out[pose] = 0;
  }
  if (!__syncthreads_or(need ? 1 : 0)) return;
  for (int i = threadIdx.x; i < rob.n_tri * 9; i += blockDim.x) rtri[i] = rob.tri[i];
  __syncthreads();
  if (!need) return;
  const bool hit = pose_exact(env, rob, rtri, stack, cand, stage, p, R, c, lane);
  if (lane == 0) hit_out[pose] = hit ? 1 : 0;
}

// ------------------------------------------------------------------ segment kernel
// boxes of the robot triangles as loaded (no rotation): lo xyz, hi xyz per triangle, next to the triangles in LDS
__device__ __forceinline__ void fill_robot_boxes(const double* rtri, double* rbox, int n_tri, int tid, int nthreads) {
  for (int r = tid; r < n_tri; r += nthreads)
    for (int ax = 0; ax < 3; ++ax) {
      rbox[6 * r + ax] = min3(rtri[9 * r + ax], rtri[9 * r + 3 + ax], rtri[9 * r + 6 + ax]);
      rbox[6 * r + 3 + ax] = max3(rtri[9 * r + ax], rtri[9 * r + 3 + ax], rtri[9 * r + 6 + ax]);
    }
}

#define QUEUE_CAP 128

// One block of an item's candidate triangles (kc <= STAGE_N of them, cand[k0 .. k0 + kc)) against the chunk's samples: the
// narrow phase of segment_chunk.  Lane = sample (SegLane); returns the smallest colliding sample index found (minhit in:
// what is known so far - later samples are skipped).  The wave's own stage / queue buffers; the candidate list may be
// another wavefront's of the same workgroup (share_items).
struct SegLane {
  double a[3], dir[3], parts;   // (uniform) the edge: start point, b - a, dist6 / 0.1
  int s0;                       // (uniform) first sample index of the chunk
  int idx;                      // this lane's sample
  bool need;
  double P[3], C[3], rr;
};
__device__ int narrow_block(const EnvView& env, const RobotView& rob, const double* rtri, const double* rbox, int32_t* queue,
                            double* stage, const int32_t* cand, int k0, int kc, const SegLane& L, int minhit, int lane DBG_ARG) {
  const double* a = L.a; const double* dir = L.dir;
  const double parts = L.parts;
  const int s0 = L.s0;
  const int idx = L.idx;
  const bool need = L.need;
  const double* P = L.P; const double* C = L.C;
  const double rr = L.rr;
  int qn = 0;
  auto flush = [&](int count) {
    DBG_ADD(3, 1);
    [[maybe_unused]] const unsigned long long tf_ = DBG_T();
    int v = 0x7fffffff;
    if (lane < count) {
      const int e = queue[lane];
      const int sl = e & 63, r = (e >> 6) & 1023, k = e >> 16;   // (k: index in the current stage)
      const int sidx = s0 + sl;
      if (sidx < minhit) {
        double S[3], Q[9];
        edge_sample_pos(a, dir, parts, sidx, S);
        // identity rotation: ((1*v0 + 0*v1) + 0*v2) + T == v0 + T
        for (int vv = 0; vv < 3; ++vv)
          for (int ax = 0; ax < 3; ++ax) Q[3 * vv + ax] = rtri[9 * r + 3 * vv + ax] + S[ax];
        if (sat17(stage + k * STAGE_TRI + 11, Q)) v = sidx;
      }
    }
    for (int off = 32; off > 0; off >>= 1) {
      int o = __shfl_xor(v, off);
      v = o < v ? o : v;
    }
    if (v < minhit) minhit = v;
    // keep the entries beyond `count` (at most 63 of them)
    int keep = 0;
    if (lane + count < qn) keep = queue[lane + count];
    if (lane + count < qn) queue[lane] = keep;
    qn -= count;
    DBG_ADD(13, DBG_T() - tf_);
  };
    [[maybe_unused]] const unsigned long long ts_ = DBG_T();
    stage_candidates(env, cand, k0, kc, lane, stage);
    DBG_ADD(10, DBG_T() - ts_);
    for (int k = 0; k < kc; ++k) {
      [[maybe_unused]] const unsigned long long tc_ = DBG_T();
      const double* bx = stage + k * STAGE_TRI;
      // extent of the (un-rotated: edge samples carry no rotation) robot along the triangle's normal, over all its
      // triangle vertices: a sample whose whole robot stays on one side of the triangle's plane touches nothing of it
      // (the exact test would separate them on that very axis; the slack, 1e-9 relative, dwarfs the rounding)
      // bx[20..21]: extent of the (un-rotated: edge samples carry no rotation) robot along the triangle's normal.  A
      // sample whose whole robot stays on one side of the triangle's plane touches nothing of it (the exact test would
      // separate them on that very axis; the slack, 1e-9 relative, dwarfs the rounding).
      const double* pl = bx + 6;
      const double elo = bx[20], ehi = bx[21];
      bool touch = false;
      if (need && idx < minhit) {
        touch = true;
        for (int ax = 0; ax < 3; ++ax) {
          // exact bounds of v + P over the robot vertices (monotone rounding of one add)
          double rlo = rob.lo[ax] + P[ax], rhi = rob.hi[ax] + P[ax];
          if (bx[ax] > rhi || rlo > bx[3 + ax]) touch = false;
        }
        if (touch && plane_clear(pl, C, rr)) touch = false;
        if (touch) {
          const double base = (pl[0] * P[0] + pl[1] * P[1]) + pl[2] * P[2] - pl[3];
          const double slack = 1e-9 * (pl[4] * (fabs(P[0]) + fabs(P[1]) + fabs(P[2]) + 1.0) + fabs(pl[3]) + (elo > -1e299 ? fabs(elo) + fabs(ehi) : 0.0));
          if (base + elo > slack || base + ehi < -slack) touch = false;
        }
        if (touch && tri_far(bx + 11, C, rr)) touch = false;
      }
      unsigned long long todo = __ballot(touch);
      DBG_ADD(11, DBG_T() - tc_);
      DBG_ADD(14, __popcll(todo));
      while (todo) {
        const int sl = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        if (s0 + sl >= minhit) break;   // later samples cannot improve the first hit
        double S[3];
        S[0] = __shfl(P[0], sl); S[1] = __shfl(P[1], sl); S[2] = __shfl(P[2], sl);
        for (int r0 = 0; r0 < rob.n_tri; r0 += 64) {
          const int r = r0 + lane;
          bool ok = false;
          if (r < rob.n_tri) {
            // box of the (un-rotated) robot triangle, shifted to the sample: min / max over its vertices of fl(v + S) =
            // fl(min / max v + S) - rounding is monotone - so this is tri_box_overlap on the shifted triangle, bit for bit
            const double* rb = rbox + 6 * r;
            ok = true;
            for (int ax = 0; ax < 3; ++ax) {
              const double qmin = rb[ax] + S[ax], qmax = rb[3 + ax] + S[ax];
              if (bx[ax] > qmax || qmin > bx[3 + ax]) ok = false;
            }
          }
          const unsigned long long mm = __ballot(ok);
          if (mm) {
            const int before = __popcll(mm & ((1ULL << lane) - 1ULL));
            if (ok) queue[qn + before] = sl | (r << 6) | (k << 16);
            qn += __popcll(mm);
            if (qn >= 64) flush(64);
          }
        }
      }
    }
  while (qn > 0) flush(qn < 64 ? qn : 64);   // (the stage is the caller's to overwrite after this)
  return minhit;
}

// a wavefront whose own items are done: blocks of its siblings' items until every wavefront of the workgroup is done
__device__ void share_help(const EnvView& env, const RobotView& rob, const double* rtri, const double* rbox, int32_t* queue,
                           double* stage, const int32_t* cand_base, ShareArea& sh, int lane DBG_ARG) {
  const int wave = threadIdx.x >> 6;
  if (lane == 0) atomicAdd(&sh.done_waves, 1);
  while (true) {
    bool worked = false;
    for (int w2 = 0; w2 < SEG_WAVES; ++w2) {
      if (w2 == wave) continue;
      ShareSlot& hs = sh.slot[w2];
      int k0 = SHARE_IDLE, nc = 0, blk = 1;
      if (lane == 0) {
        const unsigned long long cur = lds_load64(&hs.sk);
        const int nc0 = lds_load32(&hs.nc);
        if ((int)(unsigned)(cur & 0xffffffffULL) < nc0) {
          // (the grab names its item: the parameters read behind it are that item's - the owner cannot leave the item
          // before the block is accounted for; whoever reserves [k0, k0 + blk) works on exactly that range)
          blk = share_block(nc0);
          k0 = (int)(unsigned)(atomicAdd(&hs.sk, (unsigned long long)blk) & 0xffffffffULL);
          __threadfence_block();
          nc = lds_load32(&hs.nc);
        }
      }
      __threadfence_block();
      k0 = __builtin_amdgcn_readfirstlane(k0);
      nc = __builtin_amdgcn_readfirstlane(nc);
      blk = __builtin_amdgcn_readfirstlane(blk);
      if (k0 >= nc) continue;
      worked = true;
      const int kc = nc - k0 < blk ? nc - k0 : blk;
      int mh = 0;
      if (lane == 0) mh = lds_load32(&hs.minhit);
      mh = __builtin_amdgcn_readfirstlane(mh);
      if (hs.kind == 1) {   // a pose: a = the pose, b = the sphere centre, R
        if (mh != 0) {
          double pp[6], RR[9], cc[3];
          for (int k = 0; k < 6; ++k) pp[k] = hs.a[k];
          for (int k = 0; k < 3; ++k) cc[k] = hs.b[k];
          for (int k = 0; k < 9; ++k) RR[k] = hs.R[k];
          const double rr = rob.radius * (1 + 1e-9) + 1e-9 * (fabs(cc[0]) + fabs(cc[1]) + fabs(cc[2]) + 1);
          if (pose_block(env, rob, rtri, stage, cand_base + w2 * CAND_CAP, k0, kc, pp, RR, cc, rr, lane)) mh = 0;
        }
      } else {
        double a[6], b[6];
        for (int k = 0; k < 6; ++k) { a[k] = hs.a[k]; b[k] = hs.b[k]; }
        const unsigned long long mask = hs.mask;
        const int chunk = hs.chunk;
        SegLane L;
        L.parts = edge_parts(a, b);
        L.s0 = 1 + 64 * chunk;
        for (int k = 0; k < 3; ++k) { L.a[k] = a[k]; L.dir[k] = b[k] - a[k]; L.P[k] = 0.0; }
        L.idx = L.s0 + lane;
        L.need = ((mask >> lane) & 1ULL) != 0;
        if (L.idx <= edge_samples(L.parts)) edge_sample_pos(L.a, L.dir, L.parts, L.idx, L.P);
        for (int k = 0; k < 3; ++k) L.C[k] = L.P[k] + rob.center[k];
        L.rr = rob.radius * (1 + 1e-9) + 1e-9 * (fabs(L.C[0]) + fabs(L.C[1]) + fabs(L.C[2]) + 1);
        mh = narrow_block(env, rob, rtri, rbox, queue, stage, cand_base + w2 * CAND_CAP, k0, kc, L, mh, lane DBG_PASS);
      }
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) {
        if (mh != 0x7fffffff) atomicMin(&hs.minhit, mh);
        __threadfence_block();
        atomicAdd(&hs.done_k, kc);
      }
    }
    if (!worked) {
      int dw = 0;
      if (lane == 0) dw = lds_load32(&sh.done_waves);
      if (__builtin_amdgcn_readfirstlane(dw) >= SEG_WAVES) break;
      // (a waiting wavefront shares its SIMD with a working one of another workgroup: it looks again every ~0.5 us, not
      // every few hundred cycles)
      __builtin_amdgcn_s_sleep(20);
    }
  }
}

// One wavefront per (edge, chunk of 64 consecutive samples): lane = sample.  The chunk's own swept
// box gives a tight broad phase; the smallest colliding sample index of an edge is reduced with
// atomicMin, so the answer does not depend on which chunk finishes first.
__device__ void segment_chunk(const EnvView& env, const RobotView& rob, const double* rtri, const double* rbox, int32_t* stack,
                              int32_t* cand, int32_t* queue, double* stage, const double* a, const double* b, int seg,
                              int chunk, bool have_mask, unsigned long long mask, int32_t* __restrict__ first_hit,
                              int32_t* __restrict__ overflow_flag, int lane DBG_ARG, ShareArea* share = nullptr) {
  [[maybe_unused]] const unsigned long long t0_ = DBG_T();
  DBG_ADD(0, 1);
  const double parts = edge_parts(a, b);
  const int ns = edge_samples(parts);
  const int s0 = 1 + 64 * chunk;
  if (s0 > ns) return;
  const double dir[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
  const int idx = s0 + lane;
  const bool live = idx <= ns;
  double P[3] = {0, 0, 0};
  if (live) edge_sample_pos(a, dir, parts, idx, P);
  const double C[3] = {P[0] + rob.center[0], P[1] + rob.center[1], P[2] + rob.center[2]};
  // samples whose clearance bit is set need nothing; the others bound the chunk's broad-phase box
  // (the cull kernel already looked the bits up when it hands a mask over)
  const bool need = have_mask ? ((mask >> lane) & 1ULL) != 0 : (live && !surely_clear_edge(env, P));
  const unsigned long long nm = __ballot(need);
  [[maybe_unused]] const unsigned long long t1_ = DBG_T();
  DBG_ADD(4, t1_ - t0_);
  if (!nm) return;
  DBG_ADD(1, 1);
  const int l0 = __ffsll((long long)nm) - 1, l1 = 63 - __clzll((long long)nm);
  // sample positions are monotone in the index per coordinate (monotone rounding), so the first and the
  // last sample that need a test bound all of them; + the un-rotated robot box bounds every posed vertex
  double F[3], L[3], qlo[3], qhi[3];
  edge_sample_pos(a, dir, parts, s0 + l0, F);
  edge_sample_pos(a, dir, parts, s0 + l1, L);
  for (int k = 0; k < 3; ++k) {
    double lo = F[k] < L[k] ? F[k] : L[k], hi = F[k] > L[k] ? F[k] : L[k];
    double slack = 1e-9 * (fabs(lo) + fabs(hi) + 1);
    qlo[k] = lo + rob.lo[k] - slack;
    qhi[k] = hi + rob.hi[k] + slack;
  }
  WaveStack st{stack, 0, stack + STACK_CAP};
  bool overflow;
  int nc = collect_candidates(env, qlo, qhi, lane, st, cand, CAND_CAP, &overflow);
  [[maybe_unused]] const unsigned long long t2_ = DBG_T();
  DBG_ADD(5, t2_ - t1_);
#ifdef SFFK_CI_TRACE
  if (lane == 0) { const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; if (gw < 4096) { g_ci_tmp[2 * gw] = wall_clock64(); g_ci_tmp[2 * gw + 1] = (unsigned long long)nc; } }
#endif
  DBG_ADD(7, nc);
  if (overflow) {
    if (lane == 0) {   // host re-runs this edge through the pose kernel; first_hit 0 (no sample has index 0) also says so
      atomicOr(overflow_flag + seg, 1);
      atomicMin(first_hit + seg, 0);
    }
    return;
  }
  if (nc == 0) return;
  DBG_ADD(2, 1);
  SegLane SL;
  for (int k = 0; k < 3; ++k) { SL.a[k] = a[k]; SL.dir[k] = dir[k]; SL.P[k] = P[k]; SL.C[k] = C[k]; }
  SL.parts = parts; SL.s0 = s0; SL.idx = idx; SL.need = need;
  SL.rr = rob.radius * (1 + 1e-9) + 1e-9 * (fabs(C[0]) + fabs(C[1]) + fabs(C[2]) + 1);
  int minhit = 0x7fffffff;
  // many candidates: the item is published for the workgroup's idle wavefronts (ShareSlot) and the candidates are taken in
  // blocks of SHARE_BLK off the shared word - by this wavefront too; otherwise STAGE_N at a time, all here (one call site
  // of narrow_block for both: the exact kernel sits at its register limit)
  const bool shared = share && nc >= 4 && nc * __popcll(nm) >= SHARE_WORK;
  ShareSlot* const hs = shared ? &share->slot[threadIdx.x >> 6] : nullptr;
  unsigned seq = 0;
  if (shared && lane == 0) {
    seq = (unsigned)(hs->sk >> 32) + 1u;
    hs->nc = nc; hs->done_k = 0; hs->minhit = 0x7fffffff; hs->seg = seg; hs->chunk = chunk; hs->mask = nm; hs->kind = 0;
    for (int k = 0; k < 6; ++k) { hs->a[k] = a[k]; hs->b[k] = b[k]; }
    __threadfence_block();
    __hip_atomic_store(&hs->sk, (unsigned long long)seq << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  const int blk = shared ? share_block(nc) : STAGE_N;
  for (int knext = 0; ; knext += STAGE_N) {
    int k0 = knext;
    if (shared) {
      if (lane == 0) {
        k0 = (int)(unsigned)(atomicAdd(&hs->sk, (unsigned long long)blk) & 0xffffffffULL);
        const int mh = lds_load32(&hs->minhit);
        if (mh < minhit) minhit = mh;
      }
      k0 = __builtin_amdgcn_readfirstlane(k0);
      minhit = __builtin_amdgcn_readfirstlane(minhit);
    }
    if (k0 >= nc) break;
    const int kc = nc - k0 < blk ? nc - k0 : blk;
    minhit = narrow_block(env, rob, rtri, rbox, queue, stage, cand, k0, kc, SL, minhit, lane DBG_PASS);
    __builtin_amdgcn_wave_barrier();
    if (shared && lane == 0) {
      if (minhit != 0x7fffffff) atomicMin(&hs->minhit, minhit);
      atomicAdd(&hs->done_k, kc);
    }
  }
  if (shared) {
    if (lane == 0) {
      while (lds_load32(&hs->done_k) < nc) __builtin_amdgcn_s_sleep(2);   // the siblings' blocks
      minhit = lds_load32(&hs->minhit);
      // (idle again, same sequence: a late grab finds nothing to take)
      __hip_atomic_store(&hs->sk, ((unsigned long long)seq << 32) | (unsigned long long)SHARE_IDLE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    minhit = __builtin_amdgcn_readfirstlane(minhit);
  }
  DBG_ADD(6, DBG_T() - t2_);
  if (minhit != 0x7fffffff && lane == 0) atomicMin(first_hit + seg, minhit);
}

struct WorkItem {   // 64 bytes: one (edge, chunk of 64 samples) unit of collision work
  int32_t slot, chunk, ns, pad;
  double a[3];      // start point of the edge (xyz)
  double step[3];   // (b - a) / parts: distance between two consecutive samples
};

// Edge tasks are written on the device (k_classify for the forest rounds, k_seg_prepare for host batches)
// into a sparse slot table (seg_ns > 0 = live edge).  k_seg_compact turns the table into a dense list of
// 64-byte (slot, chunk, ...) work items - one block-level scan and ONE atomic per block, because returning atomics on a
// single word saturate near 90/us chip-wide and a per-item dequeue would cost more than the work itself.
// ctrl[2] = items reserved, ctrl[3] = 1 when the list ran over (the edge kernel then scans the table).
__global__ __launch_bounds__(256) void k_seg_compact(const int32_t* __restrict__ seg_ns, int n_slots,
                                                     const double* __restrict__ a6, const double* __restrict__ b6,
                                                     int32_t* __restrict__ ctrl, WorkItem* __restrict__ list, int cap,
                                                     GridView tg, const float* __restrict__ tx,
                                                     const float* __restrict__ ty, const float* __restrict__ tz,
                                                     int n_temps, const int32_t* __restrict__ dev_n, int stride) {
  if (dev_n) {
    if (dev_n[1]) return;
    n_temps = dev_n[0];
    n_slots = dev_n[0] * stride;
  }
  __shared__ int wsum[4];
  __shared__ int base_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (tg.cnt) {   // housekeeping: the round's own grid has been read (k_grid_query), empty the cells it used
    for (int t = blockIdx.x * 256 + threadIdx.x; t < n_temps; t += gridDim.x * 256) {
      const float x = tx[t];
      if (x == x) {
        const size_t cell = grid_cell_of(tg, x, ty[t], tz[t]);
        tg.cnt[cell] = 0;
        if (tg.occ) tg.occ[cell >> 5] = 0u;   // (every set bit of the word belongs to a sample of this round)
      }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) tg.ovf_cnt[0] = 0;
  }
  const int s0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  int c[4], tot = 0;
  for (int j = 0; j < 4; ++j) {
    const int ns = (s0 + j < n_slots) ? seg_ns[s0 + j] : 0;
    c[j] = ns > 0 ? (ns + 63) >> 6 : 0;
    tot += c[j];
  }
  int inc = tot;  // inclusive scan inside the wave
  for (int off = 1; off < 64; off <<= 1) {
    int o = __shfl_up(inc, off);
    if (lane >= off) inc += o;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int all = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    base_s = all > 0 ? atomicAdd(ctrl + 2, all) : 0;
  }
  __syncthreads();
  int at = base_s + inc - tot;
  for (int w = 0; w < wave; ++w) at += wsum[w];
  // an item (64 bytes) = slot, chunk, samples of the edge | its start point | the step between two samples:
  // everything the cull kernel needs to place the chunk's samples (to a tolerance, see there)
  for (int j = 0; j < 4; ++j) {
    if (!c[j]) continue;
    const double* ea = a6 + 6 * (size_t)(s0 + j);
    const double* eb = b6 + 6 * (size_t)(s0 + j);
    const double inv = 1.0 / edge_parts(ea, eb);
    WorkItem it;
    it.slot = s0 + j;
    it.ns = seg_ns[s0 + j];
    it.pad = 0;
    for (int q = 0; q < 3; ++q) { it.a[q] = ea[q]; it.step[q] = (eb[q] - ea[q]) * inv; }
    for (int k = 0; k < c[j]; ++k, ++at) {
      if (at < cap) { it.chunk = k; list[at] = it; }
      else ctrl[3] = 1;
    }
  }
}

// The exact kernel's wave w looks at items w + W * lane of each window of 64 W items (W = its wave count); their
// masks are stored so that this is one contiguous 512-byte read.
__device__ __forceinline__ size_t mask_slot(int e, int W) {
  const int win = e / (64 * W), r = e - win * 64 * W;
  return (size_t)win * 64 * W + (size_t)(r % W) * 64 + (size_t)(r / W);
}

// Lean, high-occupancy pass in front of the exact kernel: looks up the clearance bits of every pose (one thread
// each, blocks [0, pose_blocks)) and of every sample of every (edge, chunk) work item (one wavefront per item, the
// other blocks).  A pose that needs the exact test gets pose_hit = 2, the others are answered here (0); an item
// gets the 64-bit mask of its samples that need the exact test (0 = drop the item; an edge whose items are all
// dropped keeps its preset "free").  No atomics, no barriers: the exact kernel scans both arrays.
// Sample positions are only needed to a tolerance far inside the slack of the clearance bits here, so
// a + idx * dir / parts becomes a + idx * step.
__global__ __launch_bounds__(256) void k_cull(EnvView env, const double* __restrict__ pos6, int n_pose, int pose_blocks,
                                              const int32_t* __restrict__ live_flags, uint8_t* __restrict__ pose_hit,
                                              const WorkItem* __restrict__ list, unsigned long long* __restrict__ masks,
                                              int exact_waves, const int32_t* __restrict__ ctrl,
                                              const int32_t* __restrict__ dev_n) {
  if (dev_n) {
    if (dev_n[1]) return;
    n_pose = dev_n[0];
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if ((int)blockIdx.x < pose_blocks) {
    const int pose = blockIdx.x * 256 + threadIdx.x;
    if (pose >= n_pose) return;
    bool need = false;
    const bool run = !live_flags || (live_flags[pose] & 3) == 1;   // else not owned / out of limits / host path
    if (run && env.n_tri != 0) {                                     // (HasMap == false: src/environment.h:307-309)
      // the bits are built for a sphere around the model origin that holds the robot in every rotation
      const double o[3] = {pos6[6 * (size_t)pose], pos6[6 * (size_t)pose + 1], pos6[6 * (size_t)pose + 2]};
      need = !surely_clear(env, o);
    }
    pose_hit[pose] = need ? 2 : 0;
    return;
  }
  if (ctrl[3] || env.n_tri == 0) return;     // work list ran over: the exact kernel scans the slot table itself
  const int M = ctrl[2];
  const int cb = blockIdx.x - pose_blocks, nb = gridDim.x - pose_blocks;
  const int W = nb * 4;
  // four items per step: their dependent loads (item -> clearance word) are issued side by side
  for (int e = cb + nb * wave; e < M; e += 4 * W) {
    WorkItem it[4];
    bool val[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      val[u] = e + u * W < M;
      it[u] = list[val[u] ? e + u * W : e];
    }
    bool need[4];
    const uint32_t* wp[4];
    int sh[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = 1 + 64 * it[u].chunk + lane;
      const double t = (double)idx;
      const double C[3] = {it[u].a[0] + t * it[u].step[0], it[u].a[1] + t * it[u].step[1],
                           it[u].a[2] + t * it[u].step[2]};   // the model origin at this sample
      need[u] = val[u] && idx <= it[u].ns;
      wp[u] = nullptr;
      sh[u] = 0;
      if (need[u]) {
        const double fx = (C[0] - env.clear_org[0]) * env.clear_inv, fy = (C[1] - env.clear_org[1]) * env.clear_inv,
                     fz = (C[2] - env.clear_org[2]) * env.clear_inv;
        if (env.clear_bits_edge && fx == fx && fy == fy && fz == fz) {
          if (fx < 0 || fy < 0 || fz < 0 || fx >= env.clear_n[0] || fy >= env.clear_n[1] || fz >= env.clear_n[2]) {
            need[u] = false;                        // beyond the inflated box of the environment
          } else {                                  // (the grid has fewer than 2^31 cells)
            const uint32_t ci = ((uint32_t)(int)fz * (uint32_t)env.clear_n[1] + (uint32_t)(int)fy) * (uint32_t)env.clear_n[0] + (uint32_t)(int)fx;
            wp[u] = env.clear_bits_edge + (ci >> 5);
            sh[u] = (int)(ci & 31u);
          }
        }
      }
    }
    uint32_t word[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) word[u] = wp[u] ? *wp[u] : 0u;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (wp[u] && ((word[u] >> sh[u]) & 1u)) need[u] = false;
      const unsigned long long nm = __ballot(need[u]);
      if (val[u] && lane == 0) masks[mask_slot(e + u * W, exact_waves)] = nm;
    }
  }
}

// Exact kernel: persistent wavefronts scan what the cull left over - wave w looks at entries w + W * lane of a
// 64 W window, so the survivors (a few percent, clustered along the list) spread evenly over the waves without
// a compaction step - first the poses marked 2, then the (edge, chunk) items with a non-zero mask.  A chunk whose
// edge already has a hit below its first sample is skipped (only the smallest index matters).
__global__ __launch_bounds__(64 * SEG_WAVES) void k_collide_segments_dyn(EnvView env, RobotView rob,
                                                                         const double* __restrict__ pos6, int n_pose,
                                                                         uint8_t* __restrict__ pose_hit,
                                                                         const double* __restrict__ a6,
                                                                         const double* __restrict__ b6,
                                                                         const int32_t* __restrict__ seg_ns, int n_slots,
                                                                         int32_t* __restrict__ ctrl,
                                                                         const WorkItem* __restrict__ list,
                                                                         const unsigned long long* __restrict__ masks,
                                                                         int32_t* __restrict__ first_hit,
                                                                         int32_t* __restrict__ overflow_flag,
                                                                         const int32_t* __restrict__ dev_n, int stride) {
  if (dev_n) {
    if (dev_n[1]) return;
    n_pose = pose_hit ? dev_n[0] : 0;     // (edges only: no pose array to look at)
    n_slots = dev_n[0] * stride;
  }
  extern __shared__ double lds_d[];
  double* rtri = lds_d;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double* stage = rtri + (size_t)rob.n_tri * 9 + (size_t)wave * STAGE_DOUBLES;
  int32_t* ibase = reinterpret_cast<int32_t*>(rtri + (size_t)rob.n_tri * 9 + (size_t)SEG_WAVES * STAGE_DOUBLES);
  for (int i = threadIdx.x; i < rob.n_tri * 9; i += blockDim.x) rtri[i] = rob.tri[i];
  __syncthreads();
  double* rbox = reinterpret_cast<double*>(ibase + SEG_WAVES * (STACK_CAP + TG_HASH + CAND_CAP + QUEUE_CAP));
  fill_robot_boxes(rtri, rbox, rob.n_tri, threadIdx.x, blockDim.x);
  __syncthreads();
  if (env.n_tri == 0) return;
  int32_t* stack = ibase + wave * (STACK_CAP + TG_HASH);   // (+ the wave's triangle-grid hash set behind its stack)
  int32_t* cand = ibase + SEG_WAVES * (STACK_CAP + TG_HASH) + wave * CAND_CAP;
  int32_t* queue = ibase + SEG_WAVES * (STACK_CAP + TG_HASH + CAND_CAP) + wave * QUEUE_CAP;
  DBG_DECL
  [[maybe_unused]] const unsigned long long tk_ = DBG_T();
  const int W = gridDim.x * SEG_WAVES;
  const int w0 = blockIdx.x + gridDim.x * wave;   // neighbouring entries go to different CUs
  // The survivors of a workgroup's four windows are pooled in LDS and handed out one at a time (LDS atomics),
  // so a wave whose window happened to hold several of them does not become the kernel's tail.
  __shared__ int pool_n, pool_next;
  __shared__ int pool_e[64 * SEG_WAVES];
  __shared__ unsigned long long pool_m[64 * SEG_WAVES];
  auto pool_put = [&](bool s, int e, unsigned long long m) {
    const unsigned long long bal = __ballot(s);
    int off = 0;
    if (lane == 0 && bal) off = atomicAdd(&pool_n, __popcll(bal));
    off = __shfl(off, 0);
    if (s) {
      const int at = off + __popcll(bal & ((1ULL << lane) - 1ULL));
      pool_e[at] = e;
      pool_m[at] = m;
    }
  };
  auto pool_take = [&]() -> int {
    int i = 0;
    if (lane == 0) i = atomicAdd(&pool_next, 1);
    return __shfl(i, 0);
  };
  for (int base = 0; base < n_pose; base += 64 * W) {
    if (threadIdx.x == 0) { pool_n = 0; pool_next = 0; }
    __syncthreads();
    const int mine = base + w0 + W * lane;
    pool_put(mine < n_pose && pose_hit[mine] == 2, mine, 0ULL);
    __syncthreads();
    for (int i = pool_take(); i < pool_n; i = pool_take()) {
      const int pose = pool_e[i];
      double p[6], R[9], c[3];
      pose_frame(rob, pos6, pose, p, R, c);
      const bool hit = pose_exact(env, rob, rtri, stack, cand, stage, p, R, c, lane);
      if (lane == 0) pose_hit[pose] = hit ? 1 : 0;
    }
    __syncthreads();
  }
  if (!ctrl[3]) {
    const int M = ctrl[2];
    for (int base = 0; base < M; base += 64 * W) {
      if (threadIdx.x == 0) { pool_n = 0; pool_next = 0; }
      __syncthreads();
      const int mine = base + w0 + W * lane;
      const unsigned long long m = mine < M ? masks[(size_t)base + (size_t)w0 * 64 + lane] : 0ULL;
      pool_put(m != 0ULL, mine, m);
      __syncthreads();
      for (int i = pool_take(); i < pool_n; i = pool_take()) {
        const int e = pool_e[i];
        const unsigned long long nm = pool_m[i];
        const int slot = list[e].slot, chunk = list[e].chunk;
        if (chunk > 0 && first_hit[slot] <= 64 * chunk) continue;
        double a[6], b[6];
        for (int k = 0; k < 6; ++k) { a[k] = a6[6 * (size_t)slot + k]; b[k] = b6[6 * (size_t)slot + k]; }
        segment_chunk(env, rob, rtri, rbox, stack, cand, queue, stage, a, b, slot, chunk, true, nm, first_hit, overflow_flag, lane DBG_PASS);
      }
      __syncthreads();
    }
    DBG_ADD(8, DBG_T() - tk_);
    DBG_ADD(9, 1);
    DBG_FLUSH();
    return;
  }
  // the list ran over: scan the slot table, 64 slots per dequeue
  while (true) {
    int first = 0;
    if (lane == 0) first = atomicAdd(ctrl + 1, 64);
    first = __shfl(first, 0);
    if (first >= n_slots) break;
    const int slot = first + lane;
    const int ns = slot < n_slots ? seg_ns[slot] : -1;
    unsigned long long live = __ballot(ns > 0);
    while (live) {
      const int b = __ffsll((long long)live) - 1;
      live &= live - 1;
      const int nsb = __shfl(ns, b);
      double a[6], bb[6];
      for (int k = 0; k < 6; ++k) { a[k] = a6[6 * (size_t)(first + b) + k]; bb[k] = b6[6 * (size_t)(first + b) + k]; }
      for (int chunk = 0; chunk * 64 < nsb; ++chunk)
        segment_chunk(env, rob, rtri, rbox, stack, cand, queue, stage, a, bb, first + b, chunk, false, 0ULL, first_hit, overflow_flag, lane DBG_PASS);
    }
  }
}

// (an id < 0 names row -1 - id of `extra`: points that are not in the store - the RRT session's new points of the wave)
__global__ __launch_bounds__(256) void k_seg_gather(const double* __restrict__ store_pos, const int32_t* __restrict__ ida,
                                                    const int32_t* __restrict__ idb, int n, double* __restrict__ a6,
                                                    double* __restrict__ b6, const double* __restrict__ extra) {
  const int t = blockIdx.x * 256 + threadIdx.x;   // 6 threads per edge end: one double each
  const int e = t / 12, k = t % 12;
  if (e >= n) return;
  const int id = k < 6 ? ida[e] : idb[e], q = k < 6 ? k : k - 6;
  const double v = id >= 0 ? store_pos[6 * (size_t)id + q] : extra[6 * (size_t)(-1 - id) + q];
  if (k < 6) a6[6 * (size_t)e + q] = v; else b6[6 * (size_t)e + q] = v;
}

// sample counts + result presets for a host-supplied batch of edges (C-ABI sffgpu_collide_segments)
__global__ __launch_bounds__(256) void k_seg_prepare(const double* __restrict__ a6, const double* __restrict__ b6, int n,
                                                     int32_t* __restrict__ seg_ns, int32_t* __restrict__ first_hit,
                                                     int32_t* __restrict__ ovf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  seg_ns[i] = edge_samples(edge_parts(a6 + 6 * (size_t)i, b6 + 6 * (size_t)i));
  first_hit[i] = 0x7fffffff;
  ovf[i] = 0;
}

// ------------------------------------------------------------------ neighbour classification
// One wavefront per sample of the round, lane = sweep hit.  Replays, for its sample, the part of the
// reference's neighbour loop (src/forest.h:262-300) that decides WHICH edges have to be checked:
// keeps the hits that satisfy one of the two distance conditions (:276, :283), ranks them in the
// order the reference visits them (tree id, then distance, then index), cuts the list after the
// first store neighbour of another tree (the loop returns there, :296-299) and writes the edge
// tasks + their (edge, chunk) work items.
__device__ __forceinline__ void emit_items(const ClassifyArgs& A, size_t slot) {
  // the edge's sample count doubles as its "live" mark; the persistent edge kernel scans these
  A.seg_ns[slot] = edge_samples(edge_parts(A.seg_a + 6 * slot, A.seg_b + 6 * slot));
}

__global__ __launch_bounds__(256) void k_classify(ClassifyArgs A) {
  if (A.dev_n) {
    if (A.dev_n[1]) return;
    A.n = A.dev_n[0];
  }
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= A.n) return;
  const int stride = 1 + A.nbcap;
  for (int k = lane; k < stride; k += 64) {
    A.seg_ns[(size_t)i * stride + k] = -1;
    A.first_hit[(size_t)i * stride + k] = 0x7fffffff;
    A.seg_ovf[(size_t)i * stride + k] = 0;
  }
  int flags = 0, nnb = 0;
  const bool mine_shard = A.world <= 1 || i % A.world == A.rank;
  if (A.in_lim[i] && mine_shard) {
    flags |= 1;
    const int cnt = A.cnt[i];
    if (cnt > A.cap) {
      flags |= 2;
    } else {
      const int ex = A.parent[i];
      const int mine = A.tree[ex];
      const bool force = A.force[i] != 0;
      const double pdist = A.pdist[i];
      const bool have = lane < cnt;
      const int id = have ? A.hit_idx[(size_t)i * A.cap + lane] : 0x7fffffff;
      const double d = have ? A.hit_dist[(size_t)i * A.cap + lane] : 0.0;
      const int t = have ? A.tree[id] : 0x7fffffff;
      const bool same = t == mine;
      bool q = false;
      if (have) q = same ? (!force && d < pdist - SFFG_TOL)          // src/forest.h:276
                         : (d < A.dist_tree - SFFG_TOL);             // src/forest.h:283
      int rank = 0;
      for (int j = 0; j < cnt; ++j) {   // (lanes >= cnt hold no hit)
        const int tj = __shfl(t, j), idj = __shfl(id, j), qj = __shfl((int)q, j);
        const double dj = __shfl(d, j);
        if (qj && (tj < t || (tj == t && (dj < d || (dj == d && idj < id))))) ++rank;
      }
      int cut = (q && !same && id < A.N0) ? rank : 0x7fffffff;
      for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_xor(cut, off);
        cut = o < cut ? o : cut;
      }
      const bool keep = q && rank <= cut;
      const int nkeep = __popcll(__ballot(keep));
      if (nkeep > A.nbcap) {
        flags |= 2;   // the host path redoes this sample with unbounded lists
      } else {
        nnb = nkeep;
        const double* exp = A.pos + 6 * (size_t)ex;
        const double* np = A.newpos + 6 * (size_t)i;
        if (keep) {
          A.rec_nb[(size_t)i * A.nbcap + rank] = id;
          A.rec_meta[(size_t)i * A.nbcap + rank] = (t << 1) | (same ? 1 : 0);
          const size_t slot = (size_t)i * stride + 1 + rank;
          const double* nbp = A.pos + 6 * (size_t)id;
          double* sa = A.seg_a + 6 * slot;
          double* sb = A.seg_b + 6 * slot;
          if (same) { for (int k = 0; k < 6; ++k) { sa[k] = nbp[k]; sb[k] = np[k]; } }                 // isPathFree(neighbour, newPoint) :276
          else if (id == A.goal_id) { for (int k = 0; k < 6; ++k) { sa[k] = np[k]; sb[k] = nbp[k]; } } // isPathFree(newPoint, goal) :287
          else { for (int k = 0; k < 6; ++k) { sa[k] = exp[k]; sb[k] = nbp[k]; } }                     // isPathFree(expanded, neighbour) :288
          emit_items(A, slot);
        }
        if (lane == 0) {   // slot 0: isPathFree(expanded, newPoint)  (src/forest.h:246)
          const size_t slot = (size_t)i * stride;
          double* sa = A.seg_a + 6 * slot;
          double* sb = A.seg_b + 6 * slot;
          for (int k = 0; k < 6; ++k) { sa[k] = exp[k]; sb[k] = np[k]; }
          emit_items(A, slot);
        }
      }
    }
  }
  if (lane == 0) {
    A.rec_flags[i] = flags;
    A.rec_nnb[i] = nnb;
  }
}

// ------------------------------------------------------------------ neighbour query + classification, fused
// One wavefront per sample.  Candidates: the items of the grid cells the query ball touches (27 for the planner's
// radius), in the node grid and - where its occupancy bit is set - in the round's own grid, flattened over the
// lanes (lane = candidate, not lane = cell, so that a cell with several items costs no extra step), fp32 superset
// filter, exact fp64 re-test of the survivors.  The hits are compacted with __ballot into the wave's LDS slice
// (no atomics, no hit list in HBM) and classified right away exactly like k_classify does.
#define QC_WAVES 4
// A sample with more than 64 hits (the round's samples crowd together: priority-frontier mode, forests that fill the
// angular dimensions) cannot rank them with one lane per hit.  Its grid scan is then run a second time with this
// collector: only the hits that QUALIFY as neighbours (src/forest.h:276,283) matter, and of those only the first 16 in
// the reference's (tree, distance, id) order - the neighbour record holds 15, the 16th says "there are more".  Lane j
// holds the j-th of them (rank insertion like the k-nearest kernels' TopK).
#define QC_TOP 16
struct QcTop {
  int t, id;
  double d;
  int have;          // (uniform)
  int mine;          // the sample's tree, ForceChildren of the expanded node, parentDistance, treeDistance
  bool force;
  double pdist, dist_tree;
};
__device__ __forceinline__ void qc_top_offer(QcTop& T, int lane, bool hit, double d, int id, int tree) {
  bool q = false;
  if (hit) q = tree == T.mine ? (!T.force && d < T.pdist - SFFG_TOL) : (d < T.dist_tree - SFFG_TOL);
  unsigned long long take = __ballot(q);
  while (take) {
    const int src = __ffsll((long long)take) - 1;
    take &= take - 1;
    const double nd = __shfl(d, src);
    const int ni = __shfl(id, src), nt = __shfl(tree, src);
    const bool before = lane < T.have && (T.t < nt || (T.t == nt && (T.d < nd || (T.d == nd && T.id < ni))));
    const int rank = __popcll(__ballot(before));
    if (rank >= QC_TOP) continue;
    const double pd = __shfl_up(T.d, 1);
    const int pi = __shfl_up(T.id, 1), pt = __shfl_up(T.t, 1);
    if (lane > rank) { T.d = pd; T.id = pi; T.t = pt; }
    else if (lane == rank) { T.d = nd; T.id = ni; T.t = nt; }
    if (T.have < QC_TOP) T.have += 1;
  }
}
// g + tg: the node grid's and the round's own grid's items of the same cells in ONE flattened pass (lane = candidate):
// both cell counts arrive with the same round of loads, so do both grids' items
__device__ __forceinline__ void qc_candidates(const GridView& g, const GridView& tg, int m, int mt, int cell, int lane,
                                              const SweepQuery& Q, const double* qp, int32_t* h_id, double* h_d,
                                              int32_t* h_tree, double* h_pos, int& nh, QcTop* top = nullptr) {
  // exclusive prefix of the per-lane item counts
  const int mm = m + mt;
  int inc = mm;
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(inc, off);
    if (lane >= off) inc += o;
  }
  const int total = __shfl(inc, 63);
  for (int base = 0; base < total; base += 64) {
    const int j = base + lane;
    bool hit = false;
    double d = 0;
    GridItem it;
    it.id = 0; it.tree = 0;
    // the lane whose cell holds candidate j: the first lane with inclusive prefix > j.  Every lane takes part in the
    // shuffles (lanes beyond the list search for the last candidate), the loop always runs 6 steps.
    const int jj = j < total ? j : total - 1;
    int lo = 0, hi = 63;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (__shfl(inc, mid) > jj) hi = mid; else lo = mid + 1;
    }
    const int src_cell = __shfl(cell, lo);
    const int src_m = __shfl(m, lo);
    const int slot = jj - (__shfl(inc, lo) - __shfl(mm, lo));
    if (j < total) {
      const GridItem* src = slot < src_m ? g.items + ((size_t)src_cell * g.bk + slot)
                                         : tg.items + ((size_t)src_cell * tg.bk + (slot - src_m));
      it = *src;
      if (it.id < Q.max_id && (Q.tree < 0 || it.tree == Q.tree)) {
        d = dist6(it.p, qp);          // the item carries the node's fp64 position: exact at once
        hit = d < Q.r;
      }
    }
    if (top) { qc_top_offer(*top, lane, hit, d, it.id, it.tree); continue; }
    const unsigned long long hm = __ballot(hit);
    if (hit) {
      const int at = nh + __popcll(hm & ((1ULL << lane) - 1ULL));
      if (at < 64) {
        h_id[at] = it.id; h_d[at] = d; h_tree[at] = it.tree;
        for (int k = 0; k < 6; ++k) h_pos[6 * at + k] = it.p[k];
      }
    }
    nh += __popcll(hm);
  }
}
__device__ __forceinline__ void qc_overflow(const GridView& g, int no, int lane, const SweepQuery& Q, const double* qp,
                                            int32_t* h_id, double* h_d, int32_t* h_tree, double* h_pos, int& nh,
                                            QcTop* top = nullptr) {
  if (no > g.ovf_cap) no = g.ovf_cap;
  for (int base = 0; base < no; base += 64) {
    const int j = base + lane;
    bool hit = false;
    double d = 0;
    GridItem it;
    it.id = 0; it.tree = 0;
    if (j < no) {
      it = g.ovf[j];
      if (it.id < Q.max_id && (Q.tree < 0 || it.tree == Q.tree)) {
        d = dist6(it.p, qp);
        hit = d < Q.r;
      }
    }
    if (top) { qc_top_offer(*top, lane, hit, d, it.id, it.tree); continue; }
    const unsigned long long hm = __ballot(hit);
    if (hit) {
      const int at = nh + __popcll(hm & ((1ULL << lane) - 1ULL));
      if (at < 64) {
        h_id[at] = it.id; h_d[at] = d; h_tree[at] = it.tree;
        for (int k = 0; k < 6; ++k) h_pos[6 * at + k] = it.p[k];
      }
    }
    nh += __popcll(hm);
  }
}

#define QC_SURV 64  // survivors of a sample gathered in LDS before they are appended (one atomic per kind)
#ifndef QC_HEAVY
#define QC_HEAVY 40 // masked samples from which a chunk counts as a heavy item of the exact kernel (poses always do)
#endif
#define QC_TAB 128  // (task, chunk) pairs of a sample unfolded at a time
#ifndef QC_OCC
#define QC_OCC 7   // wavefronts per SIMD the register allocation aims at (59 VGPRs).  Measured 5 ... 8: 42 / 40.3 / 39.3 / 40.9 us -
                    // beyond five resident waves the kernel is bound by the memory system's rate of scattered 64-byte
                    // accesses (~1.2 TB/s of sector traffic), not by latency
#endif
__global__ __launch_bounds__(64 * QC_WAVES) __attribute__((amdgpu_waves_per_eu(QC_OCC))) void k_query_classify(GridView g, GridView tg, NodeStoreView st,
                                                                  const SweepQuery* __restrict__ queries, ClassifyArgs A,
                                                                  EnvView env, int fused_cull) {
  __shared__ SurvivorItem s_surv[QC_WAVES][QC_SURV];   // fused cull: the sample's items for the exact kernel
  __shared__ int32_t s_tab[QC_WAVES][QC_TAB];           // fused cull: (task, chunk) of a window of pairs
  __shared__ int32_t s_id[QC_WAVES][64];
  __shared__ int32_t s_tree[QC_WAVES][64];
  __shared__ double s_d[QC_WAVES][64];
  __shared__ double s_pos[QC_WAVES][64 * 6];
  if (A.dev_n) {
    if (A.dev_n[1]) return;
    A.n = A.dev_n[0];
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * QC_WAVES + wave;
  if (i >= A.n) return;
  // (every 16th workgroup reports: ten thousand atomics on one word would cost more than the kernel itself)
  const bool clocked = A.qclk_sh && threadIdx.x == 0;
  if (clocked) atomicMin(A.qclk_sh + (blockIdx.x & 63) * 16 + 1, wall_clock64());
  [[maybe_unused]] const bool qdbg_on = (blockIdx.x & 15) == 0 && lane == 0;
  [[maybe_unused]] const unsigned long long qt0 = DBG_T();
  [[maybe_unused]] unsigned long long qt1 = qt0, qt_fl = 0;
  const int stride = 1 + A.nbcap;
  // ---- everything the sample needs, loaded before the first store (a wave runs one long chain of dependent
  // memory steps: independent loads are issued together, up front)
  // (all of it is the same in every lane: pinned to scalar registers - the kernel's occupancy is set by its vector
  // registers, and a wave's sample, expanded node and query would take forty of them)
  auto uni_i = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
  auto uni_f = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
  auto uni_d = [](double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
  };
  const bool inl = uni_i(A.in_lim[i]) != 0;
  const int ex = uni_i(A.parent[i]);
  const bool force = uni_i(A.force[i]) != 0;
  const double pdist = uni_d(A.pdist[i]);
  SweepQuery Q = queries[i];
  Q.x = uni_f(Q.x); Q.y = uni_f(Q.y); Q.z = uni_f(Q.z); Q.yaw = uni_f(Q.yaw); Q.pitch = uni_f(Q.pitch); Q.roll = uni_f(Q.roll);
  Q.r2f = uni_f(Q.r2f); Q.tree = uni_i(Q.tree); Q.r = uni_d(Q.r); Q.max_id = uni_i(Q.max_id); Q.active = uni_i(Q.active);
  double qp[6], exp[6];
  for (int k = 0; k < 6; ++k) qp[k] = uni_d(A.newpos[6 * (size_t)i + k]);
  const int no_g = uni_i(g.ovf_cnt[0]);
  const int no_t = tg.cnt ? uni_i(tg.ovf_cnt[0]) : 0;
  // (the sample's temporary store entry carries the expanded node's tree; its position came with the sample)
  const int mine = uni_i(A.center ? A.tree[A.N0 + i] : A.tree[ex]);
  for (int k = 0; k < 6; ++k) exp[k] = uni_d(A.center ? A.center[6 * (size_t)i + k] : A.pos[6 * (size_t)ex + k]);
  int flags = 0, nnb = 0;
  int used_slots = 0;                     // edge-task slots this sample fills (the others are cleared at the end)
  const double parts0 = edge_parts(exp, qp);   // the parent edge (task slot 0), used by the task write and by the cull
  const int ns0 = edge_samples(parts0);
  const bool mine_shard = A.world <= 1 || i % A.world == A.rank;
  if (inl && mine_shard) {
    flags |= 1;
    int32_t* h_id = s_id[wave];
    int32_t* h_tree = s_tree[wave];
    double* h_d = s_d[wave];
    double* h_pos = s_pos[wave];
    int nh = 0;
    // ---- the cells the query ball's box touches
    const float rf = sqrtf(Q.r2f) * 1.000001f;
    const int lx = grid_coord(Q.x - rf, g.ox, g.inv_cell, g.nx), hx = grid_coord(Q.x + rf, g.ox, g.inv_cell, g.nx);
    const int ly = grid_coord(Q.y - rf, g.oy, g.inv_cell, g.ny), hy = grid_coord(Q.y + rf, g.oy, g.inv_cell, g.ny);
    const int lz = grid_coord(Q.z - rf, g.oz, g.inv_cell, g.nz), hz = grid_coord(Q.z + rf, g.oz, g.inv_cell, g.nz);
    const int wx = hx - lx + 1, wy = hy - ly + 1, wz = hz - lz + 1;
    const int total = wx * wy * wz;
    const float rwx = __frcp_rn((float)wx), rwy = __frcp_rn((float)wy);
    auto scan = [&](QcTop* top) {
    for (int c0 = 0; c0 < total; c0 += 64) {
      const int c = c0 + lane;
      int cell = 0, m = 0, mt = 0;
      if (c < total) {
        // (c, wx, wy <= 512 here: (c + 0.5) / w is at least 0.5 / w ~ 1e-3 away from an integer and the fp32 product
        // is off by less than 1e-4, so it truncates to the exact quotient - three integer divisions are ~90
        // instructions)
        int q1, q2;
        if (total <= 512) {
          q1 = (int)(((float)c + 0.5f) * rwx);
          q2 = (int)(((float)q1 + 0.5f) * rwy);
        } else {
          q1 = c / wx;
          q2 = q1 / wy;
        }
        const int cx = lx + (c - q1 * wx), cy = ly + (q1 - q2 * wy), cz = lz + q2;
        cell = (cz * g.ny + cy) * g.nx + cx;
        // (both counts with one round of loads: the round grid's count array is small and stays in the L2 - going
        // through its occupancy bits first would put one more dependent load into every sample's chain)
        m = g.cnt[cell];
        if (tg.cnt) mt = tg.cnt[cell];
        if (m > g.bk) m = g.bk;
        if (mt > tg.bk) mt = tg.bk;
      }
      qc_candidates(g, tg, m, mt, cell, lane, Q, qp, h_id, h_d, h_tree, h_pos, nh, top);
    }
    if (no_g > 0) qc_overflow(g, no_g, lane, Q, qp, h_id, h_d, h_tree, h_pos, nh, top);
    if (no_t > 0) qc_overflow(tg, no_t, lane, Q, qp, h_id, h_d, h_tree, h_pos, nh, top);
    };
    scan(nullptr);
    // ---- classification (k_classify's logic on the wave's own hit list)
    qt1 = DBG_T();
    int cnt = nh;
    // more hits than lanes (and no test asks for a smaller list): the scan again, keeping the first QC_TOP qualifying hits
    QcTop top{0x7fffffff, 0x7fffffff, 0.0, 0, mine, force, pdist, A.dist_tree};
    const bool topped = cnt > 64 && A.cap >= 64;
    if (topped) {
      scan(&top);
      cnt = top.have;
      if (lane < cnt) {   // (the kept hits' positions: from the store)
        const double* ps = A.pos + 6 * (size_t)top.id;
        for (int k = 0; k < 6; ++k) h_pos[6 * lane + k] = ps[k];
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (cnt > A.cap) {
      flags |= 2;
    } else {
      const bool have = lane < cnt;
      const int id = have ? (topped ? top.id : h_id[lane]) : 0x7fffffff;
      const double d = have ? (topped ? top.d : h_d[lane]) : 0.0;
      const int t = have ? (topped ? top.t : h_tree[lane]) : 0x7fffffff;
      const bool same = t == mine;
      bool q = false;
      if (have) q = same ? (!force && d < pdist - SFFG_TOL)          // src/forest.h:276
                         : (d < A.dist_tree - SFFG_TOL);             // src/forest.h:283
      int rank = 0;
      for (int j = 0; j < cnt; ++j) {   // (lanes >= cnt hold no hit)
        const int tj = __shfl(t, j), idj = __shfl(id, j), qj = __shfl((int)q, j);
        const double dj = __shfl(d, j);
        if (qj && (tj < t || (tj == t && (dj < d || (dj == d && idj < id))))) ++rank;
      }
      int cut = (q && !same && id < A.N0) ? rank : 0x7fffffff;
      for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_xor(cut, off);
        cut = o < cut ? o : cut;
      }
      bool keep = q && rank <= cut;
      int nkeep = __popcll(__ballot(keep));
      if (nkeep > A.nbcap && A.lazy_nb) {
        // more neighbours qualify than the record holds: the first nbcap in the reference's order are kept and the sample
        // is marked (bit 2) - the walk nearly always ends among them (k_commit faults only if it runs off their end)
        flags |= 4;
        keep = keep && rank < A.nbcap;
        nkeep = A.nbcap;
      }
      if (nkeep > A.nbcap) {
        flags |= 2;   // the host path redoes this sample with unbounded lists
      } else {
        nnb = nkeep;
        used_slots = 1 + nkeep;
        if (keep) {
          A.rec_nb[(size_t)i * A.nbcap + rank] = id;
          A.rec_meta[(size_t)i * A.nbcap + rank] = (t << 1) | (same ? 1 : 0);
          const size_t slot = (size_t)i * stride + 1 + rank;
          double nbp[6], ea[6], eb[6];
          for (int k = 0; k < 6; ++k) nbp[k] = h_pos[6 * lane + k];   // (the hit's position came with its grid item)
          if (same) { for (int k = 0; k < 6; ++k) { ea[k] = nbp[k]; eb[k] = qp[k]; } }                 // isPathFree(neighbour, newPoint) :276
          else if (id == A.goal_id) { for (int k = 0; k < 6; ++k) { ea[k] = qp[k]; eb[k] = nbp[k]; } } // isPathFree(newPoint, goal) :287
          else { for (int k = 0; k < 6; ++k) { ea[k] = exp[k]; eb[k] = nbp[k]; } }                     // isPathFree(expanded, neighbour) :288
          double* sa = A.seg_a + 6 * slot;
          double* sb = A.seg_b + 6 * slot;
          for (int k = 0; k < 6; ++k) { sa[k] = ea[k]; sb[k] = eb[k]; }
          const double parts = edge_parts(ea, eb);
          const int ns = edge_samples(parts);
          A.seg_ns[slot] = ns;   // the edge's sample count doubles as its "live" mark
          A.first_hit[slot] = 0x7fffffff;
          A.seg_ovf[slot] = 0;
          if (fused_cull) {
            // the hit list is dead from here on (every lane holds its own hit in registers): the kept edges' start
            // points, sample steps and sample counts take its place, indexed by rank
            // (in cells of the clearance grid, fp32: see the fused cull below)
            const float inv = (float)env.clear_inv * __frcp_rn((float)parts);
            float* tf = reinterpret_cast<float*>(h_pos);
            for (int k = 0; k < 3; ++k) {
              tf[8 * rank + k] = (float)((ea[k] - env.clear_org[k]) * env.clear_inv);
              tf[8 * rank + 4 + k] = (float)(eb[k] - ea[k]) * inv;
            }
            h_id[rank] = ns;
          }
        }
        if (lane == 0) {   // slot 0: isPathFree(expanded, newPoint)  (src/forest.h:246)
          const size_t slot = (size_t)i * stride;
          double* sa = A.seg_a + 6 * slot;
          double* sb = A.seg_b + 6 * slot;
          for (int k = 0; k < 6; ++k) { sa[k] = exp[k]; sb[k] = qp[k]; }
          A.seg_ns[slot] = ns0;
          A.first_hit[slot] = 0x7fffffff;
          A.seg_ovf[slot] = 0;
        }
      }
    }
  }
  for (int k = used_slots + lane; k < stride; k += 64) {   // the task slots this sample does not use
    A.seg_ns[(size_t)i * stride + k] = -1;
    A.first_hit[(size_t)i * stride + k] = 0x7fffffff;
    A.seg_ovf[(size_t)i * stride + k] = 0;
  }
  if (fused_cull) {
    // ---- clearance cull of this sample's own work: its pose and every 64-sample chunk of its edge tasks.  ~97 % are
    // answered "free" by one bit; what is not goes onto the survivor list of the exact kernel.  The (task, chunk)
    // pairs are flattened so that eight dependent lookups are in flight at a time whatever the edge lengths are;
    // survivors are gathered in LDS and reserved on the list with one atomic per sample.
    // Sample positions a + idx * step: far inside the slack of the bits.
    SurvivorItem* list = static_cast<SurvivorItem*>(A.items);
    SurvivorItem* buf = s_surv[wave];
    int n_buf = 0;
    const int sub_list = blockIdx.x & (SFFK_SUBLISTS - 1), sub_cap = A.items_cap / SFFK_SUBLISTS;
    [[maybe_unused]] const unsigned long long qt2 = DBG_T();
    QDBG(0, 1); QDBG(1, qt1 - qt0); QDBG(2, qt2 - qt1);
    auto flush = [&]() {
      [[maybe_unused]] const unsigned long long f0 = DBG_T();
      QDBG(6, n_buf);
      // heavy items (poses, chunks with most of their samples masked: 10-25 us of exact tests) go to the front half of
      // the sub-list, light ones (median 7 us) to the back half: the exact kernel starts the heavy ones first - every
      // wave takes one item and comes back for more, so its length is then the longest item, not "a light one + a long one"
      const bool mine = lane < n_buf;                 // (n_buf <= 64)
      SurvivorItem it = mine ? buf[lane] : SurvivorItem{0, 0, 0ULL};
      const bool hv = mine && (it.slot < 0 || __popcll(it.mask) >= QC_HEAVY);
      const unsigned long long hm = __ballot(hv), lm = __ballot(mine && !hv);
      const unsigned long long below = (1ULL << lane) - 1ULL;
      int bh = 0, bl = 0;
      if (lane == 0) {
        if (hm) bh = atomicAdd(A.sub + sub_list * SFFK_SUB_STRIDE, __popcll(hm));
        if (lm) bl = atomicAdd(A.sub + sub_list * SFFK_SUB_STRIDE + 2, __popcll(lm));
      }
      bh = __shfl(bh, 0); bl = __shfl(bl, 0);
      const int half = sub_cap / 2;
      if (hv) {
        const int at = bh + __popcll(hm & below);
        if (at < half) list[(size_t)sub_list * sub_cap + at] = it;
        else A.ctrl[3] = 1;
      } else if (mine) {
        const int at = bl + __popcll(lm & below);
        if (at < sub_cap - half) list[(size_t)sub_list * sub_cap + half + at] = it;
        else A.ctrl[3] = 1;
      }
      n_buf = 0;
      qt_fl += DBG_T() - f0;
    };
    const bool live = (flags & 3) == 1 && env.n_tri != 0;
    if (lane == 0) A.pose_hit[i] = 0;
    if (live) {
      // the pose's own bit: the load is issued here and looked at after the edges' (one latency for everything)
      const uint32_t* wp_pose = nullptr;
      int sh_pose = 0;
      bool need_pose = true;
      if (env.clear_bits) {
        const double fx = (qp[0] - env.clear_org[0]) * env.clear_inv, fy = (qp[1] - env.clear_org[1]) * env.clear_inv,
                     fz = (qp[2] - env.clear_org[2]) * env.clear_inv;
        if (fx == fx && fy == fy && fz == fz) {
          if (fx < 0 || fy < 0 || fz < 0 || fx >= env.clear_n[0] || fy >= env.clear_n[1] || fz >= env.clear_n[2]) {
            need_pose = false;
          } else {
            const long long ci = ((long long)(int)fz * env.clear_n[1] + (int)fy) * env.clear_n[0] + (int)fx;
            wp_pose = env.clear_bits + (ci >> 5);
            sh_pose = (int)(ci & 31);
          }
        }
      }
      const uint32_t word_pose = wp_pose ? *wp_pose : 0u;
      // lane r < nnb = the kept edge of rank r; the parent edge (task 0) is uniform.  Positions in cells of the
      // clearance grid, in fp32: cell = a + idx * step is off the exact kernel's fp64 sample position by less than
      // 1e-6 * (cells per axis) cells - the slack Ctx::build_clearance puts into the bits for exactly this
      // (steps in fp32 with the hardware reciprocal: a relative 1e-7 on a step that spans a few cells is far inside
      // the slack of the bits)
      const float inv0 = (float)env.clear_inv * __frcp_rn((float)parts0);
      const float g0[3] = {(float)((exp[0] - env.clear_org[0]) * env.clear_inv), (float)((exp[1] - env.clear_org[1]) * env.clear_inv),
                           (float)((exp[2] - env.clear_org[2]) * env.clear_inv)};
      const float st0[3] = {(float)(qp[0] - exp[0]) * inv0, (float)(qp[1] - exp[1]) * inv0, (float)(qp[2] - exp[2]) * inv0};
      const int C0 = ns0 > 0 ? (ns0 + 63) >> 6 : 0;
      const int my_ns = lane < nnb ? s_id[wave][lane] : 0;
      const int my_nch = my_ns > 0 ? (my_ns + 63) >> 6 : 0;
      int incl = my_nch;
      for (int off = 1; off < nnb; off <<= 1) {
        const int o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
      }
      const int excl = incl - my_nch;
      const int P = C0 + (nnb > 0 ? __shfl(incl, nnb - 1) : 0);
      const float* T = reinterpret_cast<const float*>(s_pos[wave]);
      const int32_t* NS = s_id[wave];
      int32_t* tab = s_tab[wave];
      const float nxd = (float)env.clear_n[0], nyd = (float)env.clear_n[1], nzd = (float)env.clear_n[2];
      // Eight consecutive samples of an edge lie within 0.4 units of the fifth one (the sample spacing never exceeds
      // the 0.1 of src/problemStruct.h:121), and the bits are built with that reach on top (Ctx::build_clearance): ONE
      // lookup answers a group of eight samples, a lane takes a group, a step of the wave eight (task, chunk) pairs.
      // A group that is not clear goes to the exact kernel whole.  tab[]: pair -> (task << 16 | chunk), built in
      // windows of QC_TAB pairs.
      const int pu = lane >> 3, g = lane & 7;
      for (int w0 = 0; w0 < P; w0 += QC_TAB) {
        __builtin_amdgcn_wave_barrier();
        for (int c = lane; c < C0; c += 64)
          if (c >= w0 && c < w0 + QC_TAB) tab[c - w0] = c;                                      // the parent edge: task 0
        if (lane < nnb)
          for (int c = 0; c < my_nch; ++c) {
            const int p = C0 + excl + c;
            if (p >= w0 && p < w0 + QC_TAB) tab[p - w0] = ((1 + lane) << 16) | (c & 0xffff);
          }
        __builtin_amdgcn_wave_barrier();
        const int wn = P - w0 < QC_TAB ? P - w0 : QC_TAB;
        for (int q0 = 0; q0 < wn; q0 += 8) {
          const bool valid = q0 + pu < wn;
          const int ent = valid ? tab[q0 + pu] : 0;
          const int t = ent >> 16, c = ent & 0xffff;
          float a0 = g0[0], a1 = g0[1], a2 = g0[2], d0 = st0[0], d1 = st0[1], d2 = st0[2];
          int ns = ns0;
          if (t > 0) {
            const float* tt = T + 8 * (t - 1);
            a0 = tt[0]; a1 = tt[1]; a2 = tt[2]; d0 = tt[4]; d1 = tt[5]; d2 = tt[6];
            ns = NS[t - 1];
          }
          const int first = 1 + 64 * c + 8 * g;                 // the group's first sample
          bool need = valid && first <= ns;
          const int left = ns - first + 1;                      // valid samples from there on
          const int probe = first + 4 <= ns ? first + 4 : ns;   // within four steps of every valid sample of the group
          const uint32_t* wp = nullptr;
          int sh = 0;
          if (need && env.clear_bits_edge) {
            const float td = (float)probe;
            const float fx = __builtin_fmaf(td, d0, a0), fy = __builtin_fmaf(td, d1, a1), fz = __builtin_fmaf(td, d2, a2);
            if (fx >= 0 && fy >= 0 && fz >= 0 && fx < nxd && fy < nyd && fz < nzd) {   // (the grid has fewer than 2^31 cells)
              const uint32_t ci = ((uint32_t)(int)fz * (uint32_t)env.clear_n[1] + (uint32_t)(int)fy) * (uint32_t)env.clear_n[0] + (uint32_t)(int)fx;
              wp = env.clear_bits_edge + (ci >> 5);
              sh = (int)(ci & 31u);
            } else if (fx == fx && fy == fy && fz == fz) {
              need = false;                                     // beyond the inflated box of the environment
            }
          }
          const uint32_t word = wp ? *wp : 0u;
          if (wp && ((word >> sh) & 1u)) need = false;
          unsigned long long m = need ? (((left >= 8 ? 0xffULL : ((1ULL << left) - 1ULL))) << (8 * g)) : 0ULL;
          m |= __shfl_xor(m, 1);
          m |= __shfl_xor(m, 2);
          m |= __shfl_xor(m, 4);
          const bool lead = g == 0 && m != 0ULL;
          if (__any(lead)) {
            surv_emit(buf, n_buf, lead, lane, (int32_t)(i * stride + t), c, m);
            if (n_buf > QC_SURV - 33) flush();
          }
        }
      }
      if (need_pose && !((word_pose >> sh_pose) & 1u)) {   // (uniform)
        if (lane == 0) buf[n_buf] = SurvivorItem{-1 - i, 0, 0ULL};
        ++n_buf;   // (flushed at 64: there is room)
      }
      if (n_buf) flush();
      QDBG(5, P); QDBG(7, 1);
    }
    QDBG(3, DBG_T() - qt2); QDBG(4, qt_fl);
  }
  if (lane == 0) {
    A.rec_flags[i] = flags;
    A.rec_nnb[i] = nnb;
  }
  // (every workgroup: its last wavefront out is not known, so every wavefront's lane 0 reports)
  if (lane == 0 && A.qclk_sh) atomicMax(A.qclk_sh + (blockIdx.x & 63) * 16, wall_clock64());
}

// ------------------------------------------------------------------ block query kernel
// k_query_classify's work (neighbour query + classification + clearance cull of a round's samples) organised as flat work
// lists of a 256-thread workgroup that serves QB_S samples.  One wavefront per sample runs most of its instructions with a
// handful of useful lanes - 27 cells, ~34 candidates, five hits, two kept edges, seven (task, chunk) pairs per sample - and
// reads every candidate as a 64-byte item.  Here every phase runs lane = work item over ALL the workgroup's samples, with
// the lists in LDS, and:
//   - candidates are 32-byte fp32 filter records (GridView::lite); the authoritative fp64 position is fetched from the
//     store only for the few that pass the superset filter (the sweep's own filter: k_sweep);
//   - what depends on the sample alone (cell box, parent edge in clearance-grid cells, ...) comes as a QRec from the kernel
//     that drew the sample; the clearance bits of the parent edge's first four chunks and of the pose are requested
//     together with the cell counts and looked at after the classification;
//   - the round's own grid is asked through its occupancy bits (96 KB, cache-resident) instead of its count array;
//   - edge end points are written only for the tasks that leave a survivor for the exact kernel (k_collide_items derives
//     them from the records when the survivor list ran over), unused task slots are not cleared (readers trust rec_nnb).
// Measured on the bench job: 28.0 -> 23.1 us per launch; two samples per wavefront (32 lanes each) 24.8 us.
// Phases:
//   0  the samples' scalars (one thread per sample) -> LDS; cell box, parent edge in clearance-grid cells
//   1  lane = (sample, cell): count, up to three filter records of the bucket and the first one of the round's own grid
//      (where its occupancy bit is set) at once, fp32 superset filter -> candidate list; deeper buckets -> a second work
//      list; beside it lane = one group of eight samples of the parent edges' first four chunks: clearance bit requested
//   2  lane = entry of the second work list (rare), the grids' overflow lists
//   3  lane = candidate: authoritative fp64 position from the store, exact distance -> per-sample hit lists
//   4  lane = (sample, hit): classification (k_classify's logic: rank by a loop over the sample's hits in LDS, the cut
//      by an LDS atomicMin), neighbour records, the kept edges' cull rows
//   5  lane = edge task: its 64-sample chunks -> (task, chunk) pair table
//   6  lane = one group of eight samples of a pair: clearance bit -> masks -> survivors (LDS, one pair of atomics per
//      workgroup on the exact kernel's list)
//   7  end points of the tasks that left a survivor
// QB_S samples per workgroup: the instruction stream of a phase is paid once per 64 work items whatever sample they
// belong to - with 8 samples per workgroup (two per wavefront, like k_query_block) the vector ALUs were as busy as before.
// Bounded lists: QB_HC exact hits per sample, 36 filter candidates and 128 bucket records per sample on average over
// the workgroup (packed entries, round 5: a workgroup's samples are spatial neighbours and dense together); a sample that loses an entry gets flag 2 (host path), like every list overflow.
#define QB_S 8
#define QB_HC 24
#define QB_TASKS 17
#ifndef QB_OCC
#define QB_OCC 8   // 8 workgroups per CU: a round's first launches bring 2 048 workgroups, and a launch that does not fit the
                   // chip at once takes twice a workgroup's lifetime (5 per CU = 1 280 slots for 1 275 workgroups: 27 us)
#endif
// the positions of `want` lanes in an LDS list (one atomic per wavefront); -1 for the others
__device__ __forceinline__ int wave_reserve(int* counter, bool want, int lane) {
  const unsigned long long m = __ballot(want);
  if (!m) return -1;
  const int leader = __ffsll((long long)m) - 1;
  int base = 0;
  if (lane == leader) base = atomicAdd(counter, __popcll(m));
  base = __shfl(base, leader);
  return want ? base + __popcll(m & ((1ULL << lane) - 1ULL)) : -1;
}
// n entries per lane (converged code only): first position
__device__ __forceinline__ int wave_reserve_n(int* counter, int n, int lane) {
  int inc = n;
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(inc, off);
    if (lane >= off) inc += o;
  }
  const int total = __shfl(inc, 63);
  if (!total) return 0;
  int base = 0;
  if (lane == 63) base = atomicAdd(counter, total);
  base = __shfl(base, 63);
  return base + inc - n;
}
// words of a sample's record in LDS: QRec as the sampling kernel wrote it, the last four words are the kernel's own
enum { QI_MAXID = 7, QI_MINE = 8, QI_EVAL = 9, QI_QTREE = 10, QI_TOTAL = 11, QI_LX = 12, QI_LY = 13, QI_LZ = 14, QI_WX = 15, QI_NS0 = 16,
       QI_T0 = 17, QI_WY = 23, QI_PD = 24, QI_QR = 26, QI_FLAGS = 28, QI_NNB = 29, QI_FORCE = 30, QI_LIVE = 31 };
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(QB_OCC))) void k_query_block(
    GridView g, GridView tg, const SweepQuery* __restrict__ queries, ClassifyArgs A, EnvView env) {
#ifndef QB_INL
#define QB_INL 0      // records of a bucket looked at in the cell pass itself; the rest through the second work list
#endif
#ifndef QB_W2
#define QB_W2 128
#endif
  // (the work lists are POOLED over the workgroup's samples; since the samples of a workgroup are spatial neighbours - OrderView -
  // they are dense together: entries are packed - sample in the top bits - so that the same LDS holds 36 candidates and 128
  // bucket records per sample instead of 24 and 64)
  constexpr int S = QB_S, HC = QB_HC, CANDCAP = S * 36, W2CAP = S * QB_W2, PAIRCAP = S * 32, SURVCAP = S * 8, INL = QB_INL;
  __shared__ __attribute__((aligned(16))) int s_i[S][32];   // QRec
  __shared__ double s_qp[S][6], s_ex[S][6];
  __shared__ int s_pref[S + 1];
  __shared__ int s_nhit[S], s_drop[S];
  __shared__ double h_d[S][HC];
  __shared__ int h_id[S][HC], h_tree[S][HC];
  __shared__ int s_rankhit[S][QB_TASKS];
  __shared__ int c_id[CANDCAP], c_tree[CANDCAP];   // c_tree: sample << 28 | tree
  __shared__ int w_at[W2CAP];                      // sample << 28 | round's own grid << 27 | record index (< 2^27: query_block_mode)
  __shared__ float s_T[S][QB_TASKS][8];
  __shared__ int s_NS[S][QB_TASKS], s_need[S][QB_TASKS];
  __shared__ int s_tab[PAIRCAP];
  __shared__ SurvivorItem s_surv[SURVCAP];
  __shared__ int s_cnt[8];               // 0 candidates, 1 second work list, 2 survivors, 3 heavy base, 4 light base
  __shared__ int s_wsum[4];
  if (A.dev_n) {
    if (A.dev_n[1]) return;
    A.n = A.dev_n[0];
  }
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // which samples: 8 consecutive sample indices, or - device engine, OrderView - entries of the sub-range lists of the
  // wave's spatial order, the sub-ranges dealt out so that every XCD (workgroup b runs on XCD b % 8) gets one contiguous
  // run of them: neighbouring samples then find each other's count, bucket and clearance lines in that XCD's L2
  __shared__ int s_map[S];
  {
    const bool ordered = A.ord_valid && *A.ord_valid;
    int i = -1;
    if (ordered) {
      // XCD x owns the sub-ranges [x perx, (x + 1) perx) - one contiguous piece of the wave's spatial order; a workgroup
      // takes ONE entry (number j) from each of 8 sub-ranges spread over that piece: its samples share the XCD's L2 with
      // their spatial neighbours, but a dense region's samples do not all land in the same workgroups (8 entries of ONE
      // list per workgroup made the dense regions' workgroups the launch's stragglers: SFF* at 2 M nodes 603 -> 654 ms)
      const int n_slots = *A.ord_nslots, sel = *A.ord_sel, R = (n_slots + 63) >> 6, perx = (R + 7) >> 3, nsets = (perx + 7) >> 3;
      const int y = (int)blockIdx.x >> 3, gset = y >> 6, j = y & 63;
      if (gset >= nsets) return;
      // (no run-time index into the argument struct: that would move all of it to scratch memory)
      const int32_t* const o_cnt = sel ? A.ord_cnt[1] : A.ord_cnt[0];
      const int32_t* const o_lst = sel ? A.ord_lst[1] : A.ord_lst[0];
      if (tid < S) {
        const int off = gset + nsets * tid, r = ((int)blockIdx.x & 7) * perx + off;
        if (off < perx && r < R) {
          const int cnt = o_cnt[r * SFFK_ORD_CNT_STRIDE];
          const int e = o_lst[r * 64 + j];   // (requested beside the count)
          if (j < cnt) i = e;
        }
        if (i >= A.n) i = -1;
      }
    } else {
      if ((int)blockIdx.x * S >= A.n) return;
      if (tid < S) { i = blockIdx.x * S + tid; if (i >= A.n) i = -1; }
    }
    if (tid < S) s_map[tid] = i;
    __syncthreads();
    bool any = false;
    for (int q = 0; q < S; ++q) any = any || s_map[q] >= 0;
    if (!any) return;
  }
  const bool clocked = A.qclk_sh && tid == 0;   // (every workgroup, into its shard: see DevForestView::qclk_sh)
  if (clocked) atomicMin(A.qclk_sh + (blockIdx.x & 63) * 16 + 1, wall_clock64());
  [[maybe_unused]] const bool qdbg_on = (blockIdx.x & 15) == 0 && tid == 0;
  [[maybe_unused]] unsigned long long qtp = DBG_T();
#ifdef SFFK_DEBUG_COUNTERS
  const unsigned long long qb_c0 = clock64(), qb_r0 = qtp;
#endif
#define QB_MARK(k) do { [[maybe_unused]] const unsigned long long t_ = DBG_T(); QDBG(k, t_ - qtp); qtp = t_; } while (0)
  QDBG(0, 1);
  const int stride = 1 + A.nbcap;
  const float nxd = (float)env.clear_n[0], nyd = (float)env.clear_n[1], nzd = (float)env.clear_n[2];
  const bool have_env = env.n_tri != 0;
  auto group_addr = [&](bool valid, const float* tt, int ns, int c, int gi, bool& need, int& left, const uint32_t*& wp, int& sh) {
    const int first = 1 + 64 * c + 8 * gi;
    need = valid && first <= ns;
    left = ns - first + 1;
    const int probe = first + 4 <= ns ? first + 4 : ns;
    wp = nullptr;
    sh = 0;
    if (need && env.clear_bits_edge) {
      const float td = (float)probe;
      const float fx = __builtin_fmaf(td, tt[4], tt[0]), fy = __builtin_fmaf(td, tt[5], tt[1]), fz = __builtin_fmaf(td, tt[6], tt[2]);
      if (fx >= 0 && fy >= 0 && fz >= 0 && fx < nxd && fy < nyd && fz < nzd) {
        const uint32_t ci = ((uint32_t)(int)fz * (uint32_t)env.clear_n[1] + (uint32_t)(int)fy) * (uint32_t)env.clear_n[0] + (uint32_t)(int)fx;
        wp = env.clear_bits_edge + (ci >> 5);
        sh = (int)(ci & 31u);
      } else if (fx == fx && fy == fy && fz == fz) {
        need = false;                                     // beyond the inflated box of the environment
      }
    }
  };
  // ---- 0. the samples' records (written by the sampling kernel: one coalesced word per thread), their fp64 positions
  auto RF = [&](int s, int w) -> float { return __int_as_float(s_i[s][w]); };
  auto RD = [&](int s, int w) -> double { return *reinterpret_cast<const double*>(&s_i[s][w]); };
  if (tid < 5) s_cnt[tid] = 0;
  if (tid == 255) {   // the grids' overflow lists (empty unless a cell holds more nodes than its bucket)
    s_cnt[5] = g.ovf_cnt[0] < g.ovf_cap ? g.ovf_cnt[0] : g.ovf_cap;
    s_cnt[6] = tg.cnt ? (tg.ovf_cnt[0] < tg.ovf_cap ? tg.ovf_cnt[0] : tg.ovf_cap) : 0;
  }
  for (int e = tid; e < S * QB_TASKS; e += 256) (&s_need[0][0])[e] = 0;
  for (int e = tid; e < S * 32; e += 256) {
    const int s = e >> 5, w = e & 31, i = s_map[s];
    int v = i >= 0 ? reinterpret_cast<const int32_t*>(A.qrec)[(size_t)i * 32 + w] : 0;
    if (w == QI_FLAGS) v = (i >= 0 && reinterpret_cast<const int32_t*>(A.qrec)[(size_t)i * 32 + QI_EVAL]) ? 1 : 0;
    if (w == QI_FORCE) v = i >= 0 && A.force[i] != 0;
    if (w == QI_NNB || w == QI_LIVE) v = 0;
    s_i[s][w] = v;
  }
  for (int e = tid; e < S * 12; e += 256) {
    const int s = e / 12, k = e - s * 12, i = s_map[s];
    double v = 0.0;
    if (i >= 0) {
      if (k < 6) v = A.newpos[6 * (size_t)i + k];
      else v = A.center ? A.center[6 * (size_t)i + k - 6] : A.pos[6 * (size_t)A.parent[i] + k - 6];
    }
    if (k < 6) s_qp[s][k] = v; else s_ex[s][k - 6] = v;
  }
  uint32_t word_pose = 0u;     // (thread s keeps its sample's pose bit to the end)
  int sh_pose = 0;
  bool need_pose = false;
  if (tid < S) {
    const int s = tid, i = s_map[s];
    const bool act = i >= 0;
    const QRec* rec = A.qrec + (act ? i : 0);
    const bool evaluate = act && rec->evaluate != 0;
    const int total = evaluate ? rec->total : 0;
    s_nhit[s] = 0; s_drop[s] = 0;
    for (int k = 0; k < 3; ++k) { s_T[s][0][k] = rec->t0[k]; s_T[s][0][4 + k] = rec->t0[3 + k]; }
    s_NS[s][0] = rec->ns0;
    int inc = total;
    for (int off = 1; off < S; off <<= 1) {
      const int o = __shfl_up(inc, off);
      if (s >= off) inc += o;
    }
    s_pref[s] = inc - total;
    if (s == S - 1) s_pref[S] = inc;
    // the pose's own clearance bit
    need_pose = evaluate && have_env;
    if (need_pose && env.clear_bits) {
      const double* q3 = A.newpos + 6 * (size_t)i;
      const double fx = (q3[0] - env.clear_org[0]) * env.clear_inv, fy = (q3[1] - env.clear_org[1]) * env.clear_inv,
                   fz = (q3[2] - env.clear_org[2]) * env.clear_inv;
      if (fx == fx && fy == fy && fz == fz) {
        if (fx < 0 || fy < 0 || fz < 0 || fx >= env.clear_n[0] || fy >= env.clear_n[1] || fz >= env.clear_n[2]) {
          need_pose = false;
        } else {
          const long long ci = ((long long)(int)fz * env.clear_n[1] + (int)fy) * env.clear_n[0] + (int)fx;
          word_pose = env.clear_bits[ci >> 5];
          sh_pose = (int)(ci & 31);
        }
      }
    }
  }
  __syncthreads();
  QB_MARK(1);
  // ---- 1. requested now, settled in phase 6: the parent edges' first four chunks, thread = (sample, chunk, group)
  bool e_need = false;
  int e_left = 0, e_sh = 0;
  const uint32_t* e_wp = nullptr;
  const int e_s = tid >> 5, e_c = (tid >> 3) & 3;
  if (e_s < S)
    group_addr(s_i[e_s][QI_EVAL] && have_env && e_c < ((s_i[e_s][QI_NS0] + 63) >> 6), s_T[e_s][0], s_i[e_s][QI_NS0], e_c, tid & 7, e_need, e_left, e_wp, e_sh);
  const uint32_t e_word = e_wp ? *e_wp : 0u;
  // fp32 superset filter (the sweep's own: k_sweep)
  auto passes = [&](int s, const GridItem32& it) -> bool {
    if (!(it.id < s_i[s][QI_MAXID]) || (s_i[s][QI_QTREE] >= 0 && it.tree != s_i[s][QI_QTREE])) return false;
    const float r2f = RF(s, 6);
    const float dx = it.x - RF(s, 0), dy = it.y - RF(s, 1), dz = it.z - RF(s, 2);
    const float d3 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    if (!(d3 <= r2f)) return false;
    const float da = wrapf(it.yaw - RF(s, 3)), db = wrapf(it.pitch - RF(s, 4)), dc = wrapf(it.roll - RF(s, 5));
    return fmaf(dc, dc, fmaf(db, db, fmaf(da, da, d3))) <= r2f;
  };
  auto add_cand = [&](bool pass, int s, const GridItem32& it) {
    const int at = wave_reserve(&s_cnt[0], pass, lane);
    if (pass) {
      if (at < CANDCAP) { c_id[at] = it.id; c_tree[at] = (s << 28) | (it.tree & 0x0fffffff); }
      else s_drop[s] = 1;
    }
  };
  {
    const int E = s_pref[S];
    for (int e0 = 0; e0 < E; e0 += 256) {
      const int e = e0 + tid;
      const bool on = e < E;
      int s = 0;
      size_t cell = 0;
      int m = 0;
      uint32_t ow = 0u;
      if (on) {
        for (int step = S >> 1; step > 0; step >>= 1)
          if (s_pref[s + step] <= e) s += step;
        const int c = e - s_pref[s];
        const int wx = s_i[s][QI_WX], wy = s_i[s][QI_WY];
        int q1, q2;
        if (s_i[s][QI_TOTAL] <= 512) {   // (exact: see k_query_classify)
          q1 = (int)(((float)c + 0.5f) * __frcp_rn((float)wx));
          q2 = (int)(((float)q1 + 0.5f) * __frcp_rn((float)wy));
        } else {
          q1 = c / wx;
          q2 = q1 / wy;
        }
        const int cx = s_i[s][QI_LX] + (c - q1 * wx), cy = s_i[s][QI_LY] + (q1 - q2 * wy), cz = s_i[s][QI_LZ] + q2;
        cell = ((size_t)cz * g.ny + cy) * g.nx + cx;
        m = g.cnt[cell];
        if (tg.cnt) ow = tg.occ ? tg.occ[cell >> 5] : 0xffffffffu;
        if (m > g.bk) m = g.bk;
      }
      // up to three records of the bucket and the first one of the round's own grid (where its occupancy bit is set) are
      // requested at once; deeper buckets go through the second work list (>= 0: record of the node grid; < 0: -1 - record
      // of the round's own grid)
      GridItem32 i0{}, i1{}, i2{}, t0{};
      const GridItem32* bucket = g.lite + cell * g.bk;
      if (INL > 0 && m > 0) i0 = bucket[0];
      if (INL > 1 && m > 1) i1 = bucket[1];
      if (INL > 2 && m > 2) i2 = bucket[2];
      int mt = 0;
      constexpr int TINL = INL > 0 ? 1 : 0;
      if (on && ((ow >> (cell & 31)) & 1u)) {
        mt = tg.cnt[cell];
        if (TINL) t0 = tg.lite[cell * tg.bk];
        if (mt > tg.bk) mt = tg.bk;
      }
      const int extra = (m > INL ? m - INL : 0) + (mt > TINL ? mt - TINL : 0);
      if (__any(extra > 0)) {
        int at = wave_reserve_n(&s_cnt[1], extra, lane);
        for (int k = INL; k < m; ++k, ++at) {
          if (at < W2CAP) w_at[at] = (s << 28) | ((int)(cell * g.bk) + k);
          else s_drop[s] = 1;
        }
        for (int k = TINL; k < mt; ++k, ++at) {
          if (at < W2CAP) w_at[at] = (s << 28) | (1 << 27) | ((int)(cell * tg.bk) + k);
          else s_drop[s] = 1;
        }
      }
      if (INL > 0) add_cand(m > 0 && passes(s, i0), s, i0);
      if (INL > 1 && __any(m > 1)) add_cand(m > 1 && passes(s, i1), s, i1);
      if (INL > 2 && __any(m > 2)) add_cand(m > 2 && passes(s, i2), s, i2);
      if (TINL && __any(mt > 0)) add_cand(mt > 0 && passes(s, t0), s, t0);
    }
  }
  __syncthreads();
  QB_MARK(2);
  // ---- 2. the second work list; the grids' overflow lists
  {
    const int n2 = s_cnt[1] < W2CAP ? s_cnt[1] : W2CAP;
    for (int e0 = 0; e0 < n2; e0 += 256) {
      const int e = e0 + tid;
      const bool on = e < n2;
      const int ent = on ? w_at[e] : 0;
      const int s = (ent >> 28) & 7, at = (ent & (1 << 27)) ? -1 - (ent & 0x07ffffff) : (ent & 0x07ffffff);
      GridItem32 it{};
      if (on) it = at >= 0 ? g.lite[at] : tg.lite[-1 - at];
      add_cand(on && passes(s, it), s, it);
    }
    const int no_g = s_cnt[5], no_t = s_cnt[6];   // (requested in phase 0)
    for (int e0 = 0; e0 < S * (no_g + no_t); e0 += 256) {
      const int e = e0 + tid;
      const bool on = e < S * (no_g + no_t);
      const int s = on ? e / (no_g + no_t) : 0, j = on ? e - s * (no_g + no_t) : 0;
      GridItem32 it{};
      const bool live_s = on && s_i[s][QI_EVAL];
      if (live_s) it = j < no_g ? g.ovf_lite[j] : tg.ovf_lite[j - no_g];
      add_cand(live_s && passes(s, it), s, it);
    }
  }
  __syncthreads();
  QB_MARK(3);
  // ---- 3. exact test of the candidates on the authoritative fp64 positions
  {
    const int nc = s_cnt[0] < CANDCAP ? s_cnt[0] : CANDCAP;
    for (int e = tid; e < nc; e += 256) {
      const int s = (c_tree[e] >> 28) & 7, id = c_id[e];
      double nbp[6], qp[6];
      const double* ps = A.pos + 6 * (size_t)id;
      for (int k = 0; k < 6; ++k) { nbp[k] = ps[k]; qp[k] = s_qp[s][k]; }
      const double d = dist6(nbp, qp);
      if (d < RD(s, QI_QR)) {
        const int slot = atomicAdd(&s_nhit[s], 1);
        if (slot < HC) {
          h_d[s][slot] = d; h_id[s][slot] = id; h_tree[s][slot] = c_tree[e] & 0x0fffffff;
        }
      }
    }
  }
  __syncthreads();
  QB_MARK(4);
  // ---- 4. classification (k_classify's logic), lane = hit: 16 lanes per sample when no sample of the workgroup has more
  // than 16 hits (the usual case: ~5) - two wavefronts do the pass, the other two skip it -, else 32 lanes per sample
  int gsh = 4;
  for (int q = 0; q < S; ++q) if (s_nhit[q] > 16) gsh = 5;
  if (tid < (S << gsh)) {
    const int gw = 1 << gsh;
    const int s = tid >> gsh, hl = tid & (gw - 1);
    const int i = s_map[s];
    const int gbase = lane & ~(gw - 1);
    const uint32_t gmask = gsh == 5 ? 0xffffffffu : 0xffffu;
    auto hballot = [&](bool p) -> uint32_t { return (uint32_t)(__ballot(p) >> gbase) & gmask; };
    int flags = s_i[s][QI_FLAGS];
    const bool evaluate = s_i[s][QI_EVAL] != 0;
    const int n = s_nhit[s];
    if (evaluate && (s_drop[s] || n > HC || n > A.cap)) flags |= 2;
    const bool hit = evaluate && !(flags & 2) && hl < n;
    const int mine = s_i[s][QI_MINE];
    const double d = hit ? h_d[s][hl] : 0.0;
    const int t = hit ? h_tree[s][hl] : 0x7fffffff;
    const int id = hit ? h_id[s][hl] : 0x7fffffff;
    const bool same = t == mine;
    const double pdist = RD(s, QI_PD);
    bool q = false;
    if (hit) q = same ? (!s_i[s][QI_FORCE] && d < pdist - SFFG_TOL)     // src/forest.h:276
                      : (d < A.dist_tree - SFFG_TOL);                  // src/forest.h:283
    int rank = 0;
    uint32_t mm = hballot(q);
    while (__any(mm != 0u)) {
      const bool on = mm != 0u;
      const int src = gbase + (on ? __ffs((int)mm) - 1 : 0);
      mm &= mm - 1u;
      const int tj = __shfl(t, src), idj = __shfl(id, src);
      const double dj = __shfl(d, src);
      if (on && (tj < t || (tj == t && (dj < d || (dj == d && idj < id))))) ++rank;
    }
    int cut = (q && !same && id < A.N0) ? rank : 0x7fffffff;
    for (int off = gw >> 1; off > 0; off >>= 1) {
      const int o = __shfl_xor(cut, off);
      cut = o < cut ? o : cut;
    }
    bool keep = q && rank <= cut;
    int nnb = __popc(hballot(keep));
    if (nnb > A.nbcap && A.lazy_nb) {   // (see k_query_classify: the first nbcap neighbours, bit 2 marks the cut)
      flags |= 4;
      keep = keep && rank < A.nbcap;
      nnb = A.nbcap;
    }
    if (nnb > A.nbcap) { flags |= 2; keep = false; nnb = 0; }
    const bool live = (flags & 3) == 1 && have_env;
    if (keep) {
      A.rec_nb[(size_t)i * A.nbcap + rank] = id;
      A.rec_meta[(size_t)i * A.nbcap + rank] = (t << 1) | (same ? 1 : 0);
      double ea[6], eb[6], nbp[6];
      const double* ps = A.pos + 6 * (size_t)id;   // (read a moment ago by this workgroup's exact test: cache-resident)
      for (int k = 0; k < 6; ++k) nbp[k] = ps[k];
      if (same) { for (int k = 0; k < 6; ++k) { ea[k] = nbp[k]; eb[k] = s_qp[s][k]; } }                   // isPathFree(neighbour, newPoint) :276
      else if (id == A.goal_id) { for (int k = 0; k < 6; ++k) { ea[k] = s_qp[s][k]; eb[k] = nbp[k]; } }  // isPathFree(newPoint, goal) :287
      else { for (int k = 0; k < 6; ++k) { ea[k] = s_ex[s][k]; eb[k] = nbp[k]; } }                        // isPathFree(expanded, neighbour) :288
      const double parts = edge_parts(ea, eb);
      const int ns = edge_samples(parts);
      const size_t slot = (size_t)i * stride + 1 + rank;
      A.seg_ns[slot] = ns;
      A.first_hit[slot] = 0x7fffffff;
      A.seg_ovf[slot] = 0;
      const float inv = (float)env.clear_inv * __frcp_rn((float)parts);
      for (int k = 0; k < 3; ++k) {
        s_T[s][1 + rank][k] = (float)((ea[k] - env.clear_org[k]) * env.clear_inv);
        s_T[s][1 + rank][4 + k] = (float)(eb[k] - ea[k]) * inv;
      }
      s_NS[s][1 + rank] = ns;
      s_rankhit[s][1 + rank] = hl;
    }
    if (hl == 0 && i >= 0) {
      if ((flags & 3) == 1) {   // slot 0: isPathFree(expanded, newPoint)  (src/forest.h:246)
        const size_t slot = (size_t)i * stride;
        A.seg_ns[slot] = s_i[s][QI_NS0];
        A.first_hit[slot] = 0x7fffffff;
        A.seg_ovf[slot] = 0;
      }
      A.pose_hit[i] = 0;
      A.rec_flags[i] = flags;
      A.rec_nnb[i] = nnb;
      s_i[s][QI_NNB] = nnb;
      s_i[s][QI_LIVE] = live ? 1 : 0;
    }
  }
  __syncthreads();
  QB_MARK(5);
  // ---- 5 + 6. (task, chunk) pairs of the live samples -> groups of eight samples -> clearance bits -> survivors
  auto add_surv = [&](bool lead, int slot, int c, unsigned long long m) {
    const int at = wave_reserve(&s_cnt[2], lead, lane);
    if (lead) {
      const SurvivorItem it{slot, c, m};
      if (at < SURVCAP) s_surv[at] = it;
      else {   // (the workgroup's buffer is full: straight onto the exact kernel's list)
        SurvivorItem* list = static_cast<SurvivorItem*>(A.items);
        const int sub_list = blockIdx.x & (SFFK_SUBLISTS - 1), sub_cap = A.items_cap / SFFK_SUBLISTS, half = sub_cap / 2;
        const bool hv = slot < 0 || __popcll(m) >= QC_HEAVY;
        const int p = atomicAdd(A.sub + sub_list * SFFK_SUB_STRIDE + (hv ? 0 : 2), 1);
        if (p < (hv ? half : sub_cap - half)) list[(size_t)sub_list * sub_cap + (hv ? 0 : half) + p] = it;
        else A.ctrl[3] = 1;
      }
    }
  };
  auto settle = [&](bool need, int left, const uint32_t* wp, int sh, uint32_t word, int s, int t, int c) {
    const int gi = lane & 7;
    if (wp && ((word >> sh) & 1u)) need = false;
    unsigned long long m = need ? (((left >= 8 ? 0xffULL : ((1ULL << left) - 1ULL))) << (8 * gi)) : 0ULL;
    m |= __shfl_xor(m, 1);
    m |= __shfl_xor(m, 2);
    m |= __shfl_xor(m, 4);
    const bool lead = gi == 0 && m != 0ULL;
    if (lead) s_need[s][t] = 1;
    add_surv(lead, (int32_t)(s_map[s] * stride + t), c, m);
  };
  // the parent edges' first four chunks (requested in phase 1)
  settle(e_s < S && e_need && s_i[e_s < S ? e_s : 0][QI_LIVE], e_left, e_wp, e_sh, e_word, e_s < S ? e_s : 0, 0, e_c);
  {
    // pairs of a task: the parent edge's chunks beyond four, every chunk of a kept edge
    const int tk_s = tid / QB_TASKS, tk_t = tid - tk_s * QB_TASKS;
    int nch = 0;
    if (tk_s < S && s_i[tk_s][QI_LIVE] && tk_t <= s_i[tk_s][QI_NNB]) {
      const int ns = s_NS[tk_s][tk_t];
      nch = ns > 0 ? (ns + 63) >> 6 : 0;
      if (tk_t == 0) nch = nch > 4 ? nch - 4 : 0;
    }
    int inc = nch;
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(inc, off);
      if (lane >= off) inc += o;
    }
    if (lane == 63) s_wsum[wv] = inc;
    __syncthreads();
    int before = 0, P = 0;
    for (int w = 0; w < 4; ++w) { if (w < wv) before += s_wsum[w]; P += s_wsum[w]; }
    const int first = before + inc - nch;
    for (int w0 = 0; w0 < P; w0 += PAIRCAP) {
      if (w0) __syncthreads();
      for (int c = 0; c < nch; ++c) {
        const int p = first + c - w0;
        if (p >= 0 && p < PAIRCAP) s_tab[p] = (tk_s << 24) | (tk_t << 16) | ((tk_t == 0 ? 4 : 0) + c);
      }
      __syncthreads();
      const int wn = P - w0 < PAIRCAP ? P - w0 : PAIRCAP;
      for (int e0 = 0; e0 < 8 * wn; e0 += 256) {
        const int e = e0 + tid;
        const bool valid = e < 8 * wn;
        const int ent = valid ? s_tab[e >> 3] : 0;
        const int s = ent >> 24, t = (ent >> 16) & 0xff, c = ent & 0xffff;
        bool need;
        int left, sh;
        const uint32_t* wp;
        group_addr(valid, s_T[s][t], s_NS[s][t], c, tid & 7, need, left, wp, sh);
        const uint32_t word = wp ? *wp : 0u;
        settle(need, left, wp, sh, word, s, t, c);
      }
    }
  }
  // the poses the bits leave open
  if (tid < S) {
    const bool open = s_i[tid][QI_LIVE] && need_pose && !((word_pose >> sh_pose) & 1u);
    if (open) {
      const int at = atomicAdd(&s_cnt[2], 1);
      const SurvivorItem it{-1 - s_map[tid], 0, 0ULL};
      if (at < SURVCAP) s_surv[at] = it;
      else {
        SurvivorItem* list = static_cast<SurvivorItem*>(A.items);
        const int sub_list = blockIdx.x & (SFFK_SUBLISTS - 1), sub_cap = A.items_cap / SFFK_SUBLISTS, half = sub_cap / 2;
        const int p = atomicAdd(A.sub + sub_list * SFFK_SUB_STRIDE, 1);
        if (p < half) list[(size_t)sub_list * sub_cap + p] = it;
        else A.ctrl[3] = 1;
      }
    }
  }
  __syncthreads();
  QB_MARK(6);
  // ---- 7. the workgroup's survivors -> the exact kernel's list (heavy items in the front half of the sub-list: see
  // k_query_classify); end points of the tasks that left a survivor
  {
    const int ns = s_cnt[2] < SURVCAP ? s_cnt[2] : SURVCAP;
    if (wv == 0 && ns > 0) {
      SurvivorItem* list = static_cast<SurvivorItem*>(A.items);
      const int sub_list = blockIdx.x & (SFFK_SUBLISTS - 1), sub_cap = A.items_cap / SFFK_SUBLISTS, half = sub_cap / 2;
      for (int b0 = 0; b0 < ns; b0 += 64) {
        const bool mineb = b0 + lane < ns;
        const SurvivorItem it = mineb ? s_surv[b0 + lane] : SurvivorItem{0, 0, 0ULL};
        const bool hv = mineb && (it.slot < 0 || __popcll(it.mask) >= QC_HEAVY);
        const unsigned long long hm = __ballot(hv), lm = __ballot(mineb && !hv), below = (1ULL << lane) - 1ULL;
        int bh = 0, bl = 0;
        if (lane == 0) {
          if (hm) bh = atomicAdd(A.sub + sub_list * SFFK_SUB_STRIDE, __popcll(hm));
          if (lm) bl = atomicAdd(A.sub + sub_list * SFFK_SUB_STRIDE + 2, __popcll(lm));
        }
        bh = __shfl(bh, 0); bl = __shfl(bl, 0);
        if (hv) {
          const int at = bh + __popcll(hm & below);
          if (at < half) list[(size_t)sub_list * sub_cap + at] = it;
          else A.ctrl[3] = 1;
        } else if (mineb) {
          const int at = bl + __popcll(lm & below);
          if (at < sub_cap - half) list[(size_t)sub_list * sub_cap + half + at] = it;
          else A.ctrl[3] = 1;
        }
      }
    }
    const int tk_s = tid / QB_TASKS, tk_t = tid - tk_s * QB_TASKS;
    if (tk_s < S && s_need[tk_s][tk_t]) {
      const int i = s_map[tk_s];
      const size_t slot = (size_t)i * stride + tk_t;
      double* sa = A.seg_a + 6 * slot;
      double* sb = A.seg_b + 6 * slot;
      if (tk_t == 0) { for (int k = 0; k < 6; ++k) { sa[k] = s_ex[tk_s][k]; sb[k] = s_qp[tk_s][k]; } }
      else {
        const int hj = s_rankhit[tk_s][tk_t];
        const int id = h_id[tk_s][hj];
        const bool same = h_tree[tk_s][hj] == s_i[tk_s][QI_MINE];
        const double* nbp = A.pos + 6 * (size_t)id;
        if (same) { for (int k = 0; k < 6; ++k) { sa[k] = nbp[k]; sb[k] = s_qp[tk_s][k]; } }
        else if (id == A.goal_id) { for (int k = 0; k < 6; ++k) { sa[k] = s_qp[tk_s][k]; sb[k] = nbp[k]; } }
        else { for (int k = 0; k < 6; ++k) { sa[k] = s_ex[tk_s][k]; sb[k] = nbp[k]; } }
      }
    }
  }
  QB_MARK(7);
#ifdef SFFK_DEBUG_COUNTERS
  QDBG(10, clock64() - qb_c0); QDBG(11, wall_clock64() - qb_r0);   // shader clock ticks over 100 MHz ticks: the clock the launch ran at
#endif
  if (clocked) {   // (thread 0 is through its workgroup's last phase: the others are at most a wavefront's tail behind)
    atomicMax(A.qclk_sh + (blockIdx.x & 63) * 16, wall_clock64());
  }
}

// Exact collision work of a round straight from the survivor list: persistent wavefronts, wave w takes items
// w, w + W, ... (about a thousand items over two thousand waves: one item per wave, no pooling needed).
// Housekeeping first: the round's own grid has been read by the query kernel, the cells it used are emptied here.
// List overflow (ctrl[3]): every live pose and every chunk of every live edge takes the exact test.
#ifndef CI_OCC
#define CI_OCC 2
#endif
// SHARE: many-candidate items are shared by the workgroup's wavefronts (share_help).  Two instantiations: the sharing code
// costs the kernel its last free registers (256 VGPRs + spills, + 7 % on dense_3D, whose items have 2-8 candidates), so
// small environments run the one without it (launch_collide_items).
template <bool SHARE>
__global__ __launch_bounds__(64 * SEG_WAVES) __attribute__((amdgpu_waves_per_eu(CI_OCC))) void k_collide_items(EnvView env, RobotView rob, const double* __restrict__ pos6,
                                                                  int n_pose, const int32_t* __restrict__ live_flags,
                                                                  uint8_t* __restrict__ pose_hit,
                                                                  const double* __restrict__ a6, const double* __restrict__ b6,
                                                                  const int32_t* __restrict__ seg_ns, int stride,
                                                                  int32_t* __restrict__ ctrl,
                                                                  const SurvivorItem* __restrict__ list, int items_cap,
                                                                  const int32_t* __restrict__ sub,
                                                                  int32_t* __restrict__ first_hit,
                                                                  int32_t* __restrict__ overflow_flag, GridView tg,
                                                                  const float* __restrict__ tx, const float* __restrict__ ty,
                                                                  const float* __restrict__ tz, int n_temps,
                                                                  const int32_t* __restrict__ dev_n, TaskSource D) {
#ifdef SFFK_CI_TRACE
  const unsigned long long ci_entry = wall_clock64();
#endif
  if (dev_n) {
    if (dev_n[1]) return;
    n_pose = dev_n[0];
    if (n_temps) n_temps = dev_n[0];
  }
  extern __shared__ double lds_d[];
  double* rtri = lds_d;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // Every load of the preamble is issued before anything waits (round 5: the trace of a launch showed its first item
  // starting 4.5 us in - four memory round trips one after the other: the round grid's cleanup, the sub-lists' counters,
  // the overflow flag, the robot): the sub-lists' fill counts, one per lane (SFFK_SUBLISTS == 64) ...
  int sub_n = sub[lane * SFFK_SUB_STRIDE], sub_nl = sub[lane * SFFK_SUB_STRIDE + 2];
  const int ctrl3 = ctrl[3];
  // ... the robot's triangles (up to CI_PRE doubles per thread in registers, the rest - large robots - by the loop below) ...
  constexpr int CI_PRE = 6;
  const int n9 = rob.n_tri * 9;
  double rv[CI_PRE];
#pragma unroll
  for (int k = 0; k < CI_PRE; ++k) {
    const int idx = (int)threadIdx.x + k * 256;
    rv[k] = idx < n9 ? rob.tri[idx] : 0.0;
  }
  // ... and the first of this thread's temporaries of the round grid that has to be emptied
  const int gt0 = blockIdx.x * blockDim.x + threadIdx.x;
  float cx0 = __int_as_float(0x7fc00000), cy0 = 0.0f, cz0 = 0.0f;
  if (tg.cnt && gt0 < n_temps) { cx0 = tx[gt0]; cy0 = ty[gt0]; cz0 = tz[gt0]; }
#pragma unroll
  for (int k = 0; k < CI_PRE; ++k) {
    const int idx = (int)threadIdx.x + k * 256;
    if (idx < n9) rtri[idx] = rv[k];
  }
  for (int i = (int)threadIdx.x + CI_PRE * 256; i < n9; i += blockDim.x) rtri[i] = rob.tri[i];
  if (tg.cnt) {
    if (cx0 == cx0) {
      const size_t cell = grid_cell_of(tg, cx0, cy0, cz0);
      tg.cnt[cell] = 0;
      if (tg.occ) tg.occ[cell >> 5] = 0u;   // (every set bit of the word belongs to a sample of this round)
    }
    for (int t = gt0 + gridDim.x * blockDim.x; t < n_temps; t += gridDim.x * blockDim.x) {
      const float x = tx[t];
      if (x == x) {
        const size_t cell = grid_cell_of(tg, x, ty[t], tz[t]);
        tg.cnt[cell] = 0;
        if (tg.occ) tg.occ[cell >> 5] = 0u;
      }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) tg.ovf_cnt[0] = 0;
  }
  double* stage = rtri + (size_t)rob.n_tri * 9 + (size_t)wave * STAGE_DOUBLES;
  int32_t* ibase = reinterpret_cast<int32_t*>(rtri + (size_t)rob.n_tri * 9 + (size_t)SEG_WAVES * STAGE_DOUBLES);
  const int sub_cap = items_cap / SFFK_SUBLISTS, sub_half = sub_cap / 2;
  // (every sub-list: heavy items in its front half, light ones in its back half, one counter each)
  sub_n = sub_n < sub_half ? sub_n : sub_half;
  sub_nl = sub_nl < sub_cap - sub_half ? sub_nl : sub_cap - sub_half;
  int sub_incl = sub_n, sub_incl_l = sub_nl;
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(sub_incl, off), ol = __shfl_up(sub_incl_l, off);
    if (lane >= off) { sub_incl += o; sub_incl_l += ol; }
  }
  const int MH = __shfl(sub_incl, 63);
  const int M = MH + __shfl(sub_incl_l, 63);
  const bool ran_over = ctrl3 != 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) ctrl[2] = M;   // (statistics)
  if ((M <= 0 && !ran_over) || env.n_tri == 0) return;   // (uniform over the launch: no barrier is left waiting)
  __shared__ ShareArea s_share;
  share_init(s_share);
  __syncthreads();
  double* rbox = reinterpret_cast<double*>(ibase + SEG_WAVES * (STACK_CAP + TG_HASH + CAND_CAP + QUEUE_CAP));
  fill_robot_boxes(rtri, rbox, rob.n_tri, threadIdx.x, blockDim.x);
  __syncthreads();
  int32_t* stack = ibase + wave * (STACK_CAP + TG_HASH);
  int32_t* cand = ibase + SEG_WAVES * (STACK_CAP + TG_HASH) + wave * CAND_CAP;
  int32_t* queue = ibase + SEG_WAVES * (STACK_CAP + TG_HASH + CAND_CAP) + wave * QUEUE_CAP;
  DBG_DECL
  const int W = gridDim.x * SEG_WAVES;
#ifdef SFFK_CI_TRACE
  const unsigned long long ci_t0 = ci_entry, ci_pre = wall_clock64();
  // (block 0 counts the launch at its END: launches are serial, so every block of launch N reads N)
  const unsigned int ci_seen = __hip_atomic_load(&g_ci_launch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const bool ci_on = ci_seen == SFFK_CI_TRACE;
  int ci_items = 0;
  unsigned long long ci_first[4] = {0, 0, 0, 0};
#endif
  // one loop for both sources of work (the exact test is inlined once): the list's items, or - list overflow -
  // every live pose followed by every live edge with all its chunks
  const int n_slots = n_pose * stride;
  const int E = ran_over ? n_pose + n_slots : M;
  // Every wave takes one item of its own (neighbouring items go to different CUs); the items beyond one per wave are
  // handed out as the waves come back for more - 64 groups of waves, one ticket counter each (a single counter would see
  // every wave's last, failing draw: returning atomics on one word retire at ~90 per us).  A pose or a chunk near the
  // obstacles takes 25 us, the median item 7: with a fixed second item the kernel's length was "a long item + another".
  int32_t* const ticket = const_cast<int32_t*>(sub) + (size_t)((blockIdx.x + gridDim.x * wave) & (SFFK_SUBLISTS - 1)) * SFFK_SUB_STRIDE + 1;
  auto next_item = [&](int e) -> int {
    if (ran_over) return e + W;
    int t = 0;
    if (lane == 0) t = atomicAdd(ticket, 1);
    t = __builtin_amdgcn_readfirstlane(t);
    return W + ((blockIdx.x + gridDim.x * wave) & (SFFK_SUBLISTS - 1)) + SFFK_SUBLISTS * t;
  };
  for (int e = blockIdx.x + gridDim.x * wave; e < E; e = next_item(e)) {
    int slot, c_lo, c_hi;
    unsigned long long mask = 0ULL;
    if (!ran_over) {
      size_t at;                                          // the heavy items first, then the light ones
      if (e < MH) {
        const int sl = __popcll(__ballot(sub_incl <= e));   // the sub-list item e lies in
        at = (size_t)sl * sub_cap + (e - (__shfl(sub_incl, sl) - __shfl(sub_n, sl)));
      } else {
        const int e2 = e - MH;
        const int sl = __popcll(__ballot(sub_incl_l <= e2));
        at = (size_t)sl * sub_cap + sub_half + (e2 - (__shfl(sub_incl_l, sl) - __shfl(sub_nl, sl)));
      }
      const SurvivorItem it = list[at];
      slot = it.slot; c_lo = it.chunk; c_hi = it.chunk + 1; mask = it.mask;
    } else if (e < n_pose) {
      if ((live_flags[e] & 3) != 1) continue;
      slot = -1 - e; c_lo = c_hi = 0;
    } else {
      slot = e - n_pose;
      if (D.on) {   // (k_query_block does not clear the task slots a sample leaves unused)
        const int si = slot / stride;
        if ((live_flags[si] & 3) != 1 || slot - si * stride > D.rec_nnb[si]) continue;
      }
      const int ns = seg_ns[slot];
      if (ns <= 0) continue;
      c_lo = 0; c_hi = (ns + 63) >> 6;
    }
    if (slot < 0) {
      const int pose = -1 - slot;
      double p[6], R[9], c[3];
#ifdef SFFK_CI_TRACE
      const unsigned long long ci_p0 = wall_clock64();
#endif
      pose_frame(rob, pos6, pose, p, R, c);
      const bool hit = pose_exact(env, rob, rtri, stack, cand, stage, p, R, c, lane, (SHARE && !ran_over) ? &s_share : nullptr);
      if (lane == 0) pose_hit[pose] = hit ? 1 : 0;
#ifdef SFFK_CI_TRACE
      ci_first[3] += ((wall_clock64() - ci_p0) << 8) | 1ULL;   // (time << 8 | count)
#endif
      continue;
    }
    double a[6], b[6];
    if (D.on && ran_over) {
      // k_query_block wrote the end points of the tasks that left a survivor only: with the list run over, every task's
      // end points come from the sample's records (the rules of src/forest.h:246,276,287,288)
      const int si = slot / stride, t = slot - si * stride;
      const double* pe = D.center ? D.center + 6 * (size_t)si : D.pos + 6 * (size_t)D.parent[si];
      const double* pq = pos6 + 6 * (size_t)si;
      if (t == 0) { for (int k = 0; k < 6; ++k) { a[k] = pe[k]; b[k] = pq[k]; } }
      else {
        const int id = D.rec_nb[(size_t)si * D.nbcap + t - 1];
        const bool same = (D.rec_meta[(size_t)si * D.nbcap + t - 1] & 1) != 0;
        const double* pn = D.pos + 6 * (size_t)id;
        if (same) { for (int k = 0; k < 6; ++k) { a[k] = pn[k]; b[k] = pq[k]; } }
        else if (id == D.goal_id) { for (int k = 0; k < 6; ++k) { a[k] = pq[k]; b[k] = pn[k]; } }
        else { for (int k = 0; k < 6; ++k) { a[k] = pe[k]; b[k] = pn[k]; } }
      }
    } else {
      for (int k = 0; k < 6; ++k) { a[k] = a6[6 * (size_t)slot + k]; b[k] = b6[6 * (size_t)slot + k]; }
    }
    for (int chunk = c_lo; chunk < c_hi; ++chunk) {
      if (chunk > 0 && first_hit[slot] <= 64 * chunk) break;
#ifdef SFFK_CI_TRACE
      const unsigned long long ci_a = wall_clock64();
#endif
      segment_chunk(env, rob, rtri, rbox, stack, cand, queue, stage, a, b, slot, chunk, !ran_over, mask, first_hit, overflow_flag, lane DBG_PASS,
                            (SHARE && !ran_over) ? &s_share : nullptr);
#ifdef SFFK_CI_TRACE
      if (ci_items == 0) { ci_first[0] = ci_a; ci_first[1] = wall_clock64(); ci_first[2] = (unsigned long long)__popcll(mask); }
      ++ci_items;
#endif
    }
  }
  // own items done: blocks of the siblings' many-candidate items (share_help), until the whole workgroup is through
  if (SHARE) share_help(env, rob, rtri, rbox, queue, stage, ibase + SEG_WAVES * (STACK_CAP + TG_HASH), s_share, lane DBG_PASS);
#ifdef SFFK_CI_TRACE
  if (ci_on && lane == 0) {
    const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    unsigned long long* o = g_ci_trace + 8 * (size_t)gw;
    o[0] = ci_t0; o[1] = ci_first[0]; o[2] = ci_items ? g_ci_tmp[2 * gw] : 0ULL; o[3] = ci_first[1];
    o[4] = ci_items ? g_ci_tmp[2 * gw + 1] : 0ULL; o[5] = ci_first[2] | (ci_pre << 8); o[6] = (unsigned long long)ci_items | (ci_first[3] << 8); o[7] = wall_clock64();
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&g_ci_launch, 1u);
#endif
  DBG_FLUSH();
}

// SFF* on the device (devstar.hip): the member-edge chunks a star pass could not answer from the clearance bits.  Same
// persistent scheme as k_collide_items; an edge's end points are store entries (the new sample's temporary entry and
// the member), named per edge slot by ida / idb.
// star_exact_items: the item loop, by workgroup `wg` of `nwg` (k_star_exact: the grid; k_star_tail: its active workgroups).
// sub = the pass's SFFK_SUBLISTS counters + tickets; sub_n / sub_incl = lane's sub-list length and their running sum.
struct ExactLds {
  double* rtri; double* rbox; double* stage; int32_t* stack; int32_t* cand; int32_t* queue; int32_t* cand_base;
};
__device__ __forceinline__ ExactLds exact_lds(double* lds_d, int n_rob_tri, int wave) {
  ExactLds E;
  E.rtri = lds_d;
  E.stage = E.rtri + (size_t)n_rob_tri * 9 + (size_t)wave * STAGE_DOUBLES;
  int32_t* ibase = reinterpret_cast<int32_t*>(E.rtri + (size_t)n_rob_tri * 9 + (size_t)SEG_WAVES * STAGE_DOUBLES);
  E.rbox = reinterpret_cast<double*>(ibase + SEG_WAVES * (STACK_CAP + TG_HASH + CAND_CAP + QUEUE_CAP));
  E.stack = ibase + wave * (STACK_CAP + TG_HASH);
  E.cand_base = ibase + SEG_WAVES * (STACK_CAP + TG_HASH);
  E.cand = E.cand_base + wave * CAND_CAP;
  E.queue = ibase + SEG_WAVES * (STACK_CAP + TG_HASH + CAND_CAP) + wave * QUEUE_CAP;
  return E;
}
template <bool COH>
__device__ __forceinline__ void star_exact_items(const EnvView& env, const RobotView& rob, const ExactLds& E,
                                                 const double* __restrict__ store_pos, const int32_t* ida,
                                                 const int32_t* idb, const SurvivorItem* list,
                                                 int sub_cap, int32_t* sub, int sub_n, int sub_incl, int M,
                                                 int32_t* first_hit, int32_t* overflow_flag,
                                                 ShareArea& s_share, int wg, int nwg, int wave, int lane) {
  DBG_DECL
  const int W = nwg * SEG_WAVES;
  // (one item per wave, the rest by tickets to the waves that come back first: see k_collide_items).  The items behind the
  // first W are dealt over SFFK_SUBLISTS ticket groups, a wave draws from the group of its first item - which only reaches
  // every group when there are at least that many waves: a smaller launch (k_star_tail bounded to a few workgroups, a
  // device with few CUs) draws all of them from ONE ticket word instead (round 5 left the items of the groups without a
  // wave untested: their first_hit stayed "free").
  const bool few = W < SFFK_SUBLISTS;
  const int grp = few ? 0 : ((wg + nwg * wave) & (SFFK_SUBLISTS - 1));
  int32_t* const ticket = sub + (size_t)grp * SFFK_STAR_SUB + 1;
  auto next_item = [&]() -> int {
    int t = 0;
    if (lane == 0) t = atomicAdd(ticket, 1);
    t = __builtin_amdgcn_readfirstlane(t);
    return few ? W + t : W + grp + SFFK_SUBLISTS * t;
  };
  for (int e = wg + nwg * wave; e < M; e = next_item()) {
    const int sl = __popcll(__ballot(sub_incl <= e));
    const int j = e - (__shfl(sub_incl, sl) - __shfl(sub_n, sl));
    const SurvivorItem it = sld<COH>(list + (size_t)sl * sub_cap + j);
    const int slot = it.slot;
    if (it.chunk > 0 && sld<COH>(first_hit + slot) <= 64 * it.chunk) continue;
    double a[6], b[6];
    const double* pa = store_pos + 6 * (size_t)sld<COH>(ida + slot);
    const double* pb = store_pos + 6 * (size_t)sld<COH>(idb + slot);
    for (int k = 0; k < 6; ++k) { a[k] = pa[k]; b[k] = pb[k]; }
    segment_chunk(env, rob, E.rtri, E.rbox, E.stack, E.cand, E.queue, E.stage, a, b, slot, it.chunk, true, it.mask, first_hit,
                  overflow_flag, lane DBG_PASS, &s_share);
  }
  share_help(env, rob, E.rtri, E.rbox, E.queue, E.stage, E.cand_base, s_share, lane DBG_PASS);
}

__global__ __launch_bounds__(64 * SEG_WAVES) __attribute__((amdgpu_waves_per_eu(CI_OCC))) void k_star_exact(
    EnvView env, RobotView rob, const double* __restrict__ store_pos, const int32_t* __restrict__ ida,
    const int32_t* __restrict__ idb, const SurvivorItem* __restrict__ list, int items_cap, int32_t* __restrict__ sub,
    int32_t* __restrict__ first_hit, int32_t* __restrict__ overflow_flag, int32_t* __restrict__ hdr) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // (the sub-list counts are asked for before the header words are tested: an idle launch is one trip to memory)
  int sub_n = sub[lane * SFFK_STAR_SUB];
  const int h_skip = hdr[1], h_fault = hdr[4];
  if (h_skip || h_fault) return;
  extern __shared__ double lds_d[];
  const ExactLds E = exact_lds(lds_d, rob.n_tri, wave);
  const int sub_cap = items_cap / SFFK_SUBLISTS;
  if (sub_n > sub_cap) {   // a sub-list ran over: items were dropped - the round is redone on the host path
    if (blockIdx.x == 0 && wave == 0) atomicOr(hdr + 4, 1);
    sub_n = sub_cap;
  }
  int sub_incl = sub_n;
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(sub_incl, off);
    if (lane >= off) sub_incl += o;
  }
  const int M = __shfl(sub_incl, 63);
  if (M <= 0 || env.n_tri == 0) return;
  __shared__ ShareArea s_share;
  share_init(s_share);
  for (int i = threadIdx.x; i < rob.n_tri * 9; i += blockDim.x) E.rtri[i] = rob.tri[i];
  __syncthreads();
  fill_robot_boxes(E.rtri, E.rbox, rob.n_tri, threadIdx.x, blockDim.x);
  __syncthreads();
  star_exact_items<false>(env, rob, E, store_pos, ida, idb, list, sub_cap, sub, sub_n, sub_incl, M, first_hit, overflow_flag, s_share,
                          blockIdx.x, gridDim.x, wave, lane);
}

// ------------------------------------------------------------------ SFF*: the passes after the first as ONE launch
// A round's fixed point takes 3.2 passes on average (configs[4]) and up to 6; as launches - 8 passes + 7 exact kernels per
// round, sized for the worst round - the ones with nothing to do cost ~3 us each and every working one a launch boundary:
// 65 of a round's 280 us.  k_star_tail runs "pass p, the exact items it left, pass p + 1, ..." until a pass changes nothing,
// in the first `g_act` workgroups of its grid (one per four accepted samples; the others return at once) with a barrier
// over them between the phases: an agent-scope release / acquire pair around one counter (what a kernel boundary does to
// the caches, without the boundary).  The grid is small enough to be resident at once (<= one workgroup per CU, <= 64 when
// several processes may share the GPU - the launcher's choice); a barrier that is not complete within STAR_BAR_TICKS
// raises the stage's fault flag instead of waiting for ever (the round is then redone on the host path like any fault).
// Pass p uses the counter set / changed flag p mod SFFK_STAR_PASSES; workgroup 0 clears the set of pass p + 1 during pass p.
#define STAR_BAR_TICKS 4000000ULL      // 40 ms of the 100 MHz wall clock
// what the workgroup decides behind a barrier (fault raised? pass changed something?) is read ONCE, by thread 0: a flag another
// workgroup raises just then must not send some of its wavefronts out of the kernel and others into a phase that waits for them
__device__ __forceinline__ void star_decide(int32_t* s_dec, const int32_t* fault, const int32_t* chg) {
  __syncthreads();
  if (threadIdx.x == 0) {
    s_dec[0] = __hip_atomic_load(fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_dec[1] = chg ? __hip_atomic_load(chg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
  }
  __syncthreads();
}
__device__ __forceinline__ void star_grid_barrier(int32_t* bar, int target, int32_t* fault, unsigned long long* dbg) {
  // (what the workgroup hands over it wrote through - sst<true> - and every such store has been acknowledged before the
  //  workgroup counts itself in; what it takes over behind the barrier it reads from memory - sld<true>: no cache-wide
  //  write-back / invalidate, the fixed data of the round stays in the L2 from pass to pass)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long ta = wall_clock64();
    __hip_atomic_fetch_add(bar, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t0 > STAR_BAR_TICKS) { atomicOr(fault, 1); break; }
    }
    const unsigned long long t1 = wall_clock64();
    if (dbg) { const unsigned long long t2 = wall_clock64(); dbg[0] += t0 - ta; dbg[1] += t1 - t0; dbg[2] += t2 - t1; }
  }
  __syncthreads();
}

__global__ __launch_bounds__(64 * SEG_WAVES) __attribute__((amdgpu_waves_per_eu(CI_OCC))) void k_star_tail(
    ResolveArgs A, EnvView env, RobotView rob, NodeStoreView st, int max_passes, int test_stall) {
  __shared__ StarPassLds L;
  __shared__ ShareArea s_share;
  __shared__ int32_t s_dec[2];
  extern __shared__ double lds_d[];
  const DevForestView& f = A.f;
  const StarView& S = A.S;
  const DevCtrl* c = f.ctrl;
  const int app_n = c->app_n, N0 = c->app_N0;
  const unsigned ep = (unsigned)c->epoch;
  const int h_n = S.hdr[0], h_skip = S.hdr[1], n_ev = S.hdr[2], ev_first = S.hdr[3], h_fault = S.hdr[STAR_FAULT];
  const int chg0 = S.changed[0];
  if (app_n <= 0 || h_skip || h_fault) return;
  const int wg = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (chg0 == 0 || max_passes <= 1) {            // the first pass wrote nothing (or is all the caller allows)
    if (wg == 0 && threadIdx.x == 0) { S.hdr[STAR_PASSES_RUN] = 1; S.hdr[STAR_CONVERGED] = chg0 == 0; }
    return;
  }
  int g_act = (h_n + 3) >> 2;
  g_act = g_act < 16 ? 16 : g_act;
  g_act = g_act > (int)gridDim.x ? (int)gridDim.x : g_act;
  if (wg >= g_act) return;
  // (tests: a workgroup that never arrives - the others' barrier must time out into the stage's fault flag, the round go to the host)
  if (test_stall > 0 && g_act > 1 && wg == g_act - 1 && ep % (unsigned)test_stall == 0u) return;
  const int Tb = f.temp_base;
  const ExactLds E = exact_lds(lds_d, rob.n_tri, wave);
  const int sub_cap = S.items_cap / SFFK_SUBLISTS;
  int32_t* const bar = S.changed + SFFK_STAR_BAR;
  int32_t* const fault = S.hdr + STAR_FAULT;
  bool robot_ready = false, conv = false;
  int n_bar = 0, pass = 1;
  // SFFGPU_PROFILE, workgroup 0: [16] launches that ran passes, [17] passes, ticks of [18] pass phases, [19] exact phases,
  // [20..22] barriers: release, wait, acquire
  unsigned long long* const dbg = (S.dbg && wg == 0) ? S.dbg + 16 : nullptr;
  unsigned long long tp = dbg ? wall_clock64() : 0ULL;
  if (dbg && threadIdx.x == 0) dbg[0] += 1;
  while (true) {
    const int slot = pass & (SFFK_STAR_PASSES - 1);
    // ---- pass `pass`
    if (wg == 0) {
      const int nx = (pass + 1) & (SFFK_STAR_PASSES - 1);
      for (int t = threadIdx.x; t < SFFK_SUBLISTS * SFFK_STAR_SUB; t += blockDim.x) sst<true>(S.sub + (size_t)nx * SFFK_SUBLISTS * SFFK_STAR_SUB + t, 0);
      if (threadIdx.x == 0) sst<true>(S.changed + nx, 0);
    }
    for (int r = wg * 4 + wave; r < h_n; r += g_act * 4)
      star_pass_sample<true>(A, env, st, slot, (r >> 2) & (SFFK_SUBLISTS - 1), S.acc_sample[r], N0, Tb, ep, L, wave, lane);
    if (dbg && threadIdx.x == 0) { const unsigned long long t = wall_clock64(); dbg[1] += 1; dbg[2] += t - tp; }
    star_grid_barrier(bar, ++n_bar * g_act, fault, dbg ? dbg + 4 : nullptr);
    if (dbg) tp = wall_clock64();
    star_decide(s_dec, fault, S.changed + slot);
    const int flt = s_dec[0], chg = s_dec[1];
    if (flt) break;
    if (chg == 0) { conv = true; break; }
    if (pass + 1 >= max_passes) break;
    // ---- the edges it sent to the exact test
    int32_t* sub = S.sub + (size_t)slot * SFFK_SUBLISTS * SFFK_STAR_SUB;
    int sub_n = __hip_atomic_load(sub + lane * SFFK_STAR_SUB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool ran_over = __any(sub_n > sub_cap);   // a sub-list ran over: items were dropped - the round is redone on the host path
    if (ran_over) {                                  // (every workgroup reads the same counts: they all leave here)
      if (wg == 0 && threadIdx.x == 0) atomicOr(fault, 1);
      break;
    }
    int sub_incl = sub_n;
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(sub_incl, off);
      if (lane >= off) sub_incl += o;
    }
    const int M = __shfl(sub_incl, 63);
    if (M > 0 && env.n_tri != 0) {
      if (!robot_ready) {
        for (int i = threadIdx.x; i < rob.n_tri * 9; i += blockDim.x) E.rtri[i] = rob.tri[i];
        __syncthreads();
        fill_robot_boxes(E.rtri, E.rbox, rob.n_tri, threadIdx.x, blockDim.x);
        robot_ready = true;
      }
      share_init(s_share);
      __syncthreads();
      star_exact_items<true>(env, rob, E, st.pos, S.ida, S.idb, static_cast<const SurvivorItem*>(S.items), sub_cap, sub, sub_n, sub_incl,
                       M, S.first_hit, S.seg_ovf, s_share, wg, g_act, wave, lane);
    }
    // (no items: every workgroup read the same counts - nothing to wait for before the next pass)
    if (M > 0) {
      if (dbg && threadIdx.x == 0) dbg[3] += wall_clock64() - tp;
      star_grid_barrier(bar, ++n_bar * g_act, fault, dbg ? dbg + 4 : nullptr);
      if (dbg) tp = wall_clock64();
      star_decide(s_dec, fault, nullptr);
      if (s_dec[0]) break;
    }
    ++pass;
  }
  // the border entries' costs, once, on the costs the fixed point ended with (all written before the last barrier)
  if (conv)
    for (int e = wg * 256 + (int)threadIdx.x; e < n_ev; e += g_act * 256) star_pass_event<true>(f, S, e, ev_first, N0, ep);
  if (wg == 0 && threadIdx.x == 0) { S.hdr[STAR_PASSES_RUN] = pass + 1; S.hdr[STAR_CONVERGED] = conv ? 1 : 0; }
}

// Samples whose fate needs no in-order replay (src/forest.h:246-299): rejected by their own pose or parent-edge
// check, or by a STORE neighbour when no sample of this round appears anywhere in their neighbour list, and
// without side effect (no border entry).  code: 0 = replay on the host, 1 = settled, 2 = outside the limits.
// The reference-equivalent counters of the settled samples are summed here: bulk[0] Collide calls,
// [1] isPathFree calls, [2] radius queries, [3] settled samples.
__global__ __launch_bounds__(256) void k_settle(SettleArgs A) {
  if (A.dev_n) {
    if (A.dev_n[1]) return;
    A.n = A.dev_n[0];
  }
  const int i = blockIdx.x * 256 + threadIdx.x;
  unsigned long long cc = 0, pf = 0, nq = 0, ns_settled = 0;
  unsigned long long ex_pose = 0, ex_seg = 0, ex_smp = 0;
  if (i < A.n) {
    int code = 0;
    if (!A.in_lim[i]) code = 2;
    else if (A.fault && (A.rec_flags[i] & 2)) atomicOr(A.fault, 1);   // hit / neighbour list overflow: host path
    else if ((A.rec_flags[i] & 3) == 1) {       // owned by this rank and fully answered on the device
      auto calls = [](int fh, int ns) -> unsigned long long {
        return fh != 0x7fffffff ? (unsigned long long)fh : (unsigned long long)ns;
      };
      const size_t s0 = (size_t)i * A.stride;
      const int nnb = A.rec_nnb[i];
      bool ovf = A.first_hit[s0] == 0;            // 0 = the edge's triangle candidate list ran over
      for (int k = 0; k < nnb; ++k) ovf |= A.first_hit[s0 + 1 + k] == 0;
      if (ovf && A.fault) atomicOr(A.fault, 1);
      if (A.count_executed) {
        ex_pose = 1;
        ex_seg = 1 + (unsigned long long)nnb;
        ex_smp = (unsigned long long)A.seg_ns[s0];
        for (int k = 0; k < nnb; ++k) ex_smp += (unsigned long long)A.seg_ns[s0 + 1 + k];
      }
      if (!ovf) {
        unsigned long long c1 = 1, p1 = 0, q1 = 0;   // :246 env.Collide(newPoint)
        bool settled = false;
        if (A.pose_hit[i]) settled = true;
        else {
          p1 += 1;
          c1 += calls(A.first_hit[s0], A.seg_ns[s0]);
          if (A.first_hit[s0] != 0x7fffffff) settled = true;   // parent edge blocked
          else {
            bool mates = false;
            for (int k = 0; k < nnb; ++k) mates |= A.rec_nb[(size_t)i * A.nbcap + k] >= A.Tb;
            if (!mates) {
              q1 += (unsigned long long)A.n_trees;
              for (int k = 0; k < nnb; ++k) {
                const bool fr = A.first_hit[s0 + 1 + k] == 0x7fffffff;
                p1 += 1;
                c1 += calls(A.first_hit[s0 + 1 + k], A.seg_ns[s0 + 1 + k]);
                if (A.rec_meta[(size_t)i * A.nbcap + k] & 1) {
                  if (fr) { settled = true; break; }            // :276-280 overcrowded
                } else {
                  if (!fr) settled = true;                      // :296-299 without a border entry
                  break;                                        // (a free edge records a border: replay)
                }
              }
            }
          }
        }
        if (settled) { code = 1; cc = c1; pf = p1; nq = q1; ns_settled = 1; }
      }
    }
    A.code[i] = (uint8_t)code;
  }
  for (int off = 32; off > 0; off >>= 1) {
    cc += __shfl_xor(cc, off); pf += __shfl_xor(pf, off); nq += __shfl_xor(nq, off); ns_settled += __shfl_xor(ns_settled, off);
  }
  if ((threadIdx.x & 63) == 0 && ns_settled) {
    atomicAdd(A.bulk + 0, cc); atomicAdd(A.bulk + 1, pf); atomicAdd(A.bulk + 2, nq); atomicAdd(A.bulk + 3, ns_settled);
  }
  if (A.count_executed) {
    for (int off = 32; off > 0; off >>= 1) {
      ex_pose += __shfl_xor(ex_pose, off); ex_seg += __shfl_xor(ex_seg, off); ex_smp += __shfl_xor(ex_smp, off);
    }
    if ((threadIdx.x & 63) == 0 && ex_pose) {
      atomicAdd(A.bulk + 4, ex_pose); atomicAdd(A.bulk + 5, ex_seg); atomicAdd(A.bulk + 6, ex_smp);
    }
  }
}

// ------------------------------------------------------------------ waves of one slot: the reference's loop, one wavefront
// Loads of forest state this launch may itself have written go past the vector L1 (agent-scope relaxed atomic loads =
// `sc1` loads served by the L2 the one workgroup's stores write through to); the environment, the robot and the engine
// words are immutable while it runs.
__device__ __forceinline__ int sq_i32(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long sq_u64(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double sq_f64(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                            __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ int sq_u8(const uint8_t* p) { return (int)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void sq_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ int sq_lemire(unsigned long long word, unsigned long long range) {   // (devforest.hip: lemire_pick)
  const unsigned long long lo = word * range;
  const unsigned long long hi = __umul64hi(word, range);
  if (lo < range) {
    const unsigned long long thr = (0ULL - range) % range;
    if (lo < thr) return -1;
  }
  return (int)hi;
}

// isPathFree's common case INLINE: every sample of the edge lies in a cell the clearance bits (edge plane) call clear - the
// same placement arithmetic as sq_path_free's first phase, so the same bits are asked; edges of up to 512 samples.  True =
// free, the reference's counters advanced; false = undecided, nothing touched: the caller runs sq_path_free (out of line:
// its call saves and restores a good part of a wavefront's registers - 1.5-2 us on the wavefront everybody waits for).
__device__ __forceinline__ bool sq_edge_clear_fast(const EnvView& env, const double* a, const double* b, int lane,
                                                   unsigned long long& calls, unsigned long long& samples) {
  if (env.n_tri == 0 || !env.clear_bits_edge) return false;
  const double parts = edge_parts(a, b);
  const int ns = edge_samples(parts);
  if (ns > 512) return false;
  const float inv = (float)env.clear_inv * __frcp_rn((float)parts);
  const float g0 = (float)((a[0] - env.clear_org[0]) * env.clear_inv), g1 = (float)((a[1] - env.clear_org[1]) * env.clear_inv),
              g2 = (float)((a[2] - env.clear_org[2]) * env.clear_inv);
  const float d0 = (float)(b[0] - a[0]) * inv, d1 = (float)(b[1] - a[1]) * inv, d2 = (float)(b[2] - a[2]) * inv;
  const float nxd = (float)env.clear_n[0], nyd = (float)env.clear_n[1], nzd = (float)env.clear_n[2];
  const uint32_t* wp[8];
  int sh[8];
  bool need[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int idx = 1 + 64 * u + lane;
    need[u] = idx <= ns;
    wp[u] = nullptr; sh[u] = 0;
    if (need[u]) {
      const float td = (float)idx;
      const float fx = __builtin_fmaf(td, d0, g0), fy = __builtin_fmaf(td, d1, g1), fz = __builtin_fmaf(td, d2, g2);
      if (fx >= 0 && fy >= 0 && fz >= 0 && fx < nxd && fy < nyd && fz < nzd) {
        const uint32_t ci = ((uint32_t)(int)fz * (uint32_t)env.clear_n[1] + (uint32_t)(int)fy) * (uint32_t)env.clear_n[0] + (uint32_t)(int)fx;
        wp[u] = env.clear_bits_edge + (ci >> 5);
        sh[u] = (int)(ci & 31u);
      } else if (fx == fx && fy == fy && fz == fz) {
        need[u] = false;                                  // beyond the inflated box of the environment
      }
    }
  }
  uint32_t word[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) word[u] = wp[u] ? *wp[u] : 0u;
  bool open = false;
#pragma unroll
  for (int u = 0; u < 8; ++u) open = open || (need[u] && !(wp[u] && ((word[u] >> sh[u]) & 1u)));
  if (__any(open)) return false;
  calls += (unsigned long long)ns;
  samples += (unsigned long long)ns;
  return true;
}

// sq_edge_clear_fast's question for up to three edges at once, the clearance words of all of them in flight together (SFF*'s
// choose-parent / rewire candidates in k_spec_waves: a wavefront alone on its SIMD pays a full memory round trip per edge
// otherwise).  clear[e]: the bits alone say "free" (an edge of more than 256 samples answers false: the caller asks
// sq_edge_clear_fast / sq_path_free); ns[e]: its sample count = the Collide calls the reference makes on a free edge.
// Counts nothing: the caller does, for the edges whose turn really comes.
__device__ __forceinline__ void sq_edges_clear_probe(const EnvView& env, const double (*a)[6], const double (*b)[6], int n, int lane,
                                                     bool* clear, int* ns_out) {
  const uint32_t* wp[3][4];
  int sh[3][4];
  bool need[3][4];
  bool can[3];
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    can[e] = false;
    ns_out[e] = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) { wp[e][u] = nullptr; sh[e][u] = 0; need[e][u] = false; }
    if (e >= n || env.n_tri == 0 || !env.clear_bits_edge) continue;
    const double parts = edge_parts(a[e], b[e]);
    const int ns = edge_samples(parts);
    ns_out[e] = ns;
    if (ns > 256) continue;
    can[e] = true;
    const float inv = (float)env.clear_inv * __frcp_rn((float)parts);
    const float g0 = (float)((a[e][0] - env.clear_org[0]) * env.clear_inv), g1 = (float)((a[e][1] - env.clear_org[1]) * env.clear_inv),
                g2 = (float)((a[e][2] - env.clear_org[2]) * env.clear_inv);
    const float d0 = (float)(b[e][0] - a[e][0]) * inv, d1 = (float)(b[e][1] - a[e][1]) * inv, d2 = (float)(b[e][2] - a[e][2]) * inv;
    const float nxd = (float)env.clear_n[0], nyd = (float)env.clear_n[1], nzd = (float)env.clear_n[2];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = 1 + 64 * u + lane;
      need[e][u] = idx <= ns;
      if (need[e][u]) {
        const float td = (float)idx;
        const float fx = __builtin_fmaf(td, d0, g0), fy = __builtin_fmaf(td, d1, g1), fz = __builtin_fmaf(td, d2, g2);
        if (fx >= 0 && fy >= 0 && fz >= 0 && fx < nxd && fy < nyd && fz < nzd) {
          const uint32_t ci = ((uint32_t)(int)fz * (uint32_t)env.clear_n[1] + (uint32_t)(int)fy) * (uint32_t)env.clear_n[0] + (uint32_t)(int)fx;
          wp[e][u] = env.clear_bits_edge + (ci >> 5);
          sh[e][u] = (int)(ci & 31u);
        } else if (fx == fx && fy == fy && fz == fz) {
          need[e][u] = false;                               // beyond the inflated box of the environment
        }
      }
    }
  }
  uint32_t word[3][4];
#pragma unroll
  for (int e = 0; e < 3; ++e)
#pragma unroll
    for (int u = 0; u < 4; ++u) word[e][u] = wp[e][u] ? *wp[e][u] : 0u;
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    bool open = false;
#pragma unroll
    for (int u = 0; u < 4; ++u) open = open || (need[e][u] && !(wp[e][u] && ((word[e][u] >> sh[e][u]) & 1u)));
    clear[e] = can[e] && !__any(open);
  }
}

// Solver::isPathFree(a, b) by the wavefront: chunk after chunk until the first hit (the chunks come in sample order, so
// the first chunk with a hit holds the edge's first hit).  Returns free; calls = Collide calls the reference makes.
__device__ __attribute__((noinline)) bool sq_path_free(const EnvView& env, const RobotView& rob, const double* rtri, int32_t* stack, int32_t* cand, int32_t* queue,
                             double* stage, const double* a, const double* b, int32_t* fh_lds, int32_t* ovf_lds, int lane,
                             unsigned long long& calls, unsigned long long& samples, bool& fault) {
  const double* rbox = reinterpret_cast<const double*>(queue + QUEUE_CAP);   // (k_seq_waves: one wave, the boxes lie behind its queue)
  const double parts = edge_parts(a, b);
  const int ns = edge_samples(parts);
  samples += (unsigned long long)ns;
  if (lane == 0) { *fh_lds = 0x7fffffff; *ovf_lds = 0; }
  __builtin_amdgcn_wave_barrier();
  int fh = 0x7fffffff;
  if (env.n_tri != 0) {
    DBG_DECL
    // The samples' clearance bits first, placed like k_query_classify's fused cull places them: in cells of the clearance
    // grid, fp32, a + idx * step (the bits carry the slack for exactly that) - one wavefront is bound by its own
    // instruction stream, and the exact positions cost three fp64 divisions per sample (measured on one box: 14.4 vs
    // 19.4 us per iteration).  Four chunks' lookups are in flight together; only a chunk with a sample left goes through
    // the exact test.
    const float inv = (float)env.clear_inv * __frcp_rn((float)parts);
    const float g0 = (float)((a[0] - env.clear_org[0]) * env.clear_inv), g1 = (float)((a[1] - env.clear_org[1]) * env.clear_inv),
                g2 = (float)((a[2] - env.clear_org[2]) * env.clear_inv);
    const float d0 = (float)(b[0] - a[0]) * inv, d1 = (float)(b[1] - a[1]) * inv, d2 = (float)(b[2] - a[2]) * inv;
    const float nxd = (float)env.clear_n[0], nyd = (float)env.clear_n[1], nzd = (float)env.clear_n[2];
    for (int c0 = 0; c0 * 64 < ns && fh == 0x7fffffff; c0 += 4) {
      const uint32_t* wp[4];
      int sh[4];
      bool need[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = 1 + 64 * (c0 + u) + lane;
        need[u] = idx <= ns;
        wp[u] = nullptr; sh[u] = 0;
        if (need[u] && env.clear_bits_edge) {
          const float td = (float)idx;
          const float fx = __builtin_fmaf(td, d0, g0), fy = __builtin_fmaf(td, d1, g1), fz = __builtin_fmaf(td, d2, g2);
          if (fx >= 0 && fy >= 0 && fz >= 0 && fx < nxd && fy < nyd && fz < nzd) {
            const uint32_t ci = ((uint32_t)(int)fz * (uint32_t)env.clear_n[1] + (uint32_t)(int)fy) * (uint32_t)env.clear_n[0] + (uint32_t)(int)fx;
            wp[u] = env.clear_bits_edge + (ci >> 5);
            sh[u] = (int)(ci & 31u);
          } else if (fx == fx && fy == fy && fz == fz) {
            need[u] = false;                                  // beyond the inflated box of the environment
          }
        }
      }
      uint32_t word[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) word[u] = wp[u] ? *wp[u] : 0u;
      unsigned long long m0, m1, m2, m3;
      {
        unsigned long long mm[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (wp[u] && ((word[u] >> sh[u]) & 1u)) need[u] = false;
          mm[u] = __ballot(need[u]);
        }
        m0 = mm[0]; m1 = mm[1]; m2 = mm[2]; m3 = mm[3];
      }
      for (int u = 0; u < 4 && fh == 0x7fffffff; ++u) {
        const unsigned long long mask = u == 0 ? m0 : (u == 1 ? m1 : (u == 2 ? m2 : m3));
        if (mask == 0ULL) continue;
        segment_chunk(env, rob, rtri, rbox, stack, cand, queue, stage, a, b, 0, c0 + u, true, mask, fh_lds, ovf_lds, lane DBG_PASS);
        __builtin_amdgcn_wave_barrier();
        fh = *fh_lds;
        if (*ovf_lds) {
          // (rare) the chunk's triangle candidate list ran over: its samples one by one through the pose test, which
          // never runs over (it falls back to every triangle) - the same contact definition, so the same first hit
          __builtin_amdgcn_wave_barrier();
          if (lane == 0) { *fh_lds = 0x7fffffff; *ovf_lds = 0; }
          __builtin_amdgcn_wave_barrier();
          fh = 0x7fffffff;
          const double dir[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
          const int last = 64 * (c0 + u) + 64 < ns ? 64 * (c0 + u) + 64 : ns;
          for (int idx = 1 + 64 * (c0 + u); idx <= last && fh == 0x7fffffff; ++idx) {
            double P[6] = {0, 0, 0, 0, 0, 0}, Rm[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, c3[3];
            edge_sample_pos(a, dir, parts, idx, P);
            if (surely_clear_edge(env, P)) continue;
            xform(Rm, P, rob.center, c3);
            if (pose_exact(env, rob, rtri, stack, cand, stage, P, Rm, c3, lane)) fh = idx;
          }
        }
      }
    }
  }
  calls += fh == 0x7fffffff ? (unsigned long long)ns : (unsigned long long)fh;
  return fh == 0x7fffffff;
}

// the k nearest nodes of one tree around qp, by the wavefront (lane j = j-th nearest, (distance, id) order): shells of grid
// cells until the k-th distance lies inside the covered ball; a tree with fewer than k nodes is complete once all are in
// max_id: only nodes with smaller ids (k_spec_waves: a node the leader commits while the step runs is not the worker's
// business); give_up / step: the search ends when that word is beyond step (the leader has moved on: nobody reads the result)
__device__ __attribute__((noinline)) void sq_knn(const GridView& g, const double* qp, int tree, int k, int tcnt, double cell_edge, double slack, int lane,
                       TopK& t, int& have, int max_id = 0x7fffffff, const int32_t* give_up = nullptr, uint32_t step = 0,
                       unsigned long long* dbg = nullptr, const int32_t* lds_cancel = nullptr, int job = 0) {
  t.d = 1.0e300; t.id = 0x7fffffff;
  have = 0;
  unsigned long long d_sh = 0, d_ob = 0, d_cg = 0, d_in = 0;
  if (k <= 0) return;
  const int k_store = k < tcnt ? k : tcnt;
  auto offer = [&](bool valid, const GridItem* src) {
    bool cand = false;
    double d = 1.0e300;
    int id = 0x7fffffff;
    if (valid) {   // (all seven words asked for together: one round trip per batch, whatever the tree)
      const unsigned long long* q8 = reinterpret_cast<const unsigned long long*>(src);
      unsigned long long pw[6];
      for (int q = 0; q < 6; ++q) pw[q] = sq_u64(q8 + q);
      const unsigned long long it = sq_u64(q8 + 6);
      id = (int)(unsigned)(it & 0xffffffffULL);
      if ((int)(unsigned)(it >> 32) == tree && id < max_id) {
        double p6[6];
        for (int q = 0; q < 6; ++q) p6[q] = __longlong_as_double((long long)pw[q]);
        d = dist6(p6, qp);
        cand = true;
      }
    }
    const double worst = topk_worst(t, k, have);
    cand = cand && (have < k || key_less(d, id, worst, 0x7fffffff));
    ++d_ob; d_in += (unsigned long long)__popcll(__ballot(cand));
    topk_merge(t, lane, k, have, cand, d, id);
  };
  int no = sq_i32(g.ovf_cnt);
  if (no > g.ovf_cap) no = g.ovf_cap;
  const int cx = grid_coord((float)qp[0], g.ox, g.inv_cell, g.nx), cy = grid_coord((float)qp[1], g.oy, g.inv_cell, g.ny),
            cz = grid_coord((float)qp[2], g.oz, g.inv_cell, g.nz);
  const int rmax = max(max(g.nx, g.ny), g.nz);
  // Shells 0-2 in one go (the 125 cells of the cube: two counts per lane asked for together, their candidates flattened over
  // the lanes, two batches of items in flight).  A search that a single shell would have ended loses nothing but the surplus
  // candidates: what lies outside shell r is farther than the k-th distance that ended the search there.
  int rr0 = 0;
  {
    int cl[2], mm[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int j = 64 * u + lane;
      cl[u] = 0; mm[u] = 0;
      if (j < 125) {
        const int x = cx + j % 5 - 2, y = cy + (j / 5) % 5 - 2, z = cz + j / 25 - 2;
        if (x >= 0 && x < g.nx && y >= 0 && y < g.ny && z >= 0 && z < g.nz) { cl[u] = (z * g.ny + y) * g.nx + x; mm[u] = sq_i32(g.cnt + cl[u]); }
      }
    }
    ++d_sh; ++d_cg;
    int inc[2], tot[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (mm[u] > g.bk) mm[u] = g.bk;
      inc[u] = mm[u];
      for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(inc[u], off);
        if (lane >= off) inc[u] += o;
      }
      tot[u] = __shfl(inc[u], 63);
    }
    const int total = tot[0] + tot[1];
    for (int base = 0; base < total; base += 128) {
      const GridItem* src[2];
      bool vld[2];
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        const int j = base + 64 * v + lane;
        vld[v] = j < total;
        const int jc = vld[v] ? j : total - 1;
        const bool second = jc >= tot[0];
        const int jj = second ? jc - tot[0] : jc;
        int lo = 0, hi = 63;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          const int im0 = __shfl(inc[0], mid), im1 = __shfl(inc[1], mid);   // (both by every lane: a shuffle under a divergent branch reads nothing from the lanes outside it)
          if ((second ? im1 : im0) > jj) hi = mid; else lo = mid + 1;
        }
        const int c_a = __shfl(cl[0], lo), c_b = __shfl(cl[1], lo);
        const int i_a = __shfl(inc[0], lo), i_b = __shfl(inc[1], lo), m_a = __shfl(mm[0], lo), m_b = __shfl(mm[1], lo);
        const int src_cell = second ? c_b : c_a;
        const int slot = jj - ((second ? i_b : i_a) - (second ? m_b : m_a));
        src[v] = g.items + (size_t)src_cell * g.bk + slot;
      }
      // (the binary search's shuffles need every lane: the loads are issued for both batches before the first is offered)
      unsigned long long pw[2][7];
#pragma unroll
      for (int v = 0; v < 2; ++v)
        if (vld[v]) {
          const unsigned long long* q8 = reinterpret_cast<const unsigned long long*>(src[v]);
          for (int q = 0; q < 7; ++q) pw[v][q] = sq_u64(q8 + q);
        }
      // the candidates of my tree: (distance, id) keys, two per lane
      double cd[2];
      int cid[2];
      bool cv[2];
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        cd[v] = 1.0e300; cid[v] = 0x7fffffff; cv[v] = false;
        if (vld[v]) {
          const int id = (int)(unsigned)(pw[v][6] & 0xffffffffULL);
          if ((int)(unsigned)(pw[v][6] >> 32) == tree && id < max_id) {
            double p6[6];
            for (int q = 0; q < 6; ++q) p6[q] = __longlong_as_double((long long)pw[v][q]);
            cd[v] = dist6(p6, qp);
            cid[v] = id;
            cv[v] = true;
          }
        }
      }
      ++d_ob;
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        if (base + 64 * v >= total) continue;
        const double worst = topk_worst(t, k, have);
        const bool cand = cv[v] && (have < k || key_less(cd[v], cid[v], worst, 0x7fffffff));
        d_in += (unsigned long long)__popcll(__ballot(cand));
        topk_merge(t, lane, k, have, cand, cd[v], cid[v]);
      }
    }
    for (int base = 0; base < no; base += 64) offer(base + lane < no, g.ovf + base + lane);
    const double covered = 2.0 * cell_edge - slack;
    const bool done = (have >= k_store && k_store == tcnt) || (have >= k && topk_worst(t, k, have) <= covered) ||
                      (cx - 2 <= 0 && cy - 2 <= 0 && cz - 2 <= 0 && cx + 2 >= g.nx - 1 && cy + 2 >= g.ny - 1 && cz + 2 >= g.nz - 1);
    rr0 = done ? rmax + 1 : 3;
  }
  for (int rr = rr0; rr <= rmax; ++rr) {
    if (have >= k_store && k_store == tcnt) break;
    if (give_up && rr >= 4 && (uint32_t)__builtin_amdgcn_readfirstlane(sq_i32(give_up)) > step) break;
    if (lds_cancel && __hip_atomic_load(lds_cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= job) break;   // (the attempt was rejected meanwhile)
    // the cells of shell rr only (two full slices + the rings of the slices between them: 6 w^2 - 12 w + 8 of the cube's w^3),
    // the counts of eight batches of 64 cells asked for together
    const int w = 2 * rr + 1, w2 = w * w, per = 4 * w - 4;
    const int S = rr == 0 ? 1 : 6 * w2 - 12 * w + 8;
    ++d_sh;
    for (int b0 = 0; b0 < S; b0 += 512) {
      ++d_cg;
      int mm[8], cl[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = b0 + 64 * u + lane;
        mm[u] = 0; cl[u] = 0;
        if (j < S) {
          int ox, oy, oz;
          if (j < w2) { oz = -rr; ox = j % w - rr; oy = j / w - rr; }
          else if (j < 2 * w2) { const int j2 = j - w2; oz = rr; ox = j2 % w - rr; oy = j2 / w - rr; }
          else {
            const int r = j - 2 * w2, sl = r / per, t2 = r - sl * per;
            oz = -rr + 1 + sl;
            if (t2 < w) { ox = t2 - rr; oy = -rr; }
            else if (t2 < 2 * w) { ox = t2 - w - rr; oy = rr; }
            else if (t2 < 3 * w - 2) { ox = -rr; oy = -rr + 1 + (t2 - 2 * w); }
            else { ox = rr; oy = -rr + 1 + (t2 - (3 * w - 2)); }
          }
          const int x = cx + ox, y = cy + oy, z = cz + oz;
          if (x >= 0 && x < g.nx && y >= 0 && y < g.ny && z >= 0 && z < g.nz) {
            cl[u] = (z * g.ny + y) * g.nx + x;
            mm[u] = sq_i32(g.cnt + cl[u]);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (b0 + 64 * u >= S) continue;
        const int cell = cl[u];
        const int m = mm[u] > g.bk ? g.bk : mm[u];
        if (!__any(m > 0)) continue;
        int inc = m;
        for (int off = 1; off < 64; off <<= 1) {
          const int o = __shfl_up(inc, off);
          if (lane >= off) inc += o;
        }
        const int tot = __shfl(inc, 63);
        for (int base = 0; base < tot; base += 64) {
          const int j = base + lane;
          const int jj = j < tot ? j : tot - 1;
          int lo = 0, hi = 63;
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (__shfl(inc, mid) > jj) hi = mid; else lo = mid + 1;
          }
          const int src_cell = __shfl(cell, lo);
          const int slot = jj - (__shfl(inc, lo) - __shfl(m, lo));
          offer(j < tot, g.items + (size_t)src_cell * g.bk + slot);
        }
      }
    }
    const double covered = (double)rr * cell_edge - slack;
    if (have >= k && topk_worst(t, k, have) <= covered) break;
    if (cx - rr <= 0 && cy - rr <= 0 && cz - rr <= 0 && cx + rr >= g.nx - 1 && cy + rr >= g.ny - 1 && cz + rr >= g.nz - 1) break;
  }
  if (dbg && lane == 0) { atomicAdd(dbg, d_sh); atomicAdd(dbg + 1, d_cg); atomicAdd(dbg + 2, d_ob); atomicAdd(dbg + 3, d_in); atomicAdd(dbg + 4, 1ULL); atomicAdd(dbg + 5, (unsigned long long)no); }
}

template <bool OPT>
__global__ __launch_bounds__(64) void k_seq_waves(SeqArgs A) {
  extern __shared__ double lds_d[];
  __shared__ int32_t s_fh, s_ovf;
  __shared__ int32_t h_id[64], h_tree[64];
  __shared__ double h_d[64], h_pos[64 * 6];
  const DevForestView& f = A.f;
  DevCtrl* c = f.ctrl;
  const int lane = threadIdx.x;
  if (c->halt || c->in_wave) return;            // (a wave the host left half done goes through the round engine)
  double* rtri = lds_d;
  double* stage = rtri + (size_t)A.rob.n_tri * 9;
  int32_t* ibase = reinterpret_cast<int32_t*>(stage + STAGE_DOUBLES);
  int32_t* stack = ibase;                        // (+ the triangle-grid hash set behind it)
  int32_t* cand = ibase + (STACK_CAP + TG_HASH);
  int32_t* queue = cand + CAND_CAP;
  for (int i = lane; i < A.rob.n_tri * 9; i += 64) rtri[i] = A.rob.tri[i];
  __builtin_amdgcn_wave_barrier();
  fill_robot_boxes(rtri, reinterpret_cast<double*>(queue + QUEUE_CAP), A.rob.n_tri, lane, 64);
  __builtin_amdgcn_wave_barrier();
  // ---- the control block, in registers (everything here is the same in every lane)
  int n_nodes = c->n_nodes, iter = c->iter, fn = c->frontier_n, cn = c->closed_n, nb = c->n_borders;
  int solved = c->solved, empty_frontier = c->empty_frontier, terminated = c->terminated;
  const int front_sel = c->front_sel;
  unsigned long long cursor = c->cursor, cc = c->collide_calls, pf = c->path_free_calls, nq = c->nn_queries;
  unsigned long long ex_pose = c->poses_executed, ex_seg = c->segments_executed, ex_smp = c->samples_executed;
  unsigned long long waves = c->waves, rounds = c->rounds, rnodes = c->round_nodes, rqueries = c->round_queries, redraws = 0;
  int32_t* frontier = front_sel ? f.frontier2 : f.frontier;
  const int TM = f.threshold_misses, WP = f.words_per, R = f.n_trees;
  int fault = 0, w_round = 0, w_node = 0, w_pos = 0, w_closed = 0, in_wave = 0;
  unsigned long long st_rounds = 0, st_members = 0, st_rewires = 0;
  // phase clocks (10 ns ticks): pick + node, sample, pose, parent edge, neighbour query, neighbour loop, append, wave end
  uint64_t pre_w[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long pre_at = ~0ULL;
  const bool clk = A.f.profile != 0;   // (a clock read is a scalar memory round trip)
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tq = clk ? wall_clock64() : 0ULL;
  auto lap = [&](int k) { if (!clk) return; const unsigned long long t = wall_clock64(); ph[k] += t - tq; tq = t; };
  for (int wv = 0; wv < A.max_waves && !terminated && !fault; ++wv) {
    // ---- what the round engine checks before a round (round_begin_scalars), and what this launch has to leave to the host
    if (n_nodes + 1 > f.node_cap - 8 || nb + TM > f.border_cap) { fault = SFFK_FAULT_CAPACITY; break; }
    if ((unsigned long long)(nb + TM) * 2ULL > f.bt_mask + 1ULL) { fault = SFFK_FAULT_BORDER_TABLE; break; }
    if (cursor + 8ULL + (unsigned long long)(TM * WP) > A.words_end) break;      // out of engine words: the host tops the ring up
    if (sq_i32(A.grid_ovf_src) > A.grid_ovf_limit) break;                         // the grid wants to re-cell itself
    // ---- node selection (src/forest.h:136-151)
    const int use_closed = cn > 0 && empty_frontier;
    const int pool = use_closed ? cn : fn;
    if (pool < 1) { terminated = 1; break; }
    int pick;
    do { pick = sq_lemire(f.ring[cursor & f.ring_mask], (unsigned long long)pool); ++cursor; if (pick < 0) ++redraws; } while (pick < 0);
    const int node = sq_i32((use_closed ? f.closed : frontier) + pick);
    ++waves;
    double cpos[6];
    for (int k = 0; k < 6; ++k) cpos[k] = sq_f64(A.st.pos + 6 * (size_t)node + k);
    const int mine = sq_i32(A.st.tree + node);
    const double droot_ex = sq_f64(f.d_root + node);
    const bool force = (sq_u8(f.nflag + node) & 1) != 0;
    bool failing = true;
    w_node = node; w_pos = pick; w_closed = use_closed;
    lap(0);
    for (int rd = 0; rd < TM && failing && iter < f.max_iterations; ++rd) {
      // (attempt-start snapshot: a fault rolls exactly this attempt back)
      const int iter_a = iter;
      const unsigned long long cur_a = cursor, cc_a = cc, pf_a = pf, nq_a = nq, xp_a = ex_pose, xs_a = ex_seg, xm_a = ex_smp;
      bool flt = false;
      // (the words of this attempt were asked for while the previous one ran, whenever the stream position was the
      // expected one; the next attempt's are asked for now)
      uint64_t w[6];
      if (pre_at == cursor) { for (int k = 0; k < 6; ++k) w[k] = pre_w[k]; }
      else { for (int k = 0; k < 6; ++k) w[k] = k < WP ? f.ring[(cursor + k) & f.ring_mask] : 0ULL; }
      pre_at = cursor + (unsigned long long)WP;
      for (int k = 0; k < 6; ++k) pre_w[k] = k < WP ? f.ring[(pre_at + k) & f.ring_mask] : 0ULL;
      SampleTrig ht{};
      if (A.trig) {
        const double* t0 = A.trig + 3 * (size_t)(cursor & f.ring_mask);
        ht.c_phi = t0[0]; ht.s_phi = t0[1];
        if (WP == 6) {
          const double* t1 = A.trig + 3 * (size_t)((cursor + 1) & f.ring_mask);
          const double* t3 = A.trig + 3 * (size_t)((cursor + 3) & f.ring_mask);
          ht.c_theta = t1[0]; ht.s_theta = t1[1]; ht.acos_u = t3[2];
        }
      }
      cursor += (unsigned long long)WP;
      ++iter;
      ++rounds; rnodes += (unsigned long long)(n_nodes + 1); ++rqueries;
      double qp[6];
      if (!A.trig) {
        // the five transcendental values of the sample, two at a time: lanes 0 / 1 evaluate the same function on phi / theta
        // (one instruction stream whatever the lane count; the same portable routines, so the same bits as sample_point)
        const double ang = sample_angle(lane == 1 ? w[1] : w[0]);
        const double sv = sffp::psin(ang), cv = sffp::pcos(ang);
        ht.s_phi = __shfl(sv, 0); ht.c_phi = __shfl(cv, 0);
        ht.s_theta = __shfl(sv, 1); ht.c_theta = __shfl(cv, 1);
        ht.acos_u = WP == 6 ? sffp::pacos(sample_acos_arg(w[3])) : 0.0;
      }
      const bool ok = sample_point_with(w, cpos, A.sampling_dist, A.dim, A.limits, qp, ht);
      lap(1);
      if (!ok) continue;                                           // :246 !result
      // ---- Environment::Collide(newPoint)
      cc += 1; ex_pose += 1;
      bool hit = false;
      if (A.env.n_tri != 0 && !surely_clear(A.env, qp)) {
        double Rm[9], c3[3];
        if (qp[3] == 0 && qp[4] == 0 && qp[5] == 0) { Rm[0] = Rm[4] = Rm[8] = 1; Rm[1] = Rm[2] = Rm[3] = Rm[5] = Rm[6] = Rm[7] = 0; }
        else rotation(qp, Rm);
        xform(Rm, qp, A.rob.center, c3);
        hit = pose_exact(A.env, A.rob, rtri, stack, cand, stage, qp, Rm, c3, lane);
      }
      lap(2);
      if (hit) continue;
      // ---- isPathFree(expanded, newPoint)
      pf += 1; ex_seg += 1;
      const bool free0 = (sq_edge_clear_fast(A.env, cpos, qp, lane, cc, ex_smp) || sq_path_free(A.env, A.rob, rtri, stack, cand, queue, stage, cpos, qp, &s_fh, &s_ovf, lane, cc, ex_smp, flt));
      bool reject = !free0;
      lap(3);
      const double pdist = dist6(cpos, qp);                        // parentDistance, :250
      int n_hit = 0;
      if (!flt && !reject) {
        nq += (unsigned long long)R;                               // :262-267 one radiusSearch per tree
        // ---- the neighbours: exact 6-D ball of radius max(parentDistance, treeDistance) from the cells its box touches
        const double r = pdist > A.dist_tree ? pdist : A.dist_tree;
        const double ri = (r + A.sweep_abs_eps) * (1.0 + 1e-5);
        const float rf = sqrtf((float)(ri * ri) * 1.000001f) * 1.000001f;
        const GridView& g = A.g;
        const float qx = (float)qp[0], qy = (float)qp[1], qz = (float)qp[2];
        const int lx = grid_coord(qx - rf, g.ox, g.inv_cell, g.nx), hx = grid_coord(qx + rf, g.ox, g.inv_cell, g.nx);
        const int ly = grid_coord(qy - rf, g.oy, g.inv_cell, g.ny), hy = grid_coord(qy + rf, g.oy, g.inv_cell, g.ny);
        const int lz = grid_coord(qz - rf, g.oz, g.inv_cell, g.nz), hz = grid_coord(qz + rf, g.oz, g.inv_cell, g.nz);
        const int wx = hx - lx + 1, wy = hy - ly + 1, wz = hz - lz + 1;
        const int total = wx * wy * wz;
        auto take = [&](bool valid, const GridItem* src) {        // one candidate per lane -> the hit list in LDS
          bool h = false;
          double d = 0, p6[6];
          int id = 0, tr = 0;
          if (valid) {
            const unsigned long long* q8 = reinterpret_cast<const unsigned long long*>(src);
            for (int k = 0; k < 6; ++k) p6[k] = __longlong_as_double((long long)sq_u64(q8 + k));
            const unsigned long long it = sq_u64(q8 + 6);
            id = (int)(unsigned)(it & 0xffffffffULL); tr = (int)(unsigned)(it >> 32);
            d = dist6(p6, qp);
            h = d < r;
          }
          const unsigned long long hm = __ballot(h);
          if (h) {
            const int at = n_hit + __popcll(hm & ((1ULL << lane) - 1ULL));
            if (at < 64) { h_id[at] = id; h_tree[at] = tr; h_d[at] = d; for (int k = 0; k < 6; ++k) h_pos[6 * at + k] = p6[k]; }
          }
          n_hit += __popcll(hm);
        };
        for (int c0 = 0; c0 < total; c0 += 64) {
          const int ci = c0 + lane;
          int cell = 0, m = 0;
          if (ci < total) {
            const int q1 = ci / wx, q2 = q1 / wy;
            cell = ((lz + q2) * g.ny + (ly + q1 - q2 * wy)) * g.nx + (lx + ci - q1 * wx);
            m = sq_i32(g.cnt + cell);
            if (m > g.bk) m = g.bk;
          }
          int inc = m;
          for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(inc, off);
            if (lane >= off) inc += o;
          }
          const int tot = __shfl(inc, 63);
          for (int base = 0; base < tot; base += 64) {
            const int j = base + lane;
            const int jj = j < tot ? j : tot - 1;
            int lo = 0, hi = 63;
            while (lo < hi) {
              const int mid = (lo + hi) >> 1;
              if (__shfl(inc, mid) > jj) hi = mid; else lo = mid + 1;
            }
            const int src_cell = __shfl(cell, lo);
            const int slot = jj - (__shfl(inc, lo) - __shfl(m, lo));
            take(j < tot, g.items + (size_t)src_cell * g.bk + slot);
          }
        }
        int no = sq_i32(g.ovf_cnt);
        if (no > g.ovf_cap) no = g.ovf_cap;
        for (int base = 0; base < no; base += 64) take(base + lane < no, g.ovf + base + lane);
        if (n_hit > A.hit_cap || n_hit > 64) flt = true;
      }
      lap(4);
      if (!flt && !reject) {
        // ---- the neighbour loop (:270-300) in the reference's order: tree id, then distance, then id; an edge is only
        // checked when the loop reaches it
        __builtin_amdgcn_wave_barrier();
        const bool have = lane < n_hit;
        const int id = have ? h_id[lane] : 0x7fffffff;
        const int t = have ? h_tree[lane] : 0x7fffffff;
        const double d = have ? h_d[lane] : 0.0;
        const bool same = t == mine;
        const bool qk = have && (same ? (!force && d < pdist - SFFG_TOL) : (d < A.dist_tree - SFFG_TOL));   // :276 / :283
        int rank = 0;
        for (int j = 0; j < n_hit; ++j) {
          const int tj = __shfl(t, j), idj = __shfl(id, j), qj = __shfl((int)qk, j);
          const double dj = __shfl(d, j);
          if (qj && (tj < t || (tj == t && (dj < d || (dj == d && idj < id))))) ++rank;
        }
        const int n_q = __popcll(__ballot(qk));
        for (int rk = 0; rk < n_q && !reject && !flt; ++rk) {
          const unsigned long long sel = __ballot(qk && rank == rk);
          const int src = __ffsll((long long)sel) - 1;
          const int s_same = __shfl((int)same, src), s_id = __shfl(id, src), s_tree = __shfl(t, src);
          double np6[6];
          for (int k = 0; k < 6; ++k) np6[k] = h_pos[6 * src + k];
          pf += 1; ex_seg += 1;
          if (s_same) {
            const bool fr = (sq_edge_clear_fast(A.env, np6, qp, lane, cc, ex_smp) || sq_path_free(A.env, A.rob, rtri, stack, cand, queue, stage, np6, qp, &s_fh, &s_ovf, lane, cc, ex_smp, flt));
            if (fr) reject = true;                                 // :276-280 overcrowded
          } else {
            const bool fr = (sq_edge_clear_fast(A.env, cpos, np6, lane, cc, ex_smp) || sq_path_free(A.env, A.rob, rtri, stack, cand, queue, stage, cpos, np6, &s_fh, &s_ovf, lane, cc, ex_smp, flt));
            if (fr && !flt) {                                      // :288-294 border entry unless the pair has one
              const int a = s_id < node ? s_id : node, b = s_id < node ? node : s_id;
              const unsigned long long key = ((unsigned long long)(uint32_t)a << 32) | ((unsigned long long)(uint32_t)b + 1ULL);
              size_t h = (size_t)((key * 0x9E3779B97F4A7C15ULL) >> 17) & (size_t)f.bt_mask;
              bool fresh = false;
              for (int guard = 0; guard < (1 << 24); ++guard) {
                const unsigned long long cur = sq_u64(f.bt_key + h);
                if (cur == key) { fresh = sq_u64(f.bt_val + h) == ~0ULL; break; }
                if (cur == 0ULL) { fresh = true; break; }
                h = (h + 1) & (size_t)f.bt_mask;
              }
              if (fresh) {
                if (lane == 0) {
                  f.bt_key[h] = key;
                  f.bt_val[h] = c->epoch << 32;
                  f.b_n1[nb] = a; f.b_n2[nb] = b;
                  f.b_ta[nb] = s_tree < mine ? s_tree : mine; f.b_tb[nb] = s_tree < mine ? mine : s_tree;
                  f.b_dist[nb] = sq_f64(f.d_root + s_id) + droot_ex + dist6(np6, cpos);
                  f.pair[(size_t)s_tree * R + mine] = 1;
                  f.pair[(size_t)mine * R + s_tree] = 1;
                }
                sq_drain();
                ++nb;
              }
            }
            reject = true;                                         // :296-299
          }
        }
      }
      lap(5);
      if (flt) {
        // a bounded list overflowed (hits, triangle candidates): this attempt never happened - the host finishes the wave
        iter = iter_a; cursor = cur_a; cc = cc_a; pf = pf_a; nq = nq_a; ex_pose = xp_a; ex_seg = xs_a; ex_smp = xm_a;
        --rounds; rnodes -= (unsigned long long)(n_nodes + 1); --rqueries;
        fault = SFFK_FAULT_LISTS; w_round = rd; in_wave = 1;
        break;
      }
      if (reject) continue;
      // ---- SFF* (:307-351): the k nearest of the tree, choose parent, (the node), rewire - each edge checked when its turn comes
      int par_new = node;
      double dcl_new = pdist, best = pdist + droot_ex;
      TopK mt{1.0e300, 0x7fffffff};
      int n_mem = 0;
      double m_droot = 0;
      if (OPT) {
        const int k = __popcll(__ballot(lane > 0 && lane <= SFFK_STAR_KMAX + 1 && A.ktab[lane] <= n_nodes));   // (size_t)(2e log10 N), :309
        if (k > SFFK_STAR_KMAX) flt = true;
        else {
          nq += 1;                                                                          // :317 knnSearch
          sq_knn(A.g, qp, mine, k, sq_i32(A.tree_cnt + 16 * mine), A.cell_edge, A.knn_slack, lane, mt, n_mem);
          if (lane < n_mem) m_droot = sq_f64(f.d_root + mt.id);
          for (int m = 0; m < n_mem && !flt; ++m) {                                         // :320-327
            const double nd = __shfl(mt.d, m) + __shfl(m_droot, m);
            if (nd < best - SFFG_TOL) {
              const int idm = __shfl(mt.id, m);
              double mp[6];
              for (int q = 0; q < 6; ++q) mp[q] = sq_f64(A.st.pos + 6 * (size_t)idm + q);
              pf += 1; ex_seg += 1;
              if ((sq_edge_clear_fast(A.env, qp, mp, lane, cc, ex_smp) || sq_path_free(A.env, A.rob, rtri, stack, cand, queue, stage, qp, mp, &s_fh, &s_ovf, lane, cc, ex_smp, flt)) && !flt) {
                best = nd; par_new = idm; dcl_new = __shfl(mt.d, m);
              }
            }
          }
        }
        if (flt) {
          iter = iter_a; cursor = cur_a; cc = cc_a; pf = pf_a; nq = nq_a; ex_pose = xp_a; ex_seg = xs_a; ex_smp = xm_a;
          --rounds; rnodes -= (unsigned long long)(n_nodes + 1); --rqueries;
          fault = SFFK_FAULT_LISTS; w_round = rd; in_wave = 1;
          break;
        }
      }
      // ---- the new node (:329, :353-367)
      const int idn = n_nodes;
      if (lane == 0) {
        const size_t o = (size_t)idn;
        A.st.x[o] = (float)qp[0]; A.st.y[o] = (float)qp[1]; A.st.z[o] = (float)qp[2];
        A.st.yaw[o] = (float)qp[3]; A.st.pitch[o] = (float)qp[4]; A.st.roll[o] = (float)qp[5];
        for (int k = 0; k < 6; ++k) A.st.pos[6 * o + k] = qp[k];
        A.st.tree[o] = mine;
        f.parent[o] = par_new;
        f.d_closest[o] = dcl_new;
        f.d_root[o] = best;
        f.iter[o] = (uint32_t)iter;
        f.nflag[o] = 2;
        frontier[fn] = idn;
        if (OPT) {
          atomicAdd(A.tree_cnt + 16 * mine, 1);
          if (A.hist) {
            const int at = atomicAdd(A.hist_ctl, 1);
            if (at < A.hist_cap) { A.hist[3 * (size_t)at] = idn; A.hist[3 * (size_t)at + 1] = par_new; A.hist[3 * (size_t)at + 2] = iter; }
            else A.hist_ctl[1] = 1;
          }
        }
        GridItem it;
        for (int k = 0; k < 6; ++k) it.p[k] = qp[k];
        it.id = idn; it.tree = mine; it.pad[0] = it.pad[1] = 0;
        grid_put(A.g, it);
      }
      sq_drain();
      ++n_nodes; ++fn;
      failing = false;
      if (OPT) {
        // rewire (:332-350): a member the new node's cost improves, if the edge member -> new is free
        ++st_rounds; st_members += (unsigned long long)n_mem;
        for (int m = 0; m < n_mem; ++m) {
          const double dm = __shfl(mt.d, m), drm = __shfl(m_droot, m);
          const double proposed = best + dm;
          if (proposed < drm - SFFG_TOL) {
            const int idm = __shfl(mt.id, m);
            double mp[6];
            for (int q = 0; q < 6; ++q) mp[q] = sq_f64(A.st.pos + 6 * (size_t)idm + q);
            pf += 1; ex_seg += 1;
            bool f2 = false;
            const bool fr = (sq_edge_clear_fast(A.env, mp, qp, lane, cc, ex_smp) || sq_path_free(A.env, A.rob, rtri, stack, cand, queue, stage, mp, qp, &s_fh, &s_ovf, lane, cc, ex_smp, f2));
            if (fr) {
              if (lane == 0) {
                f.parent[idm] = idn; f.d_closest[idm] = dm; f.d_root[idm] = proposed;
                if (A.hist) {
                  const int at = atomicAdd(A.hist_ctl, 1);
                  if (at < A.hist_cap) { A.hist[3 * (size_t)at] = idm; A.hist[3 * (size_t)at + 1] = idn; A.hist[3 * (size_t)at + 2] = iter; }
                  else A.hist_ctl[1] = 1;
                }
              }
              ++st_rewires;
            }
          }
        }
        sq_drain();
      }
      lap(6);
    }
    if (fault) break;
    // ---- the slot is exhausted: its node leaves the frontier for the closed list (:160-178; the erase keeps the order)
    if (failing && !use_closed) {
      const int fl = sq_u8(f.nflag + node);
      if (fl & 2) {
        if (lane == 0) { f.nflag[node] = (uint8_t)((fl & ~2) | 1); f.closed[cn] = node; }
        ++cn;
        for (int j0 = pick; j0 < fn - 1; j0 += 256) {
          int v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) { const int j = j0 + 64 * u + lane; v[u] = j < fn - 1 ? sq_i32(frontier + j + 1) : 0; }
#pragma unroll
          for (int u = 0; u < 4; ++u) { const int j = j0 + 64 * u + lane; if (j < fn - 1) frontier[j] = v[u]; }
          sq_drain();
        }
        --fn;
      }
    }
    sq_drain();
    // ---- termination (:184-201)
    empty_frontier = fn == 0 ? 1 : 0;
    if (!solved && empty_frontier) {
      // maxConnected() == numRoots: every tree reachable from tree 0 over pairs that hold a border (R <= 64: a lane per tree)
      unsigned long long reach = 1ULL, frontier_set = 1ULL;
      if (R <= 64) {
        unsigned long long row = 0ULL;   // lane a: bit b = pair (a, b) has a border
        if (lane < R) for (int b2 = 0; b2 < R; ++b2) if (sq_u8(f.pair + (size_t)lane * R + b2)) row |= 1ULL << b2;
        while (frontier_set) {
          const int a = __ffsll((long long)frontier_set) - 1;
          frontier_set &= frontier_set - 1;
          const unsigned long long ra = __shfl(row, a) & ~reach;
          reach |= ra; frontier_set |= ra;
        }
        solved = __popcll(reach) == R ? 1 : 0;
      } else fault = SFFK_FAULT_LISTS;   // (more trees than lanes: the round engine's serial walk)
    }
    const bool budget = f.node_budget > 0 && n_nodes >= f.node_budget;
    terminated = (solved || iter >= f.max_iterations || budget) ? 1 : 0;
    if (A.trace && wv < A.trace_cap && lane == 0) {
      int32_t* t = A.trace + 8 * (size_t)wv;
      t[0] = node; t[1] = pick; t[2] = iter; t[3] = (int32_t)(cursor & 0x7fffffffULL); t[4] = failing ? TM : 0; t[5] = n_nodes; t[6] = fn; t[7] = cn;
    }
    lap(7);
  }
  if (lane == 0) {
    for (int k = 0; k < 8; ++k) c->wprof[k] += ph[k];
    c->n_nodes = n_nodes; c->iter = iter; c->frontier_n = fn; c->closed_n = cn; c->n_borders = nb;
    c->solved = solved; c->empty_frontier = empty_frontier; c->terminated = terminated;
    c->cursor = cursor; c->collide_calls = cc; c->path_free_calls = pf; c->nn_queries = nq;
    c->poses_executed = ex_pose; c->segments_executed = ex_seg; c->samples_executed = ex_smp;
    c->waves = waves; c->rounds = rounds; c->round_nodes = rnodes; c->round_queries = rqueries;
    c->redraws += (int)redraws;
    c->star_rounds += st_rounds; c->star_passes += st_rounds; c->star_members += st_members; c->star_rewires += st_rewires;
    c->n_act = 0; c->app_n = 0; c->compact_from = 0;
    c->grid_ovf = sq_i32(A.grid_ovf_src); c->tgrid_ovf = 0;
    c->fault = fault;
    c->halt = (terminated || fault) ? 1 : 0;
    c->in_wave = in_wave;
    if (in_wave) {   // the state the host engine resumes the wave from: its one slot, still failing, w_round rounds done
      c->round = w_round; c->n_slots = 1; c->use_closed = w_closed; c->act_sel = 0; c->act_cnt = 1;
      f.slot_node[0] = w_node; f.slot_pos[0] = w_pos; f.act_slot[0] = 0;
    } else c->round = 0;
  }
}

void launch_seq_waves(hipStream_t s, const SeqArgs& a) {

  const size_t lds = collide_lds_bytes(a.rob.n_tri, 1);
  if (a.optimize) {
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_seq_waves<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_seq_waves<true>, dim3(1), dim3(64), lds, s, a);
  } else {
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_seq_waves<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_seq_waves<false>, dim3(1), dim3(64), lds, s, a);
  }
}

// ------------------------------------------------------------------ waves of one slot, SPECULATED (kernels.h: SpecArgs)
// write-through stores: what the leader writes and a worker on another XCD reads in the same launch
__device__ __forceinline__ void wt_i32(int32_t* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void wt_u64(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void wt_f64(double* p, double v) {
  wt_u64(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v));
}
__device__ __forceinline__ void wt_u8(uint8_t* p, uint8_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long sp_gran(uint32_t tag, uint32_t v) { return ((unsigned long long)tag << 32) | (unsigned long long)v; }
__device__ __forceinline__ uint32_t sp_lo(double v) { return (uint32_t)((unsigned long long)__double_as_longlong(v) & 0xffffffffULL); }
__device__ __forceinline__ uint32_t sp_hi(double v) { return (uint32_t)((unsigned long long)__double_as_longlong(v) >> 32); }
__device__ __forceinline__ double sp_f64(uint32_t lo, uint32_t hi) {
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | (unsigned long long)lo));
}
// record row 0 (granule = {value, step}); doubles as two granules (lo, hi)
#define SPG_STATUS 0
#define SPG_NODE 1
#define SPG_ITER 2
#define SPG_CUR 3      // + 4
#define SPG_NN 5
#define SPG_CC 6
#define SPG_PF 7
#define SPG_NQ 8
#define SPG_EVENT 9
#define SPG_EV_ID 10
#define SPG_EV_TREE 11
#define SPG_EV_DIST 12 // + 13
#define SPG_MINE 14
#define SPG_QP 15      // .. 26
#define SPG_BEST 27    // + 28
#define SPG_PAR 29
#define SPG_DCL 30     // + 31
#define SPG_NRW 32
#define SPG_NMEM 33
#define SPG_PICK 34     // position of the node in the frontier array (or the closed list)
#define SPG_UCL 35      // ... which of the two
#define SPG_FL 36       // the node's flags (bit 1: in the frontier)
#define SPG_USED 37
#define SPS_REJECT 1
#define SPS_ACCEPT 2
#define SPS_FAULT 3
#define SPS_INVALID 4
#define SPS_SKIPPED 5
// the published control block of a step
#define SPB_CUR 0      // + 1
#define SPB_FN 2
#define SPB_CN 3
#define SPB_NN 4
#define SPB_ITER 5
#define SPB_EF 6
#define SPB_NE 7       // frontier positions erased since the array was last compacted (SPB_ER + ...)
#define SPB_FNPC 8     // entries of the array when it was last compacted
#define SPB_NNC 9      // nodes then: the array's entry FNPC + i is node NNC + i
#define SPB_NNS 10     // nodes whose store entries are complete; the NN - NNS behind them are PENDING: accepted in the step before, written
                       // by the leader while this step is evaluated - their early rows travel in the block (SPB_PEND)
#define SPB_PEND 16    // pending nodes: 16 granules each (the early row's layout), SFFK_SPEC_DEPTH of them at most
#define SPB_ER (SPB_PEND + 16 * SFFK_SPEC_DEPTH)   // the erased positions, ascending (SP_ER_MAX)
#define SP_ER_MAX 48
#define SPB_USED (SPB_ER + SP_ER_MAX)
#define SP_BASE 128    // granules per control block
#define SP_QUIT 0xffffffffu

// LDS words two wavefronts of a workgroup hand to each other (in-order LDS operations of a wavefront; the words themselves
// through workgroup-scope atomics so that the compiler keeps every access)
__device__ __forceinline__ int lds_ld(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st(int32_t* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// one wavefront's own LDS traffic: what a lane wrote, the other lanes read behind this
#define SP_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// The sorted list of erased frontier positions (LDS, n <= 64 entries), by one wavefront.
// Position in the array of the logical-th entry that is still there.  The sequential rule - walk the list in ascending order,
// every erased position at or below the running index pushes it up by one - hits a PREFIX of the list (once an entry lies
// above the index all later ones do), and entry e is in that prefix iff er[e] - e <= logical (er[e] - e never decreases): one
// LDS read per lane and a ballot instead of a loop over the list.
__device__ __forceinline__ int sp_er_map(const int32_t* er, int n, int logical, int lane) {
  const bool hit = lane < n && er[lane] - lane <= logical;
  return logical + __popcll(__ballot(hit));
}
// inserts idx (not in the list), keeping it sorted: every lane moves its entry up by one if it lies above idx
__device__ __forceinline__ void sp_er_insert(int32_t* er, int& n, int idx, int lane) {
  SP_WAVE_SYNC();
  const int mine = lane < n ? er[lane] : 0x7fffffff;
  const int at = __popcll(__ballot(lane < n && mine < idx));
  SP_WAVE_SYNC();
  if (lane < n && mine > idx) er[lane + 1] = mine;
  if (lane == 0) er[at] = idx;
  ++n;
  SP_WAVE_SYNC();
}

template <bool OPT>
__global__ __launch_bounds__(OPT ? 192 : 128) void k_spec_waves(SpecArgs S) {
  extern __shared__ double lds_d[];
  __shared__ int32_t s_fh, s_ovf;
  __shared__ int32_t h_id[64], h_tree[64];
  __shared__ double h_d[64], h_pos[64 * 6];
  __shared__ uint32_t s_row[64];
  __shared__ double p_pos[2 * SFFK_SPEC_DEPTH * 6], p_best[2 * SFFK_SPEC_DEPTH];
  __shared__ int32_t p_tree[2 * SFFK_SPEC_DEPTH];
  __shared__ uint32_t s_rw[OPT ? SFFK_STAR_KC * 5 : 1];
  __shared__ uint32_t s_acc[SFFK_SPEC_DEPTH][64];           // leader: row 0 of the step's accepted attempts (applied behind the next step's publication); worker: the pending rows
  __shared__ int32_t s_acc_id[SFFK_SPEC_DEPTH], s_acc_iter[SFFK_SPEC_DEPTH];
  __shared__ int32_t s_er[SP_ER_MAX + SFFK_SPEC_DEPTH + 1];   // erased frontier positions, ascending (leader: since the last compaction; worker: + its scenario's)
  // the job the worker's first wavefront hands to its second one right after the sample is drawn: the neighbour query
  // (and, SFF*, the k nearest of the sample's tree) run BESIDE the pose check and the parent edge
  __shared__ int32_t j_seq, j_cancel, j_done, j_done_k, j_nhit, j_mine, j_k, j_nmem, j_nn0, j_snn, j_step;
  __shared__ double j_qp[6], j_pdist;
  __shared__ double k_d[64];
  __shared__ int32_t k_id[64];
  const SeqArgs& A = S.q;
  const DevForestView& f = A.f;
  DevCtrl* c = f.ctrl;
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  if (c->halt || c->in_wave) return;            // (a wave the host left half done goes through the round engine)
  if (threadIdx.x == 0) { j_seq = 0; j_cancel = 0; j_done = 0; j_done_k = 0; }
  __syncthreads();
  const int TM = f.threshold_misses, WP = f.words_per, R = f.n_trees;
  const int front_sel = c->front_sel;
  int32_t* frontier = front_sel ? f.frontier2 : f.frontier;

  if (blockIdx.x == 0) {
    // =================================================================== the leader: owns the forest, commits in order
    if (wv) return;
    int n_nodes = c->n_nodes, iter = c->iter, fn = c->frontier_n, cn = c->closed_n, nb = c->n_borders;
    int solved = c->solved, empty_frontier = c->empty_frontier, terminated = c->terminated;
    unsigned long long cursor = c->cursor, cc = c->collide_calls, pf = c->path_free_calls, nq = c->nn_queries;
    unsigned long long waves = c->waves, rounds = c->rounds, rnodes = c->round_nodes, rqueries = c->round_queries, redraws = 0;
    int fault = 0, w_round = 0, w_node = 0, w_pos = 0, w_closed = 0, in_wave = 0, stalled = 0;
    unsigned long long st_rounds = 0, st_members = 0, st_rewires = 0, n_steps = 0, n_commit = 0;
    uint32_t step = 0;
    int waves_done = 0;
    bool stop = false;
    const unsigned long long epoch = c->epoch;
    // The frontier ARRAY only ever grows while the launch runs (its workers read it whenever they get to it): new nodes go
    // behind its last entry, and the positions of the nodes that left it are kept as a sorted list (the workers get the list
    // with the control block and map their picks through it, adding the erases their scenario assumes).  When the list is
    // nearly full - and when the launch ends - the array is compacted in one pass.  Since the last compaction: fn_pc entries
    // then, nn_c nodes then (entry fn_pc + i holds node nn_c + i), ner erased positions in s_er.
    int fn_pc = fn, nn_c = n_nodes, ner = 0, nn_base = n_nodes;
    auto er_map = [&](int logical) -> int { return sp_er_map(s_er, ner, logical, lane); };
    auto er_insert = [&](int idx) { sp_er_insert(s_er, ner, idx, lane); };
    auto compact = [&]() {
      if (ner > 0) {
        const int pfn = fn_pc + (n_nodes - nn_c);
        // (one pass from the first erased position: an entry moves down by the number of erased positions below it; a block's
        // stores go below everything a later block loads, and loads return in order: no wait between the blocks)
        const int first = s_er[0];
        for (int j0 = first; j0 < pfn; j0 += 1024) {
          int v[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) { const int j = j0 + 64 * u + lane; v[u] = sq_i32(frontier + (j < pfn ? j : pfn - 1)); }
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            const int j = j0 + 64 * u + lane;
            if (j >= pfn) continue;
            // (erased positions below j, by bisection of the sorted list)
            int lo = 0, hi = ner;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_er[mid] < j) lo = mid + 1; else hi = mid; }
            const bool gone = lo < ner && s_er[lo] == j;
            if (!gone) wt_i32(frontier + j - lo, v[u]);
          }
        }
      }
      ner = 0; fn_pc = fn; nn_c = n_nodes;
      sq_drain();
      SP_WAVE_SYNC();
    };
    // phase clocks (10 ns ticks, SFFGPU_PROFILE): publish, wait for a wave's first record, its other records, the accepted
    // node, closed list + termination, the erases
    const bool clk = f.profile != 0;
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tq = clk ? wall_clock64() : 0ULL;
    auto lap = [&](int k) { if (!clk) return; const unsigned long long t = wall_clock64(); ph[k] += t - tq; tq = t; };
    int grid_ovf_now = sq_i32(A.grid_ovf_src);   // (read once: only apply_accept below adds to the overflow list while the launch runs)
    // the node of an accepted attempt (row 0 of its record): a lane per word - one store instruction per width instead of
    // thirty stores of lane 0
    auto apply_accept = [&](const uint32_t* row, int idn, int iter_v, int n_rw) {
      const int par_new = (int)row[SPG_PAR], mine = (int)row[SPG_MINE];
      sq_drain();                            // (an earlier node of the step may have written the cell's count)
      const GridView& g = A.g;
      const float q0 = (float)sp_f64(row[SPG_QP], row[SPG_QP + 1]), q1 = (float)sp_f64(row[SPG_QP + 2], row[SPG_QP + 3]),
                  q2 = (float)sp_f64(row[SPG_QP + 4], row[SPG_QP + 5]);
      const size_t cell = grid_cell_of(g, q0, q1, q2);
      const int gslot = sq_i32(g.cnt + cell);
      int govf = -1;
      if (gslot >= g.bk) govf = sq_i32(g.ovf_cnt);
      if (g.occ && lane == 0) atomicOr(g.occ + (cell >> 5), 1u << (cell & 31));
      const bool put = gslot < g.bk || govf < g.ovf_cap;          // (the host checks ovf_cnt against ovf_cap)
      GridItem* gi = gslot < g.bk ? g.items + cell * g.bk + gslot : g.ovf + (govf < 0 ? 0 : govf);
      GridItem32* gl = gslot < g.bk ? (g.lite ? g.lite + cell * g.bk + gslot : nullptr) : (g.ovf_lite ? g.ovf_lite + (govf < 0 ? 0 : govf) : nullptr);
      const size_t o = (size_t)idn;
      {   // 64-bit words: position (6), distance to the parent, cost, the grid item (8)
        unsigned long long* p64 = nullptr;
        unsigned long long v64 = 0;
        const int qk = lane < 6 ? lane : (lane >= 8 && lane < 14 ? lane - 8 : 0);
        const unsigned long long qbits = ((unsigned long long)row[SPG_QP + 2 * qk + 1] << 32) | (unsigned long long)row[SPG_QP + 2 * qk];
        if (lane < 6) { p64 = reinterpret_cast<unsigned long long*>(A.st.pos + 6 * o + lane); v64 = qbits; }
        else if (lane == 6) { p64 = reinterpret_cast<unsigned long long*>(f.d_closest + o); v64 = ((unsigned long long)row[SPG_DCL + 1] << 32) | row[SPG_DCL]; }
        else if (lane == 7) { p64 = reinterpret_cast<unsigned long long*>(f.d_root + o); v64 = ((unsigned long long)row[SPG_BEST + 1] << 32) | row[SPG_BEST]; }
        else if (lane < 16 && put) {
          p64 = reinterpret_cast<unsigned long long*>(gi) + (lane - 8);
          v64 = lane < 14 ? qbits : (lane == 14 ? (((unsigned long long)(uint32_t)mine << 32) | (unsigned long long)(uint32_t)idn) : 0ULL);
        }
        if (p64) wt_u64(p64, v64);
      }
      {   // 32-bit words: tree, parent, frontier entry (write-through); the fp32 columns, the filter record, the iteration (plain)
        const int fk = lane >= 19 && lane < 25 ? lane - 19 : (lane >= 25 && lane < 31 ? lane - 25 : 0);
        const float qf = (float)sp_f64(row[SPG_QP + 2 * fk], row[SPG_QP + 2 * fk + 1]);
        if (lane == 16) wt_i32(A.st.tree + o, mine);
        else if (lane == 17) wt_i32(f.parent + o, par_new);
        else if (lane == 18) wt_i32(frontier + fn_pc + (idn - nn_c), idn);
        else if (lane >= 19 && lane < 25) {
          float* col = fk == 0 ? A.st.x : (fk == 1 ? A.st.y : (fk == 2 ? A.st.z : (fk == 3 ? A.st.yaw : (fk == 4 ? A.st.pitch : A.st.roll))));
          col[o] = qf;
        } else if (lane >= 25 && lane < 31) { if (gl && put) reinterpret_cast<float*>(gl)[fk] = qf; }
        else if (lane == 31) { if (gl && put) gl->id = idn; }
        else if (lane == 32) { if (gl && put) gl->tree = mine; }
        else if (lane == 33) f.iter[o] = (uint32_t)iter_v;
        else if (lane == 34) wt_u8(f.nflag + o, 2);
      }
      if (OPT) {
        SP_WAVE_SYNC();
        if (lane == 0) {
          if (A.hist) {
            const int at = atomicAdd(A.hist_ctl, 1);
            if (at < A.hist_cap) { A.hist[3 * (size_t)at] = idn; A.hist[3 * (size_t)at + 1] = par_new; A.hist[3 * (size_t)at + 2] = iter_v; }
            else A.hist_ctl[1] = 1;
          }
          // rewire (:332-350), as the worker found them in the reference's order
          for (int j = 0; j < n_rw; ++j) {
            const int idm = (int)s_rw[5 * j];
            wt_i32(f.parent + idm, idn);
            wt_f64(f.d_closest + idm, sp_f64(s_rw[5 * j + 1], s_rw[5 * j + 2]));
            wt_f64(f.d_root + idm, sp_f64(s_rw[5 * j + 3], s_rw[5 * j + 4]));
            if (A.hist) {
              const int at = atomicAdd(A.hist_ctl, 1);
              if (at < A.hist_cap) { A.hist[3 * (size_t)at] = idm; A.hist[3 * (size_t)at + 1] = idn; A.hist[3 * (size_t)at + 2] = iter_v; }
              else A.hist_ctl[1] = 1;
            }
          }
        }
      }
      sq_drain();                            // (the item is written: now the counts that make it visible)
      if (lane == 0) {
        if (govf >= 0) wt_i32(g.ovf_cnt, govf + 1);
        wt_i32(g.cnt + cell, gslot + 1);
        if (OPT) atomicAdd(A.tree_cnt + 16 * mine, 1);
      }
      if (govf >= 0 && g.ovf_cnt == A.grid_ovf_src) grid_ovf_now = govf + 1;
    };
    int n_acc = 0;                               // accepted attempts of the step whose rows wait in s_acc
    // ---- publish the control block of the next step; n_pend of its nodes are still to be written (their rows in s_acc)
    auto publish = [&](int n_pend) {
      ++step; ++n_steps;
      const int set_p = (int)(step % (uint32_t)S.n_sets);
      for (int half = 0; half < 2; ++half) {
        const int gidx = 64 * half + lane;
        uint32_t v = 0;
        v = gidx == SPB_CUR ? (uint32_t)(cursor & 0xffffffffULL) : v;
        v = gidx == SPB_CUR + 1 ? (uint32_t)(cursor >> 32) : v;
        v = gidx == SPB_FN ? (uint32_t)fn : v;
        v = gidx == SPB_CN ? (uint32_t)cn : v;
        v = gidx == SPB_NN ? (uint32_t)n_nodes : v;
        v = gidx == SPB_ITER ? (uint32_t)iter : v;
        v = gidx == SPB_EF ? (uint32_t)empty_frontier : v;
        v = gidx == SPB_NE ? (uint32_t)ner : v;
        v = gidx == SPB_FNPC ? (uint32_t)fn_pc : v;
        v = gidx == SPB_NNC ? (uint32_t)nn_c : v;
        v = gidx == SPB_NNS ? (uint32_t)(n_nodes - n_pend) : v;
        if (gidx >= SPB_ER && gidx < SPB_ER + SP_ER_MAX) v = gidx - SPB_ER < ner ? (uint32_t)s_er[gidx - SPB_ER] : 0u;
        if (gidx >= SPB_PEND && gidx < SPB_PEND + 16 * SFFK_SPEC_DEPTH) {
          const int k = (gidx - SPB_PEND) >> 4, j = (gidx - SPB_PEND) & 15;
          if (k < n_pend) v = j < 12 ? s_acc[k][SPG_QP + j] : (j == 12 ? s_acc[k][SPG_MINE] : (j == 13 ? s_acc[k][SPG_BEST] : (j == 14 ? s_acc[k][SPG_BEST + 1] : 1u)));
        }
        wt_u64(S.base + SP_BASE * set_p + gidx, sp_gran(step, v));
      }
      if (lane == 0) wt_i32(S.cur_step, (int)step);
      nn_base = n_nodes;
    };
    bool published = false, step_desync = false;
    while (!terminated && !fault && !stop && waves_done < A.max_waves) {
      if (!published) { publish(0); lap(0); }
      published = false;
      const int set = (int)(step % (uint32_t)S.n_sets);
      lap(0);
      if (grid_ovf_now > A.grid_ovf_limit) { stop = true; }   // (once per step: the list has room for a step's nodes; the leader is the count's only writer)
      int sc = 0;
      for (; !stop;) {   // ---- the waves of this step, in the reference's order
        // what the round engine checks before a round (round_begin_scalars), and what this launch has to leave to the host
        if (n_nodes + 1 > f.node_cap - 8 || nb + TM > f.border_cap) { fault = SFFK_FAULT_CAPACITY; break; }
        if ((unsigned long long)(nb + TM) * 2ULL > f.bt_mask + 1ULL) { fault = SFFK_FAULT_BORDER_TABLE; break; }
        if (cursor + 8ULL + (unsigned long long)(TM * WP) > A.words_end) { stop = true; break; }
        // node selection (src/forest.h:136-151) - tentative until the scenario's records are known to be about this wave.
        // The leader computes the POSITION; the node behind it (and its flags) come with the records, whose worker read
        // the frontier while it could not change.
        const int use_closed = cn > 0 && empty_frontier;
        const int pool = use_closed ? cn : fn;
        if (pool < 1) { terminated = 1; break; }
        unsigned long long cur2 = cursor, rdw = 0;
        int pick;
        do { pick = sq_lemire(f.ring[cur2 & f.ring_mask], (unsigned long long)pool); ++cur2; if (pick < 0) ++rdw; } while (pick < 0);
        const int idx = use_closed ? pick : er_map(pick);
        const bool own = !use_closed && idx >= fn_pc + (nn_base - nn_c);   // (a node of this step: no scenario models that)
        const unsigned long long* rec0 = S.rec + ((size_t)set * S.n_slots + (size_t)sc * TM) * SFFK_SPEC_REC;
        // row 0 of the scenario's records, one granule per lane and attempt; all asked for together
        unsigned long long gr[8];
        auto ready = [&](unsigned long long g) -> bool { return __all(lane >= SPG_USED || (uint32_t)(g >> 32) == step); };
        auto ask_all = [&](int from) {
#pragma unroll
          for (int q = 0; q < 8; ++q) if (q >= from && q < TM) gr[q] = sq_u64(rec0 + (size_t)q * SFFK_SPEC_REC + lane);
        };
        auto fetch = [&](int rd) -> bool {       // attempt rd's row 0 -> s_row (waits for it)
          unsigned long long g = 0;
#pragma unroll
          for (int q = 0; q < 8; ++q) if (q == rd) g = gr[q];
          if (!ready(g)) {
            const unsigned long long* rp = rec0 + (size_t)rd * SFFK_SPEC_REC;
            const unsigned long long t0 = wall_clock64();
            for (int spin = 0;; ++spin) {
              g = sq_u64(rp + lane);
              if (ready(g)) break;
              if ((spin & 63) == 63 && wall_clock64() - t0 > S.timeout_ticks) {
                if (S.hb) {
                  unsigned long long* o = S.hb + (size_t)S.n_sets * S.n_slots;
                  if (lane == 0) { o[0] = step; o[1] = (unsigned long long)sc; o[2] = (unsigned long long)rd; o[3] = (unsigned long long)set; }
                  if (lane < SPG_USED) o[8 + lane] = g;
                }
                return false;
              }
              __builtin_amdgcn_s_sleep(1);
            }
            ask_all(rd + 1);                     // (the others of the wave were written about when this one was)
          }
          SP_WAVE_SYNC();
          s_row[lane] = (uint32_t)g;
          SP_WAVE_SYNC();
          return true;
        };
        bool have0 = false;
        int node = -1, fl_node = 2;
        lap(6);
        if (iter < f.max_iterations) {
          ask_all(0);
          if (!fetch(0)) { stalled = 1; stop = true; break; }
          lap(sc == 0 ? 1 : 7);
          have0 = true;
          const unsigned long long rc = ((unsigned long long)s_row[SPG_CUR + 1] << 32) | (unsigned long long)s_row[SPG_CUR];
          const bool same = !own && s_row[SPG_STATUS] != SPS_INVALID && (int)s_row[SPG_PICK] == idx && (int)s_row[SPG_UCL] == use_closed &&
                            (int)s_row[SPG_ITER] == iter && rc == cur2 && (int)s_row[SPG_NN] == n_nodes;
          if (!same) {                           // the scenario is not what happened: the step ends in front of this wave
            if (sc == 0) fault = SFFK_FAULT_INTERNAL;
            break;
          }
          node = (int)s_row[SPG_NODE]; fl_node = (int)s_row[SPG_FL];
        } else {                                 // (no attempt left: the wave only closes its node)
          node = use_closed ? sq_i32(f.closed + pick) : (idx >= fn_pc ? nn_c + (idx - fn_pc) : sq_i32(frontier + idx));
          fl_node = sq_u8(f.nflag + node);
        }
        cursor = cur2; redraws += rdw; ++waves;
        w_node = node; w_pos = pick; w_closed = use_closed;
        bool failing = true;
        int outcome = TM;
        for (int rd = 0; rd < TM && failing && iter < f.max_iterations; ++rd) {
          if (!(rd == 0 && have0) && !fetch(rd)) {
            stalled = 1; fault = SFFK_FAULT_LISTS; w_round = rd; in_wave = 1;   // (the host finishes the wave)
            break;
          }
          lap(2);
          const int status = (int)s_row[SPG_STATUS];
          if ((int)s_row[SPG_NODE] != node || (int)s_row[SPG_ITER] != iter || (int)s_row[SPG_NN] != n_nodes ||
              (((unsigned long long)s_row[SPG_CUR + 1] << 32) | (unsigned long long)s_row[SPG_CUR]) != cursor) { fault = SFFK_FAULT_INTERNAL; break; }
          if (status == SPS_FAULT) { fault = SFFK_FAULT_LISTS; w_round = rd; in_wave = 1; break; }
          if (status != SPS_REJECT && status != SPS_ACCEPT) { fault = SFFK_FAULT_INTERNAL; break; }
          cursor += (unsigned long long)WP;
          ++iter;
          ++rounds; rnodes += (unsigned long long)(n_nodes + 1); ++rqueries; ++n_commit;
          cc += (unsigned long long)s_row[SPG_CC]; pf += (unsigned long long)s_row[SPG_PF]; nq += (unsigned long long)s_row[SPG_NQ];
          const int mine = (int)s_row[SPG_MINE];
          if (s_row[SPG_EVENT]) {                // :288-294 border entry unless the pair has one
            const int s_id = (int)s_row[SPG_EV_ID], s_tree = (int)s_row[SPG_EV_TREE];
            const int a = s_id < node ? s_id : node, b = s_id < node ? node : s_id;
            const unsigned long long key = ((unsigned long long)(uint32_t)a << 32) | ((unsigned long long)(uint32_t)b + 1ULL);
            size_t h = (size_t)((key * 0x9E3779B97F4A7C15ULL) >> 17) & (size_t)f.bt_mask;
            bool fresh = false;
            // (the table is the leader's alone while the launch runs - written by this wavefront, read through its own CU's
            // caches: plain loads, no round trip to memory per probe)
            for (int guard = 0; guard < (1 << 24); ++guard) {
              const unsigned long long cur = f.bt_key[h];
              if (cur == key) { fresh = f.bt_val[h] == ~0ULL; break; }
              if (cur == 0ULL) { fresh = true; break; }
              h = (h + 1) & (size_t)f.bt_mask;
            }
            if (fresh) {
              if (lane == 0) {
                f.bt_key[h] = key;
                f.bt_val[h] = epoch << 32;
                f.b_n1[nb] = a; f.b_n2[nb] = b;
                f.b_ta[nb] = s_tree < mine ? s_tree : mine; f.b_tb[nb] = s_tree < mine ? mine : s_tree;
                f.b_dist[nb] = sp_f64(s_row[SPG_EV_DIST], s_row[SPG_EV_DIST + 1]);
                f.pair[(size_t)s_tree * R + mine] = 1;
                f.pair[(size_t)mine * R + s_tree] = 1;
              }
              sq_drain();
              ++nb;
            }
          }
          if (status != SPS_ACCEPT) { lap(6); continue; }
          // ---- the new node (:329, :353-367).  SFF*: written now (the step ends with it); plain SFF: its row is kept and written
          // behind the publication of the next step, whose workers get the node from the control block meanwhile
          const int idn = n_nodes;
          const int n_rw = OPT ? (int)s_row[SPG_NRW] : 0;
          if (OPT) {
            ++st_rounds; st_members += (unsigned long long)s_row[SPG_NMEM];
            if (n_rw > 0) {                      // the rewires behind row 0 (written and drained before it)
              const unsigned long long* rp = rec0 + (size_t)rd * SFFK_SPEC_REC + 64;
              for (int q = lane; q < 5 * n_rw; q += 64) s_rw[q] = (uint32_t)sq_u64(rp + q);
            }
            apply_accept(s_row, idn, iter, n_rw);
          } else {
            s_acc[n_acc][lane] = s_row[lane];
            if (lane == 0) { s_acc_id[n_acc] = idn; s_acc_iter[n_acc] = iter; }
            ++n_acc;
            SP_WAVE_SYNC();
          }
          st_rewires += (unsigned long long)n_rw;
          ++n_nodes; ++fn;
          failing = false;
          outcome = rd;
          lap(3);
        }
        if (fault) break;
        // ---- the slot is exhausted: its node leaves the frontier for the closed list (:160-178; the erase keeps the order)
        bool desync = false;
        if (failing && !use_closed) {
          const int fl = fl_node;
          if (fl & 2) {
            if (lane == 0) { wt_u8(f.nflag + node, (uint8_t)((fl & ~2) | 1)); wt_i32(f.closed + cn, node); }
            ++cn;
            er_insert(idx);
            --fn;
          } else desync = true;                  // (the scenarios below assumed the erase)
        }
        step_desync = step_desync || desync;
        // ---- termination (:184-201)
        empty_frontier = fn == 0 ? 1 : 0;
        if (!solved && empty_frontier) {
          unsigned long long reach = 1ULL, frontier_set = 1ULL;
          if (R <= 64) {
            unsigned long long row = 0ULL;   // lane a: bit b = pair (a, b) has a border
            if (lane < R) for (int b2 = 0; b2 < R; ++b2) if (sq_u8(f.pair + (size_t)lane * R + b2)) row |= 1ULL << b2;
            while (frontier_set) {
              const int a = __ffsll((long long)frontier_set) - 1;
              frontier_set &= frontier_set - 1;
              const unsigned long long ra = __shfl(row, a) & ~reach;
              reach |= ra; frontier_set |= ra;
            }
            solved = __popcll(reach) == R ? 1 : 0;
          } else fault = SFFK_FAULT_LISTS;
        }
        const bool budget = f.node_budget > 0 && n_nodes >= f.node_budget;
        terminated = (solved || iter >= f.max_iterations || budget) ? 1 : 0;
        if (A.trace && waves_done < A.trace_cap && lane == 0) {
          int32_t* t = A.trace + 8 * (size_t)waves_done;
          t[0] = node; t[1] = pick; t[2] = iter; t[3] = (int32_t)(cursor & 0x7fffffffULL); t[4] = failing ? TM : 0; t[5] = n_nodes; t[6] = fn; t[7] = cn;
        }
        ++waves_done;
        lap(4);
        if (terminated || fault || desync || waves_done >= A.max_waves) break;
        sc = S.sc_tab[sc * SFFK_SPEC_TAB + 1 + 2 * SFFK_SPEC_DEPTH + outcome];
        if (sc < 0) break;
      }
      // ---- the step is decided.  Plain SFF: the next step is published NOW - its workers evaluate while the accepted nodes
      // of this one are written (they get those nodes from the control block and skip their grid items) - unless something
      // the workers could trip over is due: the array's compaction, the closed-list mode, the launch's end
      const bool more = !terminated && !fault && !stop && waves_done < A.max_waves;
      const bool pipe = !OPT && S.pipeline && more && !step_desync && !empty_frontier && n_acc > 0 && ner + SFFK_SPEC_DEPTH <= SP_ER_MAX;
      if (pipe) { publish(n_acc); published = true; lap(0); }
      for (int k = 0; k < n_acc; ++k) apply_accept(s_acc[k], s_acc_id[k], s_acc_iter[k], 0);
      n_acc = 0;
      if (ner + SFFK_SPEC_DEPTH > SP_ER_MAX) compact();   // (room for a step's erases)
      sq_drain();                                // (everything written, before anybody is told about the step after)
      step_desync = false;
      lap(5);
    }
    // ---- the launch is over: every set's next control block says so; the frontier array as everybody else expects it
    for (int s2 = 0; s2 < S.n_sets; ++s2) { wt_u64(S.base + SP_BASE * s2 + lane, sp_gran(SP_QUIT, 0u)); wt_u64(S.base + SP_BASE * s2 + 64 + lane, sp_gran(SP_QUIT, 0u)); }
    compact();
    if (lane == 0) wt_i32(S.cur_step, -1);
    if (lane == 0) {
      c->n_nodes = n_nodes; c->iter = iter; c->frontier_n = fn; c->closed_n = cn; c->n_borders = nb;
      c->solved = solved; c->empty_frontier = empty_frontier; c->terminated = terminated;
      c->cursor = cursor; c->collide_calls = cc; c->path_free_calls = pf; c->nn_queries = nq;
      c->waves = waves; c->rounds = rounds; c->round_nodes = rnodes; c->round_queries = rqueries;
      c->redraws += (int)redraws;
      c->star_rounds += st_rounds; c->star_passes += st_rounds; c->star_members += st_members; c->star_rewires += st_rewires;
      c->spec_steps += n_steps; c->spec_committed += n_commit; c->spec_stalled = stalled;
      for (int k = 0; k < 8; ++k) c->wprof[k] += ph[k];
      c->n_act = 0; c->app_n = 0; c->compact_from = 0;
      c->grid_ovf = sq_i32(A.grid_ovf_src); c->tgrid_ovf = 0;
      c->fault = fault;
      c->halt = (terminated || fault) ? 1 : 0;
      c->in_wave = in_wave;
      if (in_wave) {   // the state the host engine resumes the wave from: its one slot, still failing, w_round rounds done
        c->round = w_round; c->n_slots = 1; c->use_closed = w_closed; c->act_sel = 0; c->act_cnt = 1;
        f.slot_node[0] = w_node; f.slot_pos[0] = w_pos; f.act_slot[0] = 0;
      } else c->round = 0;
    }
    return;
  }

  // ===================================================================== a worker: one (scenario, attempt) of every step of its set
  if (wv >= 1) {
    // ---- its second wavefront: the neighbour query of the sample the first one has just drawn; SFF*: a third one for the
    // k nearest of the sample's tree
    int seen = 0;
    for (;;) {
      const int sq = lds_ld(&j_seq);
      if (sq < 0) break;
      if (sq == seen) { __builtin_amdgcn_s_sleep(1); continue; }
      double qp[6];
      for (int k = 0; k < 6; ++k) qp[k] = j_qp[k];
      const double pdist = j_pdist;
      const int mine = lds_ld(&j_mine), kk = lds_ld(&j_k), nn0 = lds_ld(&j_nn0), snn = lds_ld(&j_snn);
      const uint32_t step = (uint32_t)lds_ld(&j_step);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      if (lds_ld(&j_seq) != sq) continue;        // (overwritten while it was read: that job is nobody's business any more)
      seen = sq;
      int n_hit = 0;
      if (wv == 1) {
        // the neighbours: exact 6-D ball of radius max(parentDistance, treeDistance) from the cells its box touches (:262-267)
        const double r = pdist > A.dist_tree ? pdist : A.dist_tree;
        const double ri = (r + A.sweep_abs_eps) * (1.0 + 1e-5);
        const float rf = sqrtf((float)(ri * ri) * 1.000001f) * 1.000001f;
        const GridView& g = A.g;
        const float qx = (float)qp[0], qy = (float)qp[1], qz = (float)qp[2];
        const int lx = grid_coord(qx - rf, g.ox, g.inv_cell, g.nx), hx = grid_coord(qx + rf, g.ox, g.inv_cell, g.nx);
        const int ly = grid_coord(qy - rf, g.oy, g.inv_cell, g.ny), hy = grid_coord(qy + rf, g.oy, g.inv_cell, g.ny);
        const int lz = grid_coord(qz - rf, g.oz, g.inv_cell, g.nz), hz = grid_coord(qz + rf, g.oz, g.inv_cell, g.nz);
        const int wx = hx - lx + 1, wy = hy - ly + 1, wz = hz - lz + 1;
        const int total = wx * wy * wz;
        auto take = [&](bool vld, const GridItem* src) {        // one candidate per lane -> the hit list in LDS
          bool h = false;
          double d = 0, p6[6];
          int id = 0, tr = 0;
          if (vld) {
            const unsigned long long* q8 = reinterpret_cast<const unsigned long long*>(src);
            for (int k = 0; k < 6; ++k) p6[k] = __longlong_as_double((long long)sq_u64(q8 + k));
            const unsigned long long it = sq_u64(q8 + 6);
            id = (int)(unsigned)(it & 0xffffffffULL); tr = (int)(unsigned)(it >> 32);
            d = dist6(p6, qp);
            h = d < r && id < nn0;                               // (a node the leader commits while this step runs reaches the attempt through its scenario)
          }
          const unsigned long long hm = __ballot(h);
          if (h) {
            const int at = n_hit + __popcll(hm & ((1ULL << lane) - 1ULL));
            if (at < 64) { h_id[at] = id; h_tree[at] = tr; h_d[at] = d; for (int k = 0; k < 6; ++k) h_pos[6 * at + k] = p6[k]; }
          }
          n_hit += __popcll(hm);
        };
        for (int c0 = 0; c0 < total; c0 += 64) {
          const int ci = c0 + lane;
          int cell = 0, m = 0;
          if (ci < total) {
            const int q1 = ci / wx, q2 = q1 / wy;
            cell = ((lz + q2) * g.ny + (ly + q1 - q2 * wy)) * g.nx + (lx + ci - q1 * wx);
            m = sq_i32(g.cnt + cell);
            if (m > g.bk) m = g.bk;
          }
          int inc = m;
          for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(inc, off);
            if (lane >= off) inc += o;
          }
          const int tot = __shfl(inc, 63);
          for (int base = 0; base < tot; base += 64) {
            const int j = base + lane;
            const int jj = j < tot ? j : tot - 1;
            int lo = 0, hi = 63;
            while (lo < hi) {
              const int mid = (lo + hi) >> 1;
              if (__shfl(inc, mid) > jj) hi = mid; else lo = mid + 1;
            }
            const int src_cell = __shfl(cell, lo);
            const int slot2 = jj - (__shfl(inc, lo) - __shfl(m, lo));
            take(j < tot, g.items + (size_t)src_cell * g.bk + slot2);
          }
        }
        int no = sq_i32(g.ovf_cnt);
        if (no > g.ovf_cap) no = g.ovf_cap;
        for (int base = 0; base < no; base += 64) take(base + lane < no, g.ovf + base + lane);
      }
      if (wv == 1) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) { lds_st(&j_nhit, n_hit); }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) lds_st(&j_done, seen);
      }
      if (OPT && wv == 2) {
        TopK mt{1.0e300, 0x7fffffff};
        int n_mem = 0;
        if (kk <= SFFK_STAR_KMAX && lds_ld(&j_cancel) < seen)
          sq_knn(A.g, qp, mine, kk, sq_i32(A.tree_cnt + 16 * mine), A.cell_edge, A.knn_slack, lane, mt, n_mem, snn, S.cur_step, step,
                 S.hb ? S.hb + (size_t)S.n_sets * S.n_slots + 80 : nullptr, &j_cancel, seen);
        k_d[lane] = mt.d; k_id[lane] = mt.id;
        if (lane == 0) lds_st(&j_nmem, n_mem);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) lds_st(&j_done_k, seen);
      }
    }
    return;
  }
  const int wid = (int)blockIdx.x - 1;
  const int set = wid / S.n_slots, slot = wid - set * S.n_slots;
  const int sc = slot / TM, att = slot - sc * TM;
  const int32_t* T = S.sc_tab + sc * SFFK_SPEC_TAB;
  const int level = T[0];
  int out_[SFFK_SPEC_DEPTH], anc_[SFFK_SPEC_DEPTH];
  for (int l = 0; l < SFFK_SPEC_DEPTH; ++l) { out_[l] = T[1 + l]; anc_[l] = T[1 + SFFK_SPEC_DEPTH + l]; }
  unsigned long long* my_rec = S.rec + ((size_t)set * S.n_slots + slot) * SFFK_SPEC_REC;
  double* rtri = lds_d;
  double* stage = rtri + (size_t)A.rob.n_tri * 9;
  int32_t* ibase = reinterpret_cast<int32_t*>(stage + STAGE_DOUBLES);
  int32_t* stack = ibase;                        // (+ the triangle-grid hash set behind it)
  int32_t* cand = ibase + (STACK_CAP + TG_HASH);
  int32_t* queue = cand + CAND_CAP;
  for (int i = lane; i < A.rob.n_tri * 9; i += 64) rtri[i] = A.rob.tri[i];
  __builtin_amdgcn_wave_barrier();
  fill_robot_boxes(rtri, reinterpret_cast<double*>(queue + QUEUE_CAP), A.rob.n_tri, lane, 64);
  __builtin_amdgcn_wave_barrier();
  uint32_t last = 0;
  unsigned long long ex_pose = 0, ex_seg = 0, ex_smp = 0, evals = 0;
  int myseq = 0;                                 // jobs handed to the second wavefront so far
  for (;;) {
    // ---- the next step of my set
    uint32_t step = 0;
    bool quit = false;
    SP_WAVE_SYNC();
    for (;;) {
      const unsigned long long g = sq_u64(S.base + SP_BASE * set + lane), g2 = sq_u64(S.base + SP_BASE * set + 64 + lane);
      const uint32_t tg = (uint32_t)(g >> 32), tg2 = (uint32_t)(g2 >> 32);
      const uint32_t t0 = (uint32_t)__shfl((int)tg, 0);
      const bool whole = __all(tg == t0 && (64 + lane >= SPB_USED || tg2 == t0));
      if (whole && t0 == SP_QUIT) { quit = true; break; }
      if (whole && t0 > last) {
        step = t0; s_row[lane] = (uint32_t)g;
        // (the pending rows: granules SPB_PEND .. SPB_USED - 1 of the block, 16 per node)
        {   // granule gi of the block -> the pending rows / the erase list
          const int gi0 = lane, gi1 = 64 + lane;
          if (gi0 >= SPB_PEND && gi0 < SPB_ER) s_acc[(gi0 - SPB_PEND) >> 4][(gi0 - SPB_PEND) & 15] = (uint32_t)g;
          if (gi1 >= SPB_PEND && gi1 < SPB_ER) s_acc[(gi1 - SPB_PEND) >> 4][(gi1 - SPB_PEND) & 15] = (uint32_t)g2;
          if (gi0 >= SPB_ER && gi0 < SPB_USED) s_er[gi0 - SPB_ER] = (int)(uint32_t)g;
          if (gi1 >= SPB_ER && gi1 < SPB_USED) s_er[gi1 - SPB_ER] = (int)(uint32_t)g2;
        }
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    if (quit) { if (lane == 0) lds_st(&j_seq, -1); break; }
    SP_WAVE_SYNC();
    last = step;
    const unsigned long long cur0 = ((unsigned long long)s_row[SPB_CUR + 1] << 32) | (unsigned long long)s_row[SPB_CUR];
    const int fn0 = (int)s_row[SPB_FN], cn0 = (int)s_row[SPB_CN], nn0 = (int)s_row[SPB_NN], it0 = (int)s_row[SPB_ITER];
    const int ef0 = (int)s_row[SPB_EF];
    const int ne0 = (int)s_row[SPB_NE], fnpc = (int)s_row[SPB_FNPC], nnc = (int)s_row[SPB_NNC], nns = (int)s_row[SPB_NNS];
    const int n_pend = nn0 - nns;                // nodes the leader is still writing: from the block, never from the store
    if (lane < 6 * SFFK_SPEC_DEPTH) { const int k = lane / 6, q = lane - 6 * k; p_pos[6 * k + q] = sp_f64(s_acc[k][2 * q], s_acc[k][2 * q + 1]); }
    if (lane < SFFK_SPEC_DEPTH) { p_tree[lane] = (int)s_acc[lane][12]; p_best[lane] = sp_f64(s_acc[lane][13], s_acc[lane][14]); }
    SP_WAVE_SYNC();
    // (the leader is past my step; its store of cur_step may become visible after the control block's: never "!=")
    auto stale = [&]() -> bool { return (uint32_t)__builtin_amdgcn_readfirstlane(sq_i32(S.cur_step)) > step; };
    unsigned long long wt0 = S.hb ? wall_clock64() : 0ULL, wlast = wt0, wph[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto beat = [&](int phase) {
      if (!S.hb) return;
      const unsigned long long t = wall_clock64();
      if (phase >= 1 && phase <= 11) wph[phase - 1] = t - wlast;
      wlast = t;
    };
    // ---- my scenario: the waves before mine, with the outcomes it assumes
    unsigned long long cur = cur0;
    int sfn = fn0, scn = cn0, snn = nn0, sit = it0;
    // (s_er: the positions erased since the array was last compacted - from the control block - and, behind them in the same
    // sorted list, the ones my scenario's failed waves erase; ne = mine among them)
    int ne = 0, na = 0, nel = ne0, pslot[SFFK_SPEC_DEPTH];
    for (int l = 0; l < SFFK_SPEC_DEPTH; ++l) pslot[l] = 0;
    auto er_map = [&](int logical) -> int { return sp_er_map(s_er, nel, logical, lane); };
    bool valid = ne0 >= 0 && ne0 <= SP_ER_MAX && n_pend >= 0 && n_pend <= SFFK_SPEC_DEPTH;
    for (int l = 0; l < level && valid; ++l) {
      const int ucl = scn > 0 && ef0;
      const int pool = ucl ? scn : sfn;
      if (pool < 1) { valid = false; break; }
      int pk;
      do { pk = sq_lemire(f.ring[cur & f.ring_mask], (unsigned long long)pool); ++cur; } while (pk < 0);
      const int o = out_[l];
      if (o < TM) {                              // accepted at attempt o: one more node, at the frontier's end
        if (sit + o + 1 > f.max_iterations) { valid = false; break; }
        cur += (unsigned long long)((o + 1) * WP); sit += o + 1;
        pslot[na++] = anc_[l] * TM + o;
        ++sfn; ++snn;
        if (f.node_budget > 0 && snn >= f.node_budget) valid = false;
      } else {                                   // all attempts failed: the node leaves the frontier (order kept)
        if (sit + TM > f.max_iterations) { valid = false; break; }
        cur += (unsigned long long)(TM * WP); sit += TM;
        if (!ucl) {
          if (pk >= fn0 - ne) { valid = false; break; }   // (a node of this step: not modelled)
          const int idx = er_map(pk);
          sp_er_insert(s_er, nel, idx, lane);   // (keeps the list sorted)
          ++ne; --sfn; ++scn;
        }
      }
      if (sit >= f.max_iterations) valid = false;
      if ((sfn == 0) != (ef0 != 0)) valid = false;   // frontier <-> closed-list mode switches end the step
    }
    int node = -1, pick_idx = -1, my_ucl = 0, fl_node = 0;
    if (valid) {
      const int ucl = scn > 0 && ef0;
      const int pool = ucl ? scn : sfn;
      my_ucl = ucl;
      if (pool < 1) valid = false;
      else {
        int pk;
        do { pk = sq_lemire(f.ring[cur & f.ring_mask], (unsigned long long)pool); ++cur; } while (pk < 0);
        if (ucl) { if (pk >= cn0) valid = false; else { pick_idx = pk; node = sq_i32(f.closed + pk); } }
        else if (pk >= fn0 - ne) valid = false;
        else {
          const int idx = er_map(pk);
          pick_idx = idx;
          // (behind the entries the array had when it was last compacted: the nodes created since, in order)
          node = idx >= fnpc ? nnc + (idx - fnpc) : sq_i32(frontier + idx);
        }
      }
    }
    if (OPT && na > 0) valid = false;            // (SFF*: an accepted node's rewires change what a later attempt reads)
    beat(1);
    const unsigned long long cursor_a = cur + (unsigned long long)(att * WP);
    const int iter_a = sit + att;
    int status = SPS_REJECT;
    if (!valid) status = SPS_INVALID;
    else if (iter_a >= f.max_iterations) status = SPS_SKIPPED;
    // ---- the attempt (k_seq_waves's, without its writes)
    unsigned long long cc_l = 0;
    int pf_l = 0, nq_l = 0, event = 0, ev_id = 0, ev_tree = 0, mine = 0, par_new = node, n_rw = 0, n_mem = 0;
    double ev_dist = 0, pdist = 0, best = 0, dcl_new = 0, qp[6] = {0, 0, 0, 0, 0, 0};
    bool aborted = false;
    if (status == SPS_REJECT) {
      double cpos[6], droot_ex;
      if (node >= nns) {                         // (a node of the step before: the leader is just writing it - its row came with the block)
        for (int k = 0; k < 6; ++k) cpos[k] = p_pos[6 * (node - nns) + k];
        mine = p_tree[node - nns]; droot_ex = p_best[node - nns]; fl_node = 2;
      } else {
        for (int k = 0; k < 6; ++k) cpos[k] = sq_f64(A.st.pos + 6 * (size_t)node + k);
        mine = sq_i32(A.st.tree + node);
        droot_ex = sq_f64(f.d_root + node);
        fl_node = sq_u8(f.nflag + node);
      }
      const bool force = (fl_node & 1) != 0;
      uint64_t w[6];
      for (int k = 0; k < 6; ++k) w[k] = k < WP ? f.ring[(cursor_a + k) & f.ring_mask] : 0ULL;
      SampleTrig ht{};
      if (A.trig) {
        const double* t0 = A.trig + 3 * (size_t)(cursor_a & f.ring_mask);
        ht.c_phi = t0[0]; ht.s_phi = t0[1];
        if (WP == 6) {
          const double* t1 = A.trig + 3 * (size_t)((cursor_a + 1) & f.ring_mask);
          const double* t3 = A.trig + 3 * (size_t)((cursor_a + 3) & f.ring_mask);
          ht.c_theta = t1[0]; ht.s_theta = t1[1]; ht.acos_u = t3[2];
        }
      } else {
        const double ang = sample_angle(lane == 1 ? w[1] : w[0]);
        const double sv = sffp::psin(ang), cv = sffp::pcos(ang);
        ht.s_phi = __shfl(sv, 0); ht.c_phi = __shfl(cv, 0);
        ht.s_theta = __shfl(sv, 1); ht.c_theta = __shfl(cv, 1);
        ht.acos_u = WP == 6 ? sffp::pacos(sample_acos_arg(w[3])) : 0.0;
      }
      beat(6);
      const bool ok = sample_point_with(w, cpos, A.sampling_dist, A.dim, A.limits, qp, ht);
      pdist = dist6(cpos, qp);                                     // parentDistance, :250
      best = pdist + droot_ex; dcl_new = pdist;
      ++evals;
      beat(7);
      {   // the early row: what the scenarios that assume this sample accepted need of it
        uint32_t v = 0;
        for (int k = 0; k < 6; ++k) { v = lane == 2 * k ? sp_lo(qp[k]) : v; v = lane == 2 * k + 1 ? sp_hi(qp[k]) : v; }
        v = lane == 12 ? (uint32_t)mine : v;
        v = lane == 13 ? sp_lo(best) : v; v = lane == 14 ? sp_hi(best) : v;
        v = lane == 15 ? (ok ? 1u : 0u) : v;
        if (lane < 16) wt_u64(my_rec + SFFK_SPEC_EARLY + lane, sp_gran(step, v));
      }
      bool flt = false, reject = true;
      int kstar = 0;
      if (OPT) kstar = __popcll(__ballot(lane > 0 && lane <= SFFK_STAR_KMAX + 1 && A.ktab[lane] <= snn));   // (size_t)(2e log10 N), :309
      if (ok) {   // the second wavefront starts on the neighbours now
        if (lane == 0) {
          for (int k = 0; k < 6; ++k) j_qp[k] = qp[k];
          j_pdist = pdist;
          lds_st(&j_mine, mine); lds_st(&j_k, kstar); lds_st(&j_nn0, nns); lds_st(&j_snn, snn); lds_st(&j_step, (int)step);
        }
        ++myseq;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) lds_st(&j_seq, myseq);
      }
      auto helper_wait = [&](const int32_t* word, int want) {   // the other wavefront's answer (every lane reads the word: uniform)
        while (lds_ld(word) < want) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      };
      if (ok) {
        // ---- Environment::Collide(newPoint)
        cc_l += 1; ex_pose += 1;
        bool hit = false;
        if (A.env.n_tri != 0 && !surely_clear(A.env, qp)) {
          double Rm[9], c3[3];
          if (qp[3] == 0 && qp[4] == 0 && qp[5] == 0) { Rm[0] = Rm[4] = Rm[8] = 1; Rm[1] = Rm[2] = Rm[3] = Rm[5] = Rm[6] = Rm[7] = 0; }
          else rotation(qp, Rm);
          xform(Rm, qp, A.rob.center, c3);
          hit = pose_exact(A.env, A.rob, rtri, stack, cand, stage, qp, Rm, c3, lane);
        }
        beat(2);
        if (!hit) {
          // ---- isPathFree(expanded, newPoint)
          pf_l += 1; ex_seg += 1;
          const bool free0 = (sq_edge_clear_fast(A.env, cpos, qp, lane, cc_l, ex_smp) || sq_path_free(A.env, A.rob, rtri, stack, cand, queue, stage, cpos, qp, &s_fh, &s_ovf, lane, cc_l, ex_smp, flt));
          reject = !free0;
          beat(3);
          if (!reject && stale()) aborted = true;
          int n_hit = 0;
          if (!flt && !reject && !aborted) {
            nq_l += R;                                               // :262-267 one radiusSearch per tree
            const double r = pdist > A.dist_tree ? pdist : A.dist_tree;
            helper_wait(&j_done, myseq);                             // (asked for when the sample was drawn)
            n_hit = lds_ld(&j_nhit);
            // ---- the nodes of the step before that the leader is still writing: nodes nns .. nn0 - 1, from the control block
            for (int p = 0; p < n_pend; ++p) {
              double pp[6];
              for (int k = 0; k < 6; ++k) pp[k] = p_pos[6 * p + k];
              const double d = dist6(pp, qp);
              if (d < r) {
                if (lane == 0 && n_hit < 64) {
                  h_id[n_hit] = nns + p; h_tree[n_hit] = p_tree[p]; h_d[n_hit] = d;
                  for (int k = 0; k < 6; ++k) h_pos[6 * n_hit + k] = pp[k];
                }
                ++n_hit;
              }
            }
            // ---- the samples my scenario assumes accepted before me: nodes nn0 .. nn0 + na - 1
            for (int p = 0; p < na && !aborted; ++p) {
              const unsigned long long* ep = S.rec + ((size_t)set * S.n_slots + pslot[p]) * SFFK_SPEC_REC + SFFK_SPEC_EARLY;
              SP_WAVE_SYNC();
              for (int spin = 0;; ++spin) {   // (its worker publishes it as long as the step is the current one)
                const unsigned long long g2 = sq_u64(ep + (lane & 15));
                if (__all((uint32_t)(g2 >> 32) == step)) { s_row[lane] = (uint32_t)g2; break; }
                if ((spin & 15) == 15 && stale()) { aborted = true; break; }
                __builtin_amdgcn_s_sleep(1);
              }
              SP_WAVE_SYNC();
              if (aborted) break;
              if (s_row[15] == 0u) { status = SPS_INVALID; break; }      // (that sample left the limits: my scenario cannot happen)
              double pp[6];
              for (int k = 0; k < 6; ++k) pp[k] = sp_f64(s_row[2 * k], s_row[2 * k + 1]);
              const double pb = sp_f64(s_row[13], s_row[14]);
              const int ptree = (int)s_row[12];
              SP_WAVE_SYNC();
              if (lane == 0) { for (int k = 0; k < 6; ++k) p_pos[6 * (n_pend + p) + k] = pp[k]; p_best[n_pend + p] = pb; p_tree[n_pend + p] = ptree; }
              const double d = dist6(pp, qp);
              if (d < r) {
                if (lane == 0 && n_hit < 64) {
                  h_id[n_hit] = nn0 + p; h_tree[n_hit] = ptree; h_d[n_hit] = d;
                  for (int k = 0; k < 6; ++k) h_pos[6 * n_hit + k] = pp[k];
                }
                ++n_hit;
              }
            }
            if (n_hit > A.hit_cap || n_hit > 64) flt = true;
          }
          beat(4);
          if (!flt && !reject && !aborted && status == SPS_REJECT) {
            // ---- the neighbour loop (:270-300) in the reference's order: tree id, then distance, then id
            SP_WAVE_SYNC();
            const bool have = lane < n_hit;
            const int id = have ? h_id[lane] : 0x7fffffff;
            const int t = have ? h_tree[lane] : 0x7fffffff;
            const double d = have ? h_d[lane] : 0.0;
            const bool same = t == mine;
            const bool qk = have && (same ? (!force && d < pdist - SFFG_TOL) : (d < A.dist_tree - SFFG_TOL));   // :276 / :283
            int rank = 0;
            for (int j = 0; j < n_hit; ++j) {
              const int tj = __shfl(t, j), idj = __shfl(id, j), qj = __shfl((int)qk, j);
              const double dj = __shfl(d, j);
              if (qj && (tj < t || (tj == t && (dj < d || (dj == d && idj < id))))) ++rank;
            }
            const int n_q = __popcll(__ballot(qk));
            for (int rk = 0; rk < n_q && !reject && !flt && !aborted; ++rk) {
              const unsigned long long sel = __ballot(qk && rank == rk);
              const int src = __ffsll((long long)sel) - 1;
              const int s_same = __shfl((int)same, src), s_id = __shfl(id, src), s_tree = __shfl(t, src);
              double np6[6];
              for (int k = 0; k < 6; ++k) np6[k] = h_pos[6 * src + k];
              pf_l += 1; ex_seg += 1;
              if (s_same) {
                const bool fr = (sq_edge_clear_fast(A.env, np6, qp, lane, cc_l, ex_smp) || sq_path_free(A.env, A.rob, rtri, stack, cand, queue, stage, np6, qp, &s_fh, &s_ovf, lane, cc_l, ex_smp, flt));
                if (fr) reject = true;                                 // :276-280 overcrowded
              } else {
                const bool fr = (sq_edge_clear_fast(A.env, cpos, np6, lane, cc_l, ex_smp) || sq_path_free(A.env, A.rob, rtri, stack, cand, queue, stage, cpos, np6, &s_fh, &s_ovf, lane, cc_l, ex_smp, flt));
                if (fr && !flt) {                                      // :288-294 border entry (the leader knows whether the pair has one)
                  event = 1; ev_id = s_id; ev_tree = s_tree;
                  const double dr = s_id >= nns ? p_best[s_id - nns] : sq_f64(f.d_root + s_id);
                  ev_dist = dr + droot_ex + dist6(np6, cpos);
                }
                reject = true;                                         // :296-299
              }
              if (!reject && stale()) aborted = true;
            }
          }
          beat(5);
          if (OPT && (reject || flt || aborted || status != SPS_REJECT) && lane == 0) lds_st(&j_cancel, myseq);   // (no k nearest needed)
          if (status == SPS_REJECT && !aborted) {
            if (flt) status = SPS_FAULT;
            else if (!reject) {
              status = SPS_ACCEPT;
              if (OPT) {
                // ---- SFF* (:307-351): the k nearest of the tree, choose parent, rewire - each edge checked when its turn comes
                if (kstar > SFFK_STAR_KMAX) flt = true;
                else {
                  TopK mt{1.0e300, 0x7fffffff};
                  double m_droot = 0;
                  nq_l += 1;                                                                        // :317 knnSearch
                  beat(9);
                  helper_wait(&j_done_k, myseq);                                                     // (the third wavefront's, beside everything above)
                  mt.d = k_d[lane]; mt.id = k_id[lane];
                  n_mem = lds_ld(&j_nmem);
                  beat(10);
                  if (stale()) aborted = true;
                  // (every member's cost and position in one round trip, lane = member: a candidate's position is a shuffle away)
                  double m_pos[6] = {0, 0, 0, 0, 0, 0};
                  if (lane < n_mem) {
                    m_droot = sq_f64(f.d_root + mt.id);
                    for (int q = 0; q < 6; ++q) m_pos[q] = sq_f64(A.st.pos + 6 * (size_t)mt.id + q);
                  }
                  // (:320-327 walks the members in order and checks an edge when its cost beats the best so far.  Only members
                  // that beat the cost through the expansion's parent can ever be checked: their positions and clearance words
                  // are fetched three edges at a time, the walk itself stays the reference's)
                  {
                    const double nd_l = mt.d + m_droot;
                    unsigned long long cm = __ballot(lane < n_mem && nd_l < best - SFFG_TOL);
                    while (cm && !flt && !aborted) {
                      int ix[3] = {0, 0, 0};
                      int nb = 0;
                      unsigned long long t = cm;
#pragma unroll
                      for (int e = 0; e < 3; ++e)
                        if (t) { ix[e] = __ffsll((long long)t) - 1; t &= t - 1; nb = e + 1; }
                      double mp3[3][6], qp3[3][6];
                      int id3[3];
#pragma unroll
                      for (int e = 0; e < 3; ++e) {
                        id3[e] = __shfl(mt.id, ix[e]);
                        for (int q = 0; q < 6; ++q) { mp3[e][q] = __shfl(m_pos[q], ix[e]); qp3[e][q] = qp[q]; }
                      }
                      bool clr[3];
                      int ns3[3];
                      sq_edges_clear_probe(A.env, qp3, mp3, nb, lane, clr, ns3);
#pragma unroll
                      for (int e = 0; e < 3; ++e) {
                        if (e >= nb || flt || aborted) continue;
                        const double nd = __shfl(nd_l, ix[e]);
                        if (!(nd < best - SFFG_TOL)) continue;
                        pf_l += 1; ex_seg += 1;
                        bool fr;
                        if (clr[e]) { cc_l += (unsigned long long)ns3[e]; ex_smp += (unsigned long long)ns3[e]; fr = true; }
                        else fr = sq_edge_clear_fast(A.env, qp, mp3[e], lane, cc_l, ex_smp) || sq_path_free(A.env, A.rob, rtri, stack, cand, queue, stage, qp, mp3[e], &s_fh, &s_ovf, lane, cc_l, ex_smp, flt);
                        if (fr && !flt) { best = nd; par_new = id3[e]; dcl_new = __shfl(mt.d, ix[e]); }
                      }
                      cm = t & __ballot(lane < n_mem && nd_l < best - SFFG_TOL);
                    }
                  }
                  // rewire (:332-350): a member the new node's cost improves, if the edge member -> new is free
                  SP_WAVE_SYNC();
                  beat(11);
                  if (stale()) aborted = true;
                  // (every member the new node's cost improves is checked, whatever the others answer: three edges at a time)
                  if (!flt && !aborted) {
                    unsigned long long cm = __ballot(lane < n_mem && best + mt.d < m_droot - SFFG_TOL);
                    while (cm) {
                      int ix[3] = {0, 0, 0};
                      int nb = 0;
#pragma unroll
                      for (int e = 0; e < 3; ++e)
                        if (cm) { ix[e] = __ffsll((long long)cm) - 1; cm &= cm - 1; nb = e + 1; }
                      double mp3[3][6], qp3[3][6];
                      int id3[3];
#pragma unroll
                      for (int e = 0; e < 3; ++e) {
                        id3[e] = __shfl(mt.id, ix[e]);
                        for (int q = 0; q < 6; ++q) { mp3[e][q] = __shfl(m_pos[q], ix[e]); qp3[e][q] = qp[q]; }
                      }
                      bool clr[3];
                      int ns3[3];
                      sq_edges_clear_probe(A.env, mp3, qp3, nb, lane, clr, ns3);
#pragma unroll
                      for (int e = 0; e < 3; ++e) {
                        if (e >= nb) continue;
                        const double dm = __shfl(mt.d, ix[e]);
                        const double proposed = best + dm;
                        pf_l += 1; ex_seg += 1;
                        bool f2 = false;
                        bool fr;
                        if (clr[e]) { cc_l += (unsigned long long)ns3[e]; ex_smp += (unsigned long long)ns3[e]; fr = true; }
                        else fr = sq_edge_clear_fast(A.env, mp3[e], qp, lane, cc_l, ex_smp) || sq_path_free(A.env, A.rob, rtri, stack, cand, queue, stage, mp3[e], qp, &s_fh, &s_ovf, lane, cc_l, ex_smp, f2);
                        if (fr) {
                          if (lane == 0) {
                            s_rw[5 * n_rw] = (uint32_t)id3[e];
                            s_rw[5 * n_rw + 1] = sp_lo(dm); s_rw[5 * n_rw + 2] = sp_hi(dm);
                            s_rw[5 * n_rw + 3] = sp_lo(proposed); s_rw[5 * n_rw + 4] = sp_hi(proposed);
                          }
                          ++n_rw;
                        }
                      }
                    }
                  }
                }
                if (flt) status = SPS_FAULT;
              }
            }
          }
        }
      }
    }
    if (OPT && (status != SPS_ACCEPT || aborted) && lane == 0) lds_st(&j_cancel, myseq);   // (the second wavefront need not finish the k nearest)
    if (aborted) continue;                       // (the leader has moved on: nobody reads this record)
    if (S.test_stall && (int)step == (S.test_stall >> 3) && slot == (S.test_stall & 7)) continue;   // (tests: the leader's time-out path)
    beat(8);
    if (S.hb && status == SPS_ACCEPT && lane == 0) {   // where an ACCEPTED attempt's time went (debugging)
      unsigned long long* o = S.hb + (size_t)S.n_sets * S.n_slots + 64;
      for (int k = 0; k < 11; ++k) atomicAdd(o + k + 1, wph[k]);
      atomicAdd(o, 1ULL);
    }
    // ---- the record: SFF*'s rewires first, drained, then row 0
    if (OPT && status == SPS_ACCEPT && n_rw > 0) {
      SP_WAVE_SYNC();
      for (int q = lane; q < 5 * n_rw; q += 64) wt_u64(my_rec + 64 + q, sp_gran(step, s_rw[q]));
      sq_drain();
    }
    {
      uint32_t v = 0;
      v = lane == SPG_STATUS ? (uint32_t)status : v;
      v = lane == SPG_NODE ? (uint32_t)node : v;
      v = lane == SPG_ITER ? (uint32_t)iter_a : v;
      v = lane == SPG_CUR ? (uint32_t)(cursor_a & 0xffffffffULL) : v;
      v = lane == SPG_CUR + 1 ? (uint32_t)(cursor_a >> 32) : v;
      v = lane == SPG_NN ? (uint32_t)snn : v;
      v = lane == SPG_CC ? (uint32_t)cc_l : v;
      v = lane == SPG_PF ? (uint32_t)pf_l : v;
      v = lane == SPG_NQ ? (uint32_t)nq_l : v;
      v = lane == SPG_EVENT ? (uint32_t)event : v;
      v = lane == SPG_EV_ID ? (uint32_t)ev_id : v;
      v = lane == SPG_EV_TREE ? (uint32_t)ev_tree : v;
      v = lane == SPG_EV_DIST ? sp_lo(ev_dist) : v;
      v = lane == SPG_EV_DIST + 1 ? sp_hi(ev_dist) : v;
      v = lane == SPG_MINE ? (uint32_t)mine : v;
      for (int k = 0; k < 6; ++k) { v = lane == SPG_QP + 2 * k ? sp_lo(qp[k]) : v; v = lane == SPG_QP + 2 * k + 1 ? sp_hi(qp[k]) : v; }
      v = lane == SPG_BEST ? sp_lo(best) : v;
      v = lane == SPG_BEST + 1 ? sp_hi(best) : v;
      v = lane == SPG_PAR ? (uint32_t)par_new : v;
      v = lane == SPG_DCL ? sp_lo(dcl_new) : v;
      v = lane == SPG_DCL + 1 ? sp_hi(dcl_new) : v;
      v = lane == SPG_NRW ? (uint32_t)n_rw : v;
      v = lane == SPG_NMEM ? (uint32_t)n_mem : v;
      v = lane == SPG_PICK ? (uint32_t)pick_idx : v;
      v = lane == SPG_UCL ? (uint32_t)my_ucl : v;
      v = lane == SPG_FL ? (uint32_t)fl_node : v;
      wt_u64(my_rec + lane, sp_gran(step, v));
    }
  }
  if (lane == 0) {
    if (ex_pose) atomicAdd(&c->poses_executed, ex_pose);
    if (ex_seg) atomicAdd(&c->segments_executed, ex_seg);
    if (ex_smp) atomicAdd(&c->samples_executed, ex_smp);
    if (evals) atomicAdd(&c->spec_evaluated, evals);
  }
}

// cos / sin of every engine word as an angle, acos of it as the pitch draw (src/randGen.h:78-100) with the portable routines the
// sampling code itself calls - the same bits - computed once per word by the whole chip instead of by the one wavefront
// that is waited for (the table the libm parity mode fills on the host, DevRound::trig)
__global__ __launch_bounds__(256) void k_ring_trig(const uint64_t* __restrict__ words, double* __restrict__ trig, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t w = words[i];
  const double ang = sample_angle(w);
  trig[3 * (size_t)i] = sffp::pcos(ang);
  trig[3 * (size_t)i + 1] = sffp::psin(ang);
  trig[3 * (size_t)i + 2] = sffp::pacos(sample_acos_arg(w));
}
void launch_ring_trig(hipStream_t s, const uint64_t* words, double* trig, int n) {
  if (n > 0) hipLaunchKernelGGL(k_ring_trig, dim3((n + 255) / 256), dim3(256), 0, s, words, trig, n);
}

// RRT session (csrc/rrt.cpp): nearest node -> steered point, on the device, so that the pose check, the parent edge and the
// k nearest of the new point follow in the same enqueued chain (src/rrt.h:143-166).  q1: the steering targets, idx1 their
// nearest nodes (k1 per query, the first one is used); writes the nearest node's position (a6), the new point (np6) and,
// q2 != null, the k-nearest query of the new point.
__global__ __launch_bounds__(256) void k_rrt_steer(const KnnQuery* __restrict__ q1, const int32_t* __restrict__ idx1, int k1,
                                                   const double* __restrict__ store_pos, double dist, double* __restrict__ a6,
                                                   double* __restrict__ np6, KnnQuery* __restrict__ q2, int kmax, int n,
                                                   SweepQuery* __restrict__ sq, double sq_r, float sq_r2f,
                                                   double* __restrict__ np_copy, const int32_t* __restrict__ alt_slot,
                                                   const int32_t* __restrict__ alt_mate, int row0, int32_t* __restrict__ seg_ns,
                                                   int32_t* __restrict__ conn_cnt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // alt_slot != null: row i is the REPAIRED version of slot alt_slot[i] - the wave's new point alt_mate[i] (a row of np6)
  // would be its nearest node if it becomes a node - and is written as row row0 + i of np6
  // (alt_slot[i] < 0: an unused row of a list built on the device - a degenerate copy of row 0, nothing to check)
  const bool dummy = alt_slot && alt_slot[i] < 0;
  const int slot = alt_slot && !dummy ? alt_slot[i] : (dummy ? 0 : i);
  double a[6], t[6], o[6];
  if (alt_slot) {
    const int m = dummy ? 0 : alt_mate[i];
    for (int k = 0; k < 6; ++k) a[k] = np6[6 * (size_t)m + k];
  } else {
    const int near = idx1[(size_t)i * k1];
    for (int k = 0; k < 6; ++k) a[k] = store_pos[6 * (size_t)near + k];
  }
  for (int k = 0; k < 6; ++k) t[k] = q1[slot].pos[k];
  if (dummy) { for (int k = 0; k < 6; ++k) o[k] = a[k]; }
  else steer(a, t, dist, o);
  for (int k = 0; k < 6; ++k) { a6[6 * (size_t)i + k] = a[k]; np6[6 * (size_t)(row0 + i) + k] = o[k]; np_copy[6 * (size_t)i + k] = o[k]; }
  // the parent edge's sample count and result presets (k_seg_prepare) and the edge kernels' control words (seg_ns: n sample
  // counts | n first hits | n overflow marks | 16 control words), so that the chain needs no launch of its own for them
  seg_ns[i] = edge_samples(edge_parts(a, o));
  seg_ns[(size_t)n + i] = 0x7fffffff;
  seg_ns[2 * (size_t)n + i] = 0;
  if (i < 16) seg_ns[3 * (size_t)n + i] = 0;
  if (n < 16 && i == 0)
    for (int k = n; k < 16; ++k) seg_ns[3 * (size_t)n + k] = 0;
  if (conn_cnt) conn_cnt[i] = 0;
  if (q2) {
    KnnQuery q;
    for (int k = 0; k < 6; ++k) q.pos[k] = o[k];
    q.tree = q1[slot].tree; q.max_id = 0x7fffffff; q.k = kmax; q.mate_base = 0x7fffffff; q.whole_tree = 0; q.pad_ = 0;
    q2[i] = q;
  }
  if (sq) {   // the nodes of the OTHER trees within sq_r of the new point (src/rrt.h:228-231), same fields as Ctx::sweep_lists writes
    SweepQuery Q;
    Q.x = (float)o[0]; Q.y = (float)o[1]; Q.z = (float)o[2]; Q.yaw = (float)o[3]; Q.pitch = (float)o[4]; Q.roll = (float)o[5];
    Q.r2f = sq_r2f; Q.tree = -2 - q1[slot].tree; Q.r = sq_r; Q.max_id = 0x7fffffff; Q.active = 1; Q.pad = Q.pad2 = 0;
    sq[i] = Q;
  }
}
void launch_rrt_steer(hipStream_t s, const KnnQuery* q1, const int32_t* idx1, int k1, const double* store_pos, double dist,
                      double* a6, double* np6, KnnQuery* q2, int kmax, int n, SweepQuery* sq, double sq_r, float sq_r2f,
                      double* np_copy, int32_t* seg_ns, int32_t* conn_cnt, const int32_t* alt_slot, const int32_t* alt_mate, int row0) {
  if (n > 0) hipLaunchKernelGGL(k_rrt_steer, dim3((n + 255) / 256), dim3(256), 0, s, q1, idx1, k1, store_pos, dist, a6, np6, q2, kmax, n,
                                sq, sq_r, sq_r2f, np_copy, alt_slot, alt_mate, row0, seg_ns, conn_cnt);
}

// RRT session: which EARLIER new point of the wave would be slot j's nearest node?  (The replay of Rrt::run_wave cuts the
// wave at the first such slot unless the repaired version of the slot was evaluated too, k_rrt_steer's alt rows.)
// One wavefront per slot: the nearest - ties: the oldest - of the earlier new points of the same tree whose pose is free and
// whose parent edge is free (or not known yet: the candidate list ran over) and that are strictly nearer to the steering target
// than the nearest node of the frozen tree (near_d, k1 per slot).  mate[j] = that slot or -1.
__global__ __launch_bounds__(256) void k_rrt_mates(const KnnQuery* __restrict__ q1, const double* __restrict__ near_d, int k1,
                                                   const double* __restrict__ np6, const uint8_t* __restrict__ hit,
                                                   const int32_t* __restrict__ fh, const int32_t* __restrict__ ov, int n,
                                                   int32_t* __restrict__ mate) {
  const int j = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (j >= n) return;
  double t[6];
  for (int k = 0; k < 6; ++k) t[k] = q1[j].pos[k];
  const int tree = q1[j].tree;
  const double dn = near_d[(size_t)j * k1];
  double best = dn;
  int bi = 0x7fffffff;
  for (int i = lane; i < j; i += 64) {
    if (hit[i] || (fh[i] != 0x7fffffff && !ov[i]) || q1[i].tree != tree) continue;
    const double x = np6[6 * (size_t)i];
    if (!(fabs(x - t[0]) < dn)) continue;
    double p[6];
    for (int k = 0; k < 6; ++k) p[k] = np6[6 * (size_t)i + k];
    const double d = dist6(t, p);
    if (d < best) { best = d; bi = i; }   // (ascending i per lane: the oldest of equal distances stays)
  }
  for (int off = 32; off; off >>= 1) {
    const double ob = __shfl_xor(best, off);
    const int oi = __shfl_xor(bi, off);
    if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if (lane == 0) mate[j] = bi == 0x7fffffff ? -1 : bi;
}
// the slots that have a mate, in slot order: alt_slot / alt_mate (cap entries, the unused ones -1 / 0), cnt[0] = listed (<= cap),
// cnt[1] = found.  One workgroup (a wave holds at most 4 096 slots): per-thread counts, block prefix sum.
__global__ __launch_bounds__(1024) void k_rrt_alt_list(const int32_t* __restrict__ mate, int n, int cap, int32_t* __restrict__ alt_slot,
                                                       int32_t* __restrict__ alt_mate, int32_t* __restrict__ cnt) {
  __shared__ int s_sum[1024];
  const int t = threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int j0 = t * per, j1 = j0 + per < n ? j0 + per : n;
  int own = 0;
  for (int j = j0; j < j1; ++j) own += mate[j] >= 0 ? 1 : 0;
  s_sum[t] = own;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int v = t >= off ? s_sum[t - off] : 0;
    __syncthreads();
    s_sum[t] += v;
    __syncthreads();
  }
  const int total = s_sum[1023];
  int at = s_sum[t] - own;
  for (int j = j0; j < j1; ++j)
    if (mate[j] >= 0) {
      if (at < cap) { alt_slot[at] = j; alt_mate[at] = mate[j]; }
      ++at;
    }
  for (int r = total + t; r < cap; r += 1024) { alt_slot[r] = -1; alt_mate[r] = 0; }
  if (t == 0) { cnt[0] = total < cap ? total : cap; cnt[1] = total; }
}
void launch_rrt_alt_list(hipStream_t s, const int32_t* mate, int n, int cap, int32_t* alt_slot, int32_t* alt_mate, int32_t* cnt) {
  hipLaunchKernelGGL(k_rrt_alt_list, dim3(1), dim3(1024), 0, s, mate, n, cap, alt_slot, alt_mate, cnt);
}
void launch_rrt_mates(hipStream_t s, const KnnQuery* q1, const double* near_d, int k1, const double* np6, const uint8_t* hit,
                      const int32_t* fh, const int32_t* ov, int n, int32_t* mate) {
  if (n > 0) hipLaunchKernelGGL(k_rrt_mates, dim3((n + 3) / 4), dim3(256), 0, s, q1, near_d, k1, np6, hit, fh, ov, n, mate);
}

void launch_spec_waves(hipStream_t s, const SpecArgs& a) {
  const size_t lds = collide_lds_bytes(a.q.rob.n_tri, 1);
  const unsigned grid = 1u + (unsigned)(a.n_sets * a.n_slots);
  if (a.q.optimize) {
    if (lds > 32 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_spec_waves<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_spec_waves<true>, dim3(grid), dim3(192), lds, s, a);
  } else {
    if (lds > 32 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_spec_waves<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_spec_waves<false>, dim3(grid), dim3(128), lds, s, a);
  }
}

// ------------------------------------------------------------------ node store writes
// Writes n positions into the SoA store at [base, base+n): the same double->float cast the
// reference applies when it fills FLANN matrices (src/forest.h:258-260).  Inactive entries are
// written as NaN so that no query can match them.
__global__ __launch_bounds__(256) void k_store_write(NodeStoreMut st, const double* __restrict__ pos6,
                                                     const int32_t* __restrict__ tree, const int32_t* __restrict__ parent,
                                                     const uint8_t* __restrict__ active, int n, int base, GridView g,
                                                     int with_grid) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool on = active ? active[i] != 0 : true;
  const float nanv = __int_as_float(0x7fc00000);
  const size_t o = (size_t)base + i;
  double p[6];
  for (int k = 0; k < 6; ++k) p[k] = pos6[6 * (size_t)i + k];
  st.x[o] = on ? (float)p[0] : nanv;
  st.y[o] = on ? (float)p[1] : nanv;
  st.z[o] = on ? (float)p[2] : nanv;
  st.yaw[o] = on ? (float)p[3] : nanv;
  st.pitch[o] = on ? (float)p[4] : nanv;
  st.roll[o] = on ? (float)p[5] : nanv;
  for (int k = 0; k < 6; ++k) st.pos[6 * o + k] = p[k];
  const int tr = tree ? tree[i] : st.tree[parent[i]];
  st.tree[o] = tr;
  if (with_grid && on) {   // permanent nodes also enter the neighbour grid (replaces a separate k_grid_insert launch)
    GridItem it;
    for (int k = 0; k < 6; ++k) it.p[k] = p[k];
    it.id = (int)o;
    it.tree = tr;
    it.pad[0] = it.pad[1] = 0;
    grid_put(g, it);
  }
}

// ------------------------------------------------------------------ launchers
size_t collide_lds_bytes(int n_robot_tri, int waves) {
  return (size_t)n_robot_tri * 9 * sizeof(double) + (size_t)waves * STAGE_DOUBLES * sizeof(double) +
         (size_t)waves * (STACK_CAP + CAND_CAP + QUEUE_CAP + TG_HASH) * sizeof(int32_t) +
         (size_t)n_robot_tri * 6 * sizeof(double);   // + the robot triangles' boxes (edge samples carry no rotation)
}

void launch_sample_steer(hipStream_t s, const uint64_t* words, const int32_t* parent, const double* node_pos,
                         const double* center_in, int n, double dist, int dim, const SampleParams& prm, double* out6,
                         uint8_t* in_lim, double* parent_dist, SweepQuery* queries, int32_t q_max_base,
                         const RoundTemps& tmp, const DevRound* dev) {
  if (n <= 0) return;
  // (64-thread workgroups: a round has a few thousand samples, this spreads them over the whole chip)
  SampleLaunch P{words, parent, node_pos, center_in, n, dist, dim, prm, out6, in_lim, parent_dist, queries, q_max_base, tmp,
                 dev ? *dev : DevRound{}};
  hipLaunchKernelGGL(k_sample_steer, dim3((n + 63) / 64), dim3(64), 0, s, P);
}
void launch_sample_steer(hipStream_t s, const SampleLaunch& P) {
  if (P.n <= 0) return;
  hipLaunchKernelGGL(k_sample_steer, dim3((P.n + 63) / 64), dim3(64), 0, s, P);
}

void launch_store_write(hipStream_t s, const NodeStoreMut& st, const double* pos6, const int32_t* tree,
                        const int32_t* parent, const uint8_t* active, int n, int base, const GridView* grid) {
  if (n <= 0) return;
  GridView g{};
  if (grid) g = *grid;
  hipLaunchKernelGGL(k_store_write, dim3((n + 255) / 256), dim3(256), 0, s, st, pos6, tree, parent, active, n, base, g,
                     grid ? 1 : 0);
}

void launch_sweep(hipStream_t s, const NodeStoreView& st, int first, int n_nodes, const SweepQuery* queries,
                  const double* qpos, int nq, int32_t* cnt, int32_t* hit_idx, double* hit_dist, int cap) {
  if (n_nodes <= 0 || nq <= 0) return;
  int n4 = (n_nodes + 3) / 4;
  int blocks = (n4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  // aim at >= ~2048 workgroups (8 per CU) but keep >= 4 queries per slice so that a node tile
  // loaded into registers is reused
  int qsplit = (2048 + blocks - 1) / blocks;
  int max_split = (nq + 3) / 4;
  if (qsplit > max_split) qsplit = max_split;
  if (qsplit < 1) qsplit = 1;
  int q_per_block = (nq + qsplit - 1) / qsplit;
  qsplit = (nq + q_per_block - 1) / q_per_block;
  hipLaunchKernelGGL(k_sweep, dim3(blocks, qsplit), dim3(256), 0, s, st, first, n_nodes, queries, qpos, nq, q_per_block, cnt,
                     hit_idx, hit_dist, cap);
}

void launch_grid_insert(hipStream_t s, const GridView& g, const NodeStoreView& st, int first, int n) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_grid_insert, dim3((n + 255) / 256), dim3(256), 0, s, g, st, first, n);
}
void launch_grid_query(hipStream_t s, const GridView& g, const GridView* tg, const NodeStoreView& st,
                       const SweepQuery* queries, const double* qpos, int nq, int32_t* cnt, int32_t* hit_idx,
                       double* hit_dist, int cap, const int32_t* dev_n) {
  if (nq <= 0) return;
  GridView none{};
  hipLaunchKernelGGL(k_grid_query, dim3((nq + 7) / 8), dim3(256), 0, s, g, tg ? *tg : none, st, queries, qpos, nq, cnt,
                     hit_idx, hit_dist, cap, dev_n);
}
void launch_knn_linear(hipStream_t s, const NodeStoreView& st, int n_store, const KnnQuery* q, int nq, int kcap,
                       int32_t* idx, double* dist, int32_t* cnt, double abs_eps) {
  if (nq <= 0) return;
  if (nq <= 2048) {   // a few queries: a workgroup each (k_knn_grid_wg's sweep: the same keys, sixteen times the loads in flight)
    GridView none{};
    hipLaunchKernelGGL(k_knn_grid_wg, dim3(nq), dim3(256), 0, s, none, st, q, nq, kcap, idx, dist, cnt, 1.0, abs_eps, n_store, 1);
    return;
  }
  hipLaunchKernelGGL(k_knn_linear, dim3((nq + 3) / 4), dim3(256), 0, s, st, n_store, q, nq, kcap, idx, dist, cnt, abs_eps);
}
void launch_knn_grid(hipStream_t s, const GridView& g, const GridView* tg, const NodeStoreView& st, const KnnQuery* q, int nq,
                     int kcap, int32_t* idx, double* dist, int32_t* cnt, int32_t* mate_idx, int32_t* mate_cnt, double cell,
                     double slack, int mate_cap, int n_store) {
  if (nq <= 0) return;
  GridView none{};
  if (!tg && !mate_idx && !mate_cnt && nq <= 2048) {   // (more queries than that fill the chip with a wavefront each)
    hipLaunchKernelGGL(k_knn_grid_wg, dim3(nq), dim3(256), 0, s, g, st, q, nq, kcap, idx, dist, cnt, cell, slack, n_store, 0);
    return;
  }
  hipLaunchKernelGGL(k_knn_grid, dim3((nq + 3) / 4), dim3(256), 0, s, g, tg ? *tg : none, st, q, nq, kcap, idx, dist, cnt,
                     mate_idx, mate_cnt, cell, slack, mate_cap, n_store);
}
void launch_set_tree(hipStream_t s, int32_t* tree_col, const int32_t* ids, int n, int32_t value) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_set_tree, dim3((n + 255) / 256), dim3(256), 0, s, tree_col, ids, n, value);
}

void launch_collide_poses(hipStream_t s, const EnvView& env, const RobotView& rob, const double* pos6, int n,
                          const int32_t* live_flags, uint8_t* hit, bool explicit_rt) {
  if (n <= 0) return;
  size_t lds = collide_lds_bytes(rob.n_tri, POSE_WAVES);
  if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_collide_poses), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k_collide_poses, dim3((n + POSE_WAVES - 1) / POSE_WAVES), dim3(64 * POSE_WAVES), lds, s, env,
                     rob, pos6, n, live_flags, hit, explicit_rt ? 1 : 0);
}

#ifdef SFFK_DEBUG_COUNTERS
void debug_counters(unsigned long long* out16) { (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_dbg), sizeof(unsigned long long) * 16); }
void debug_counters_query(unsigned long long* out16) { (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_dbg_q), sizeof(unsigned long long) * 16); }
#endif
#ifdef SFFK_CI_TRACE
void debug_ci_trace(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ci_trace), sizeof(unsigned long long) * 4096 * 8); }
#endif

void launch_clear_build(hipStream_t s, const EnvView& env, const ClearBuildArgs& P, uint32_t* bits_pose, uint32_t* bits_edge,
                        long long n_cells) {
  // every cell starts clear (padding words included: no lookup reaches them); the triangles clear what they block
  const size_t bytes = (size_t)((n_cells + 255) / 256 * 256 / 8);
  (void)hipMemsetAsync(bits_pose, 0xff, bytes, s);
  (void)hipMemsetAsync(bits_edge, 0xff, bytes, s);
  const unsigned split = env.n_tri < 4096 ? 32u : (env.n_tri < 65536 ? 4u : 1u);
  if (env.n_tri > 0) hipLaunchKernelGGL(k_clear_scatter, dim3((unsigned)env.n_tri, split), dim3(256), 0, s, env, P, bits_pose, bits_edge);
}

void launch_settle(hipStream_t s, const SettleArgs& a) {
  if (a.n <= 0) return;
  hipLaunchKernelGGL(k_settle, dim3((a.n + 255) / 256), dim3(256), 0, s, a);
}

// k_query_block serves forests whose buckets are shallow and whose samples see few neighbours (24 exact hits per sample:
// Forest::query_wide); SFFGPU_QUERY=wide / block overrides the choice
bool query_block_mode(const GridView& g, const GridView* tg, const ClassifyArgs& a, const EnvView* env) {
  static const char* const knob = getenv("SFFGPU_QUERY");
  if (!env || !g.lite || !g.ovf_lite || !a.qrec || a.nbcap > 16) return false;
  if (a.wide) return false;
  if (tg && tg->cnt && (!tg->lite || !tg->ovf_lite)) return false;
  if (knob && !strcmp(knob, "wide")) return false;
  // (the kernel's bucket work list packs a record index into 27 bits)
  if ((long long)g.nx * g.ny * g.nz * g.bk >= (1LL << 27)) return false;
  if (tg && tg->cnt && (long long)tg->nx * tg->ny * tg->nz * tg->bk >= (1LL << 27)) return false;
  if (knob && !strcmp(knob, "block")) return true;
  return g.bk <= 8;
}
bool launch_query_classify(hipStream_t s, const GridView& g, const GridView* tg, const NodeStoreView& st,
                           const SweepQuery* queries, const ClassifyArgs& a, const EnvView* env) {
  if (a.n <= 0) return false;
  GridView none{};
  ClassifyArgs aa = a;
  if (env) {   // tests shrink the survivor list to drive the exact kernel's table-scan path
    const int cap_override = getenv("SFFGPU_SEG_LISTCAP") ? atoi(getenv("SFFGPU_SEG_LISTCAP")) : -1;
    if (cap_override >= 0 && cap_override < aa.items_cap) aa.items_cap = cap_override;
  }
  if (query_block_mode(g, tg, a, env)) {
    // (the ordered walk: per XCD ceil(perx / 8) sets of 8 sub-ranges x 64 entry numbers, perx = sub-ranges per XCD)
    const int perx_max = (((a.n + 63) / 64) + 7) / 8;
    const int by_sub = 8 * 64 * ((perx_max + 7) / 8);
    const int by_n = (a.n + QB_S - 1) / QB_S;
    hipLaunchKernelGGL(k_query_block, dim3(a.ord_valid && by_sub > by_n ? by_sub : by_n), dim3(256), 0, s, g, tg ? *tg : none, queries, aa, *env);
    return true;
  }
  hipLaunchKernelGGL(k_query_classify, dim3((a.n + QC_WAVES - 1) / QC_WAVES), dim3(64 * QC_WAVES), 0, s, g, tg ? *tg : none, st,
                     queries, aa, env ? *env : EnvView{}, env ? 1 : 0);
  return false;
}
void launch_collide_items(hipStream_t s, const EnvView& env, const RobotView& rob, const double* pos6, int n_pose,
                          const int32_t* live_flags, uint8_t* pose_hit, const double* a6, const double* b6,
                          const int32_t* seg_ns, int stride, int32_t* ctrl, const void* items, int items_cap,
                          const int32_t* sub, int32_t* first_hit, int32_t* overflow_flag, const TempGridRef* temps,
                          const int32_t* dev_n, const ClassifyArgs* block_src) {
  if (n_pose <= 0) return;
  TaskSource D{};
  if (block_src) {
    D.on = 1; D.rec_nnb = block_src->rec_nnb; D.rec_nb = block_src->rec_nb; D.rec_meta = block_src->rec_meta;
    D.parent = block_src->parent; D.center = block_src->center; D.pos = block_src->pos; D.nbcap = block_src->nbcap;
    D.goal_id = block_src->goal_id;
  }
  {
    const int cap_override = getenv("SFFGPU_SEG_LISTCAP") ? atoi(getenv("SFFGPU_SEG_LISTCAP")) : -1;
    if (cap_override >= 0 && cap_override < items_cap) items_cap = cap_override;
  }
  size_t lds = collide_lds_bytes(rob.n_tri, SEG_WAVES);
  // many-candidate items shared by the workgroup's wavefronts: environments of more than 4 096 triangles (building.obj:
  // 26 908, chunks with 35-130 candidates; dense_3D's 1 832 give 2-8 per item); SFFGPU_SHARE=0 / 1 overrides
  const char* se = getenv("SFFGPU_SHARE");
  const bool share = se ? atoi(se) != 0 : env.n_tri > 4096;
  auto kern = share ? k_collide_items<true> : k_collide_items<false>;
  if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  // 2 workgroups of 4 waves per CU = what the exact kernel's register budget keeps resident (256 CUs)
  static const int blocks = std::min(4096, std::max(1, getenv("SFFGPU_SEG_BLOCKS") ? atoi(getenv("SFFGPU_SEG_BLOCKS")) : 256 * CI_OCC));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * SEG_WAVES), lds, s, env, rob, pos6, n_pose, live_flags, pose_hit,
                     a6, b6, seg_ns, stride, ctrl, static_cast<const SurvivorItem*>(items), items_cap, sub, first_hit,
                     overflow_flag,
                     temps ? temps->tg : GridView{}, temps ? temps->x : nullptr, temps ? temps->y : nullptr,
                     temps ? temps->z : nullptr, temps ? temps->n : 0, dev_n, D);
}

void launch_star_exact(hipStream_t s, const EnvView& env, const RobotView& rob, const double* store_pos, const StarView& S,
                       int pass) {
  size_t lds = collide_lds_bytes(rob.n_tri, SEG_WAVES);
  if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_star_exact), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  static const int blocks = std::min(4096, std::max(1, getenv("SFFGPU_SEG_BLOCKS") ? atoi(getenv("SFFGPU_SEG_BLOCKS")) : 256 * CI_OCC));
  hipLaunchKernelGGL(k_star_exact, dim3(blocks), dim3(64 * SEG_WAVES), lds, s, env, rob, store_pos, S.ida, S.idb,
                     static_cast<const SurvivorItem*>(S.items), S.items_cap, S.sub + (size_t)pass * SFFK_SUBLISTS * SFFK_STAR_SUB,
                     S.first_hit, S.seg_ovf, S.hdr);
}

void launch_star_tail(hipStream_t s, const ResolveArgs& a, const EnvView& env, const RobotView& rob, const NodeStoreView& st,
                      int n_bound, int max_passes, int wgs_bound, int test_stall) {
  // (per call, for the CURRENT device: a process may hold contexts on several)
  const size_t lds = collide_lds_bytes(rob.n_tri, SEG_WAVES);
  if (lds > 32 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_star_tail), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  // resident at once: at most one workgroup per CU of an otherwise idle GPU; wgs_bound (SFFGPU_STAR_TAIL_WGS) bounds it further
  // (processes sharing one GPU: the sum of their grids must fit, or their barriers wait for each other until the time-out faults)
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  static int cu_of[64] = {0};
  if (dev >= 0 && dev < 64 && cu_of[dev] > 0) cus = cu_of[dev];
  else {
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    cus = std::max(1, cus);
    if (dev >= 0 && dev < 64) cu_of[dev] = cus;
  }
  const int cap = wgs_bound > 0 ? std::min(wgs_bound, cus) : cus;
  const int blocks = std::max(1, std::min(cap, std::max(16, (n_bound + 3) / 4)));
  hipLaunchKernelGGL(k_star_tail, dim3(blocks), dim3(64 * SEG_WAVES), lds, s, a, env, rob, st, max_passes, test_stall);
}

void launch_classify(hipStream_t s, const ClassifyArgs& a) {
  if (a.n <= 0) return;
  hipLaunchKernelGGL(k_classify, dim3((a.n + 3) / 4), dim3(256), 0, s, a);
}

void launch_seg_gather(hipStream_t s, const double* store_pos, const int32_t* ida, const int32_t* idb, int n, double* a6,
                       double* b6, const double* extra) {
  if (n <= 0) return;
  const long long threads = (long long)n * 12;
  hipLaunchKernelGGL(k_seg_gather, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, store_pos, ida, idb, n, a6, b6, extra);
}

void launch_seg_prepare(hipStream_t s, const double* a6, const double* b6, int n, int32_t* seg_ns, int32_t* first_hit,
                        int32_t* ovf) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_seg_prepare, dim3((n + 255) / 256), dim3(256), 0, s, a6, b6, n, seg_ns, first_hit, ovf);
}

__global__ __launch_bounds__(256) void k_tgrid_clear(GridView tg, const float* __restrict__ tx, const float* __restrict__ ty,
                                                     const float* __restrict__ tz, int n) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t == 0) tg.ovf_cnt[0] = 0;
  if (t >= n) return;
  const float x = tx[t];
  if (x == x) {
    const size_t cell = grid_cell_of(tg, x, ty[t], tz[t]);
    tg.cnt[cell] = 0;
    if (tg.occ) tg.occ[cell >> 5] = 0u;
  }
}
void launch_tgrid_clear(hipStream_t s, const TempGridRef& t) {
  hipLaunchKernelGGL(k_tgrid_clear, dim3((std::max(t.n, 1) + 255) / 256), dim3(256), 0, s, t.tg, t.x, t.y, t.z, t.n);
}

// compact -> cull -> exact.  pos6 / pose_hit may be null (edges only); temps (optional) = the round's own grid
// and the fp32 coordinates of its n_temps samples, emptied by the compaction launch.
void launch_round_collide(hipStream_t s, const EnvView& env, const RobotView& rob, const double* pos6, int n_pose,
                          const int32_t* live_flags, uint8_t* pose_hit, const double* a6, const double* b6,
                          const int32_t* seg_ns, int n_slots, int32_t* ctrl, void* list, int list_cap, void* masks,
                          int32_t* first_hit, int32_t* overflow_flag, const TempGridRef* temps, const int32_t* dev_n,
                          int stride) {
  if (!pose_hit) n_pose = 0;
  if (n_slots <= 0 && n_pose <= 0) return;
  size_t lds = collide_lds_bytes(rob.n_tri, SEG_WAVES);
  if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_collide_segments_dyn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  // 2 workgroups of 4 waves per CU = what the exact kernel's register budget keeps resident (256 CUs)
  static const int blocks = std::min(4096, std::max(1, getenv("SFFGPU_SEG_BLOCKS") ? atoi(getenv("SFFGPU_SEG_BLOCKS")) : 512));
  static const int cull_blocks = getenv("SFFGPU_CULL_BLOCKS") ? atoi(getenv("SFFGPU_CULL_BLOCKS")) : 2048;
  const int cap_override = getenv("SFFGPU_SEG_LISTCAP") ? atoi(getenv("SFFGPU_SEG_LISTCAP")) : -1;  // tests
  if (cap_override >= 0 && cap_override < list_cap) list_cap = cap_override;
  if (n_slots > 0)
    hipLaunchKernelGGL(k_seg_compact, dim3((n_slots + 1023) / 1024), dim3(256), 0, s, seg_ns, n_slots, a6, b6, ctrl,
                       static_cast<WorkItem*>(list), list_cap, temps ? temps->tg : GridView{}, temps ? temps->x : nullptr,
                       temps ? temps->y : nullptr, temps ? temps->z : nullptr, temps ? temps->n : 0, dev_n, stride);
  const int pose_blocks = (n_pose + 255) / 256;
  hipLaunchKernelGGL(k_cull, dim3(pose_blocks + (n_slots > 0 ? cull_blocks : 0)), dim3(256), 0, s, env, pos6, n_pose,
                     pose_blocks, live_flags, pose_hit, static_cast<const WorkItem*>(list),
                     static_cast<unsigned long long*>(masks), blocks * SEG_WAVES, ctrl, dev_n);
  hipLaunchKernelGGL(k_collide_segments_dyn, dim3(blocks), dim3(64 * SEG_WAVES), lds, s, env, rob, pos6, n_pose,
                     pose_hit, a6, b6, seg_ns, n_slots, ctrl, static_cast<const WorkItem*>(list),
                     static_cast<const unsigned long long*>(masks), first_hit, overflow_flag, dev_n, stride);
}

void launch_collide_segments_dyn(hipStream_t s, const EnvView& env, const RobotView& rob, const double* a6,
                                 const double* b6, const int32_t* seg_ns, int n_slots, int32_t* ctrl,
                                 void* list, int list_cap, void* masks, int32_t* first_hit,
                                 int32_t* overflow_flag) {
  if (n_slots <= 0) return;
  launch_round_collide(s, env, rob, nullptr, 0, nullptr, nullptr, a6, b6, seg_ns, n_slots, ctrl, list, list_cap, masks,
                       first_hit, overflow_flag, nullptr);
}

}  // namespace sffk
