// kernels_dev.h — device-side helpers shared by kernels.hip and devforest.hip (neighbour-grid addressing).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"

namespace sffk {

__device__ __forceinline__ int grid_coord(float v, float o, float inv, int n) {
  float f = floorf((v - o) * inv);
  int c = f < 0.0f ? 0 : (f > (float)(n - 1) ? n - 1 : (int)f);  // NaN compares false twice -> cast of NaN; guarded by callers
  return c;
}

__device__ __forceinline__ size_t grid_cell_of(const GridView& g, float x, float y, float z) {
  const int cx = grid_coord(x, g.ox, g.inv_cell, g.nx), cy = grid_coord(y, g.oy, g.inv_cell, g.ny),
            cz = grid_coord(z, g.oz, g.inv_cell, g.nz);
  return ((size_t)cz * g.ny + cy) * g.nx + cx;
}
__device__ __forceinline__ void grid_put(const GridView& g, const GridItem& it) {
  const size_t cell = grid_cell_of(g, (float)it.p[0], (float)it.p[1], (float)it.p[2]);
  const int slot = atomicAdd(g.cnt + cell, 1);
  if (g.occ) atomicOr(g.occ + (cell >> 5), 1u << (cell & 31));
  if (slot < g.bk) {
    g.items[cell * g.bk + slot] = it;
  } else {
    const int o = atomicAdd(g.ovf_cnt, 1);
    if (o < g.ovf_cap) g.ovf[o] = it;   // the host checks ovf_cnt against ovf_cap
  }
}

// ---- survivors of the clearance cull -> items of the exact kernel.  The lead lanes (lane % 8 == 0) of a step each hold one
// (edge slot, 64-sample chunk, mask) survivor.  The exact kernel's length is its longest item (one wavefront per item:
// broad phase of the masked samples' swept box, then every sample against every candidate triangle), so a survivor with
// many samples is emitted as up to four 16-sample items: four wavefronts, each with a tighter box and a quarter of the
// samples.  buf must have room for 32 more entries.
// Measured (profiles/r3_head_a / r3_c5_d, split at > 20 samples): the longest launch shrinks (178 -> 104 us on dense_3D)
// but the average grows (37 -> 49 us: every item pays its own broad phase), on building.obj 102 -> 96 us: off by default.
#ifndef SFFK_SPLIT_MIN
#define SFFK_SPLIT_MIN 64
#endif
__device__ __forceinline__ void surv_emit(SurvivorItem* buf, int& n_buf, bool lead, int lane, int slot, int chunk,
                                          unsigned long long m) {
  int parts = 0;
  const bool split = lead && __popcll(m) > SFFK_SPLIT_MIN;
  if (lead) parts = split ? ((m & 0xffffULL) != 0) + ((m & 0xffff0000ULL) != 0) + ((m & 0xffff00000000ULL) != 0) + ((m >> 48) != 0) : 1;
  int inc = parts;
  for (int off = 8; off < 64; off <<= 1) {
    const int o = __shfl_up(inc, off);
    if (lane >= off) inc += o;
  }
  const int total = __shfl(inc, 56);
  if (lead) {
    int at = n_buf + inc - parts;
    if (!split) buf[at] = SurvivorItem{slot, chunk, m};
    else
      for (int q = 0; q < 4; ++q) {
        const unsigned long long mq = m & (0xffffULL << (16 * q));
        if (mq) buf[at++] = SurvivorItem{slot, chunk, mq};
      }
  }
  n_buf += total;
}

// ---- exact k nearest: the wave-resident top-k list shared by k_knn_linear / k_knn_grid (kernels.hip) and k_star_knn
// (devstar.hip)
// The wave's k best so far: lane j holds the j-th smallest (distance, id) key; lanes >= have hold +inf.
struct TopK {
  double d;
  int id;
};
__device__ __forceinline__ bool key_less(double da, int ia, double db, int ib) { return da < db || (da == db && ia < ib); }
// inserts the candidates flagged in `take` (one per lane: cd, cid), smallest lanes first; k = capacity, have = filled
__device__ __forceinline__ void topk_insert(TopK& t, int lane, int k, int& have, unsigned long long take, double cd, int cid) {
  while (take) {
    const int src = __ffsll((long long)take) - 1;
    take &= take - 1;
    const double nd = __shfl(cd, src);
    const int ni = __shfl(cid, src);
    // rank of the newcomer = entries that sort before it
    const bool before = lane < have && key_less(t.d, t.id, nd, ni);
    const int rank = __popcll(__ballot(before));
    if (rank >= k) continue;                       // (beaten by k entries that arrived in the meantime)
    const double pd = __shfl_up(t.d, 1);
    const int pi = __shfl_up(t.id, 1);
    if (lane > rank) { t.d = pd; t.id = pi; }
    else if (lane == rank) { t.d = nd; t.id = ni; }
    if (have < k) ++have;
  }
}
__device__ __forceinline__ double topk_worst(const TopK& t, int k, int have) {   // current k-th distance (inf while not full)
  return have < k ? 1.0e300 : __shfl(t.d, k - 1);
}


}  // namespace sffk
