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

}  // namespace sffk
