// kernels_dev.h — device-side helpers shared by kernels.hip and devforest.hip (neighbour-grid addressing).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"
#include "sff_geom.h"

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
  GridItem32 lt;
  lt.x = (float)it.p[0]; lt.y = (float)it.p[1]; lt.z = (float)it.p[2];
  lt.yaw = (float)it.p[3]; lt.pitch = (float)it.p[4]; lt.roll = (float)it.p[5];
  lt.id = it.id; lt.tree = it.tree;
  if (slot < g.bk) {
    g.items[cell * g.bk + slot] = it;
    if (g.lite) g.lite[cell * g.bk + slot] = lt;
  } else {
    const int o = atomicAdd(g.ovf_cnt, 1);
    if (o < g.ovf_cap) {   // the host checks ovf_cnt against ovf_cap
      g.ovf[o] = it;
      if (g.ovf_lite) g.ovf_lite[o] = lt;
    }
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
// merges one candidate per lane (cv: valid, key (cd, cid)) into the list by RANKING instead of inserting one by one: every
// candidate and every list entry is broadcast once (scalar reads of its lane), every lane counts the keys below its own
// entry and below its own candidate, and each survivor is pushed to the lane of its rank (ds_permute; lane 63 takes what
// falls out - k <= 56).  One insertion is a dozen DEPENDENT cross-lane operations, ~0.6 us on a wavefront that is alone on
// its SIMD; a merge of five candidates into seventeen entries costs about what two insertions cost.  Same list as
// topk_insert leaves (keys are unique: ids are).
__device__ __forceinline__ void topk_merge(TopK& t, int lane, int k, int& have, bool cv, double cd, int cid) {
  unsigned long long mb = __ballot(cv);
  if (!mb) return;
  const int nb = __popcll(mb);
  const unsigned long long tb = (unsigned long long)__double_as_longlong(t.d), cb = (unsigned long long)__double_as_longlong(cd);
  const int tlo = (int)(unsigned)(tb & 0xffffffffULL), thi = (int)(unsigned)(tb >> 32);
  const int clo = (int)(unsigned)(cb & 0xffffffffULL), chi = (int)(unsigned)(cb >> 32);
  int ra = lane, rb = 0;       // my list entry's / my candidate's rank in the union (the list is sorted: lane entries of it are below mine)
  while (mb) {
    const int j = __ffsll((long long)mb) - 1;
    mb &= mb - 1;
    const double sd = __longlong_as_double((long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane(chi, j) << 32) |
                                                       (unsigned long long)(unsigned)__builtin_amdgcn_readlane(clo, j)));
    const int si = __builtin_amdgcn_readlane(cid, j);
    ra += key_less(sd, si, t.d, t.id) ? 1 : 0;
    rb += key_less(sd, si, cd, cid) ? 1 : 0;
  }
  for (int j = 0; j < have; ++j) {
    const double sd = __longlong_as_double((long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane(thi, j) << 32) |
                                                       (unsigned long long)(unsigned)__builtin_amdgcn_readlane(tlo, j)));
    const int si = __builtin_amdgcn_readlane(t.id, j);
    rb += key_less(sd, si, cd, cid) ? 1 : 0;
  }
  const int da = (lane < have && ra < k) ? ra : 63, db = (cv && rb < k) ? rb : 63;
  const int a_lo = __builtin_amdgcn_ds_permute(da << 2, tlo), a_hi = __builtin_amdgcn_ds_permute(da << 2, thi);
  const int a_id = __builtin_amdgcn_ds_permute(da << 2, t.id), a_on = __builtin_amdgcn_ds_permute(da << 2, 1);
  const int b_lo = __builtin_amdgcn_ds_permute(db << 2, clo), b_hi = __builtin_amdgcn_ds_permute(db << 2, chi);
  const int b_id = __builtin_amdgcn_ds_permute(db << 2, cid);
  const int total = have + nb;
  have = total < k ? total : k;
  if (lane < have) {
    const unsigned long long bits = a_on ? (((unsigned long long)(unsigned)a_hi << 32) | (unsigned)a_lo) : (((unsigned long long)(unsigned)b_hi << 32) | (unsigned)b_lo);
    t.d = __longlong_as_double((long long)bits);
    t.id = a_on ? a_id : b_id;
  } else { t.d = 1.0e300; t.id = 0x7fffffff; }
}
__device__ __forceinline__ double topk_worst(const TopK& t, int k, int have) {   // current k-th distance (inf while not full)
  return have < k ? 1.0e300 : __shfl(t.d, k - 1);
}

// ---- SFF* (devstar.hip, k_star_knn_wg in kernels.hip): is sample i of the committed round accepted, and its rank among them
__device__ __forceinline__ bool star_accepted(const DevForestView& f, int i, int& rank) {
  const unsigned long long w = f.w_acc[i >> 6];
  rank = f.acc_pref[i >> 6] + __popcll(w & ((1ULL << (i & 63)) - 1ULL));
  return (w >> (i & 63)) & 1ULL;
}

// the same over cells of the round's own grid: samples accepted EARLIER in the round (temporary id in [Tb, self)) of
// the same tree, not farther than `limit`
__device__ __forceinline__ void star_cells_mates(const GridView& tg, int m, int cell, int lane, const double* qp, int tree, int Tb,
                                                 int self, double limit, const DevForestView& f, TopK& t, int k, int& have) {
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
      const GridItem it = tg.items[(size_t)src_cell * tg.bk + slot];
      id = it.id;
      int rk;
      if (it.tree == tree && id >= Tb && id < self && star_accepted(f, id - Tb, rk)) {
        d = sffg::dist6(it.p, qp);
        cand = d <= limit;
      }
    }
    const double worst = topk_worst(t, k, have);
    cand = cand && (have < k || key_less(d, id, worst, 0x7fffffff));
    topk_insert(t, lane, k, have, __ballot(cand), d, id);
  }
}


// ---- sample + steer of one slot (k_sample_steer; the device engine's k_append_sample runs it for the NEXT round right
// behind the append).  tid = the thread's index in the launch (per-round housekeeping), i = the sample it draws (< 0:
// none), slot >= 0: the sample's slot (k_append_sample knows it; otherwise it is read from the active list).
__device__ __forceinline__ void sample_steer_one(int tid, int i, int slot, const SampleLaunch& P) {
  using namespace sffg;
  const uint64_t* __restrict__ words = P.words;
  const int32_t* __restrict__ parent = P.parent;
  const double* __restrict__ node_pos = P.node_pos;
  const double* __restrict__ center_in = P.center_in;
  int n = P.n;
  const double dist = P.dist;
  const int dim = P.dim;
  const SampleParams& prm = P.prm;
  double* __restrict__ out6 = P.out6;
  uint8_t* __restrict__ in_lim = P.in_lim;
  double* __restrict__ parent_dist = P.parent_dist;
  SweepQuery* __restrict__ queries = P.queries;
  const int32_t q_max_base = P.q_max_base;
  const RoundTemps& tmp = P.tmp;
  const DevRound& dv = P.dv;
  if (dv.ctrl) {   // device-resident forest: the round's size lives in HBM
    if (dv.ctrl->halt) return;
    n = dv.ctrl->n_act;
    if (tid == 0 && dv.qclk) { dv.qclk[0] = ~0ULL; dv.qclk[1] = 0ULL; }
    if (dv.qclk_sh && tid < 128) dv.qclk_sh[(tid >> 1) * 16 + (tid & 1)] = (tid & 1) ? ~0ULL : 0ULL;
  }
  if (tmp.cnt) {
    // per-round housekeeping folded into this launch: hit counters, the work-list cursor, and NaN
    // placeholders for the store entries between the permanent nodes and the 4-aligned temporaries
    if (tid < n) tmp.cnt[tid] = 0;
    if (tid < 32) tmp.ctrl[tid] = 0;
    if (tmp.sub && tid < SFFK_SUBLISTS) {   // heavy items, the exact kernel's item tickets, light items
      tmp.sub[tid * SFFK_SUB_STRIDE] = 0; tmp.sub[tid * SFFK_SUB_STRIDE + 1] = 0; tmp.sub[tid * SFFK_SUB_STRIDE + 2] = 0;
    }
    if (tid < tmp.base - tmp.n_perm) {
      const float nanv = __int_as_float(0x7fc00000);
      const size_t o = (size_t)tmp.n_perm + tid;
      tmp.st.x[o] = nanv; tmp.st.y[o] = nanv; tmp.st.z[o] = nanv;
      tmp.st.yaw[o] = nanv; tmp.st.pitch[o] = nanv; tmp.st.roll[o] = nanv;
    }
  }
  // the wave's first sampling launch (sample index == slot): every slot gets its position in the wave's spatial order
  if (dv.ctrl && dv.ord.hist && slot == -1 && dv.ctrl->ord_valid && dv.ctrl->round == 1 && tid < dv.ctrl->n_slots) {
    const int pos = dv.ord.start[dv.ord.slot_key[tid]] + dv.ord.slot_rank[tid];
    const int sel = dv.ctrl->act_sel, n_slots = dv.ctrl->n_slots;
    dv.ord.slot_pos[tid] = pos;
    (sel ? dv.ord.lst[1] : dv.ord.lst[0])[pos] = tid < n ? tid : -1;      // (this round: every position holds its slot's sample)
    if ((pos & 63) == 0) (sel ? dv.ord.cnt[1] : dv.ord.cnt[0])[(pos >> 6) * SFFK_ORD_CNT_STRIDE] = n_slots - pos < 64 ? n_slots - pos : 64;
  }
  if (i < 0 || i >= n) return;
  double c[6], o[6];
  int par = 0;
  if (dv.ctrl) {
    par = dv.slot_node[slot >= 0 ? slot : (dv.ctrl->act_sel ? dv.act_slot2 : dv.act_slot)[i]];
    dv.parent_out[i] = par;
    dv.force_out[i] = dv.nflag[par] & 1;
  } else if (!center_in) {
    par = parent[i];
  }
  const double* src = center_in ? center_in + 6 * (size_t)i : node_pos + 6 * (size_t)par;
  for (int k = 0; k < 6; ++k) c[k] = src[k];
  if (tmp.center_out) for (int k = 0; k < 6; ++k) tmp.center_out[6 * (size_t)i + k] = c[k];
  uint64_t w[6];
  SampleTrig host_trig{};
  if (dv.ctrl) {   // the sample's words sit in the engine-word ring, in the reference's draw order
    const unsigned long long base = dv.ctrl->words_base + (unsigned long long)dv.words_per * (unsigned long long)i;
    for (int k = 0; k < 6; ++k) w[k] = k < dv.words_per ? dv.ring[(base + k) & dv.ring_mask] : 0ULL;
    if (dv.trig) {   // libm parity mode: the transcendental values of these words, evaluated by the host's C library
      const double* t0 = dv.trig + 3 * (size_t)(base & dv.ring_mask);
      host_trig.c_phi = t0[0]; host_trig.s_phi = t0[1];
      if (dv.words_per == 6) {
        const double* t1 = dv.trig + 3 * (size_t)((base + 1) & dv.ring_mask);
        const double* t3 = dv.trig + 3 * (size_t)((base + 3) & dv.ring_mask);
        host_trig.c_theta = t1[0]; host_trig.s_theta = t1[1];
        host_trig.acos_u = t3[2];
      }
    }
  } else {
    for (int k = 0; k < 6; ++k) w[k] = words[6 * (size_t)i + k];
  }
  bool ok;
  if (dv.ctrl && dv.trig) {
    ok = sample_point_with(w, c, dist, dim, prm.limits, o, host_trig);
  } else if (tmp.preset) {
    for (int k = 0; k < 6; ++k) o[k] = tmp.preset[6 * (size_t)i + k];
    ok = in_limits(o, prm.limits);
  } else {
    ok = sample_point(w, c, dist, dim, prm.limits, o);
  }
  for (int k = 0; k < 6; ++k) out6[6 * (size_t)i + k] = o[k];
  in_lim[i] = ok ? 1 : 0;
  if (tmp.cnt) {
    // the sample becomes temporary store entry base + i (NaN floats when out of limits, so that no query
    // can match it), with the tree of the node it was expanded from
    const float nanv = __int_as_float(0x7fc00000);
    const size_t t = (size_t)tmp.base + i;
    tmp.st.x[t] = ok ? (float)o[0] : nanv;
    tmp.st.y[t] = ok ? (float)o[1] : nanv;
    tmp.st.z[t] = ok ? (float)o[2] : nanv;
    tmp.st.yaw[t] = ok ? (float)o[3] : nanv;
    tmp.st.pitch[t] = ok ? (float)o[4] : nanv;
    tmp.st.roll[t] = ok ? (float)o[5] : nanv;
    for (int k = 0; k < 6; ++k) tmp.st.pos[6 * t + k] = o[k];
    const int tr = tmp.st.tree[par];
    tmp.st.tree[t] = tr;
    if (ok && tmp.tg.cnt) {   // and into the round's own grid, where the later samples of the round look for it
      GridItem it;
      for (int k = 0; k < 6; ++k) it.p[k] = o[k];
      it.id = (int32_t)t;
      it.tree = tr;
      it.pad[0] = it.pad[1] = 0;
      grid_put(tmp.tg, it);
    }
  }
  if (parent_dist) {
    double pd = dist6(c, o);  // parentDistance, src/forest.h:250
    parent_dist[i] = pd;
    if (queries) {
      SweepQuery q;
      q.x = (float)o[0]; q.y = (float)o[1]; q.z = (float)o[2];
      q.yaw = (float)o[3]; q.pitch = (float)o[4]; q.roll = (float)o[5];
      double r = pd > prm.dist_tree ? pd : prm.dist_tree;
      q.r = r;
      double ri = (r + prm.sweep_abs_eps) * (1.0 + 1e-5);
      q.r2f = (float)(ri * ri) * 1.000001f;
      q.tree = -1;
      q.max_id = q_max_base + i;
      q.active = (ok && (prm.world <= 1 || i % prm.world == prm.rank)) ? 1 : 0;
      q.pad = 0;
      queries[i] = q;
      if (tmp.qrec) {
        QRec r;
        r.q[0] = q.x; r.q[1] = q.y; r.q[2] = q.z; r.q[3] = q.yaw; r.q[4] = q.pitch; r.q[5] = q.roll;
        r.r2f = q.r2f; r.max_id = q.max_id; r.q_tree = q.tree;
        r.mine = tmp.cnt ? tmp.st.tree[par] : 0;
        r.evaluate = q.active;
        // the cells the query ball's box touches (the round's own grid has the node grid's cells)
        const GridView& g = tmp.tg;
        const float rf = sqrtf(q.r2f) * 1.000001f;
        const int lx = grid_coord(q.x - rf, g.ox, g.inv_cell, g.nx), hx = grid_coord(q.x + rf, g.ox, g.inv_cell, g.nx);
        const int ly = grid_coord(q.y - rf, g.oy, g.inv_cell, g.ny), hy = grid_coord(q.y + rf, g.oy, g.inv_cell, g.ny);
        const int lz = grid_coord(q.z - rf, g.oz, g.inv_cell, g.nz), hz = grid_coord(q.z + rf, g.oz, g.inv_cell, g.nz);
        r.lx = lx; r.ly = ly; r.lz = lz; r.wx = hx - lx + 1; r.wy = hy - ly + 1;
        r.total = q.active ? r.wx * r.wy * (hz - lz + 1) : 0;
        // the parent edge, isPathFree(expanded, newPoint) (src/forest.h:246): its length is parentDistance
        const double parts0 = pd / 0.1;   // = edge_parts(c, o)
        r.ns0 = edge_samples(parts0);
        const float inv0 = (float)tmp.clear_inv * __frcp_rn((float)parts0);
        for (int k = 0; k < 3; ++k) {
          r.t0[k] = (float)((c[k] - tmp.clear_org[k]) * tmp.clear_inv);
          r.t0[3 + k] = (float)(o[k] - c[k]) * inv0;
        }
        r.pdist = pd; r.qr = q.r;
        r.scratch[0] = r.scratch[1] = r.scratch[2] = r.scratch[3] = 0;
        tmp.qrec[i] = r;
      }
    }
  }
}

}  // namespace sffk
