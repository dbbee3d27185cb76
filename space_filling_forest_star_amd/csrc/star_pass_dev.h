// star_pass_dev.h - one pass of SFF*'s fixed point for ONE accepted sample (a wavefront), and the border entries' costs:
// the bodies of k_star_pass (devstar.hip, one launch per pass) and of k_star_tail (kernels.hip: the passes after the first
// as one launch).  See kernels.h (StarView) for the scheme; reference: src/forest.h:307-351.
#pragma once

namespace sffk {

using namespace sffg;

#define STAR_INF __longlong_as_double(0x7ff0000000000000LL)
#define STAR_FAULT 4     // indices in StarView::hdr
#define STAR_PASSES_RUN 5  // (k_star_tail: passes the round took ...
#define STAR_CONVERGED 6   //  ... and whether the last of them changed nothing)

// COH: the accesses of k_star_tail, where the workgroups of ONE launch exchange these words between the passes - relaxed
// agent-scope atomics (served by memory, written through) for every word another workgroup writes or reads between two of
// its barriers; the rest (fixed after k_star_knn, or only ever touched by the sample's own wavefront) stays on the caches.
// Without COH (one launch per pass: the kernel boundary does it) plain loads and stores.
template <bool COH> __device__ __forceinline__ int sld(const int32_t* p) {
  if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else return *p;
}
template <bool COH> __device__ __forceinline__ double sld(const double* p) {
  if constexpr (COH) return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  else return *p;
}
template <bool COH> __device__ __forceinline__ void sst(int32_t* p, int v) {
  if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}
template <bool COH> __device__ __forceinline__ void sst(double* p, double v) {
  if constexpr (COH) __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
template <bool COH> __device__ __forceinline__ SurvivorItem sld(const SurvivorItem* p) {
  if constexpr (COH) {
    const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
    const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return SurvivorItem{(int)(unsigned)(a & 0xffffffffULL), (int)(unsigned)(a >> 32), b};
  } else return *p;
}
template <bool COH> __device__ __forceinline__ void sst(SurvivorItem* p, const SurvivorItem& v) {
  if constexpr (COH) {
    unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
    __hip_atomic_store(q, (unsigned long long)(unsigned)v.slot | ((unsigned long long)(unsigned)v.chunk << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 1, v.mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else *p = v;
}

// DistanceToRoot of node x as sample `i` finds it: the proposal of the latest accepted sample before i whose rewire
// of x is active, else the node's own cost (a node created by this round: its sample's chosen cost)
template <bool COH> __device__ __forceinline__ double star_view(const DevForestView& f, const StarView& S, int x, int i, int N0, unsigned ep) {
  const unsigned long long h = S.head[x];
  int q = (unsigned)(h >> 32) == ep ? (int)(unsigned)(h & 0xffffffffULL) : 0;
  // (the node's own cost is asked for now, beside the list: most views end there)
  const int own = x < N0 ? -1 : S.acc_sample[x - N0];
  const double base = x < N0 ? f.d_root[x] : sld<COH>(S.best + own);
  int bs = -1;
  double bv = 0;
  // a list holds at most one pair per accepted sample; four links are followed before the proposals of the earlier samples
  // among them are asked for together (a link is fixed since k_star_knn - cache; a proposal may come from memory)
  for (int guard = 0; q && guard < (1 << 15); ++guard) {
    int pp[4];
    bool ok[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool take = q != 0;
      const int p = take ? q - 1 : 0;
      const int s = p / SFFK_STAR_KC;
      pp[k] = p;
      ok[k] = take && s < i && s > bs;
      q = take ? S.next[p] : 0;
    }
    double pr[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) pr[k] = ok[k] ? sld<COH>(S.prop + pp[k]) : STAR_INF;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int s = pp[k] / SFFK_STAR_KC;
      if (ok[k] && pr[k] < STAR_INF && s > bs) { bs = s; bv = pr[k]; }
    }
  }
  return bs >= 0 ? bv : base;
}

#ifdef STAR_PASS_TRACE   // (profiling build: ticks of the sections of star_pass_sample, wave 0 of workgroup 0, in S.dbg[24..31])
#define STAR_T(k) do { if (S.dbg && COH && blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long t_ = wall_clock64(); S.dbg[24 + (k)] += t_ - tt_; tt_ = t_; } } while (0)
#define STAR_T0 unsigned long long tt_ = wall_clock64();
#else
#define STAR_T(k) do { } while (0)
#define STAR_T0
#endif
#define STAR_SURV 64   // survivors of a sample gathered in LDS before they are appended (one atomic)
#define STAR_TAB 128   // (edge, chunk) pairs of a sample unfolded at a time
__device__ __forceinline__ int star_pack(int calls, bool free_) { return (((calls << 1) | (free_ ? 1 : 0)) << 1) | 1; }


struct StarPassLds {
  SurvivorItem surv[4][STAR_SURV];
  int32_t tab[4][STAR_TAB];
  float edge[4][128][8];     // per requested edge (lane * 2 + dir): start (clearance cells), step, samples
  int32_t any[4][128];       // ... 1 = some chunk of it went to the exact kernel
};

// border entries of the round (src/forest.h:288-294): d = cost(neighbour) + cost(expanded) + their distance, the costs as
// the rejected sample's turn finds them
template <bool COH> __device__ __forceinline__ void star_pass_event(const DevForestView& f, const StarView& S, int e, int first, int N0, unsigned ep) {
  const int s = S.ev_sample[e];
  const double vn = star_view<COH>(f, S, S.ev_nb[e], s, N0, ep), ve = star_view<COH>(f, S, S.ev_ex[e], s, N0, ep);
  f.b_dist[first + e] = vn + ve + S.ev_dist[e];
}

// sample i (rank among the accepted ones irrelevant here) by the wavefront `wave` of its workgroup; slot = which of the
// SFFK_STAR_PASSES counter sets / changed flags this pass writes, sub_list = the survivor sub-list its requests go onto
template <bool COH> __device__ __forceinline__ void star_pass_sample(const ResolveArgs& A, const EnvView& env, const NodeStoreView& st, int slot,
                                                 int sub_list, int i, int N0, int Tb, unsigned ep, StarPassLds& L, int wave,
                                                 int lane) {
  const DevForestView& f = A.f;
  const StarView& S = A.S;
  STAR_T0
  const int cnt = S.m_cnt[i];
  const size_t p = (size_t)i * SFFK_STAR_KC + lane;
  const bool mem = lane < cnt;
  const int ex = A.parent[i];
  const int x = mem ? S.m_id[p] : (lane == cnt ? ex : -1);
  const double d = mem ? S.m_d[p] : 0.0;
  const size_t s0 = p * 2;
  int ew_f = mem ? S.ew[s0] : 0, ew_b = mem ? S.ew[s0 + 1] : 0;
  const double pd = A.pdist[i];
  // (what the sample wrote last time, compared at the end: asked for now, in the shadow of the views)
  const double o_prop = mem ? S.prop[p] : 0.0, o_best = S.best[i], o_dcl = S.dcl[i];
  const int o_psel = S.psel[i];
  const unsigned long long o_cc = S.cnt[2 * (size_t)i], o_pf = S.cnt[2 * (size_t)i + 1];
  // ---- edges sent to the exact kernel by the pass before: their answers are in (asked for beside the views)
  const int fh_f = ew_f == -1 ? sld<COH>(S.first_hit + s0) : 0x7fffffff, fh_b = ew_b == -1 ? sld<COH>(S.first_hit + s0 + 1) : 0x7fffffff;
#ifdef STAR_PASS_TRACE
  if (pd == 1e300 || d == 1e300 || fh_f == -7) return;   // (the loads above are in before the clock is read)
#endif
  STAR_T(0);
  const double v = x >= 0 ? star_view<COH>(f, S, x, i, N0, ep) : 0.0;
#ifdef STAR_PASS_TRACE
  if (v == 1e300) return;
#endif
  STAR_T(1);
  {
    bool bad = false;
    if (ew_f == -1) {
      const int fh = fh_f, ns = S.ens[s0];
      bad |= fh == 0;                                     // 0 = the edge's triangle candidate list ran over
      ew_f = star_pack(fh == 0x7fffffff ? ns : fh, fh == 0x7fffffff);
      S.ew[s0] = ew_f;
    }
    if (ew_b == -1) {
      const int fh = fh_b, ns = S.ens[s0 + 1];
      bad |= fh == 0;
      ew_b = star_pack(fh == 0x7fffffff ? ns : fh, fh == 0x7fffffff);
      S.ew[s0 + 1] = ew_b;
    }
    if (__any(bad)) {
      if (lane == 0) atomicOr(S.hdr + STAR_FAULT, 1);
      return;
    }
  }
  // ---- which member edges can the two loops reach at all, given the views?  Choose-parent (:320-327) only looks at a
  // member whose cost through it beats the running best, and the running best never exceeds its start value; the rewire
  // loop (:332-350) only at a member the new node's cost - at least the smallest cost any member offers - improves.
  const double best0 = pd + __shfl(v, cnt);           // dist(new, expanded) + expanded->DistanceToRoot (:308)
  const double nd = d + v;
  const bool need_f = mem && nd < best0 - SFFG_TOL;
  double best_low = need_f ? nd : best0;
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_xor(best_low, off);
    best_low = o < best_low ? o : best_low;
  }
  const bool need_b = mem && best_low + d < v - SFFG_TOL;
  const bool req_f = need_f && ew_f == 0, req_b = need_b && ew_b == 0;
  if (__any(req_f || req_b)) {
    // ---- new requests: sample counts, then the clearance cull of their samples right here (~97 % of the chunks are
    // answered "free" by the bits); what is not goes onto the exact kernel's list and is answered before the next pass
    const double qp[6] = {A.newpos[6 * (size_t)i], A.newpos[6 * (size_t)i + 1], A.newpos[6 * (size_t)i + 2],
                          A.newpos[6 * (size_t)i + 3], A.newpos[6 * (size_t)i + 4], A.newpos[6 * (size_t)i + 5]};
    int nch_f = 0, nch_b = 0, ns_f = 0, ns_b = 0;
    const int sid = mem ? (x < N0 ? x : Tb + S.acc_sample[x - N0]) : 0;
    if (req_f || req_b) {
      double mp[6];
      for (int q = 0; q < 6; ++q) mp[q] = st.pos[6 * (size_t)sid + q];
      float* ef = L.edge[wave][2 * lane];
      float* eb = L.edge[wave][2 * lane + 1];
      if (req_f) {
        const double parts = edge_parts(qp, mp);
        ns_f = edge_samples(parts);
        nch_f = ns_f > 0 ? (ns_f + 63) >> 6 : 0;
        const float inv = (float)env.clear_inv * __frcp_rn((float)parts);
        for (int q = 0; q < 3; ++q) {
          ef[q] = (float)((qp[q] - env.clear_org[q]) * env.clear_inv);
          ef[4 + q] = (float)(mp[q] - qp[q]) * inv;
        }
        ef[3] = __int_as_float(ns_f);
        L.any[wave][2 * lane] = 0;
      }
      if (req_b) {
        const double parts = edge_parts(mp, qp);
        ns_b = edge_samples(parts);
        nch_b = ns_b > 0 ? (ns_b + 63) >> 6 : 0;
        const float inv = (float)env.clear_inv * __frcp_rn((float)parts);
        for (int q = 0; q < 3; ++q) {
          eb[q] = (float)((mp[q] - env.clear_org[q]) * env.clear_inv);
          eb[4 + q] = (float)(qp[q] - mp[q]) * inv;
        }
        eb[3] = __int_as_float(ns_b);
        L.any[wave][2 * lane + 1] = 0;
      }
    }
    const int my_nch = nch_f + nch_b;
    int incl = my_nch;
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(incl, off);
      if (lane >= off) incl += o;
    }
    const int excl = incl - my_nch;
    const int P = __shfl(incl, 63);
    SurvivorItem* list = static_cast<SurvivorItem*>(S.items);
    SurvivorItem* buf = L.surv[wave];
    int32_t* tab = L.tab[wave];
    int n_buf = 0;
    const int sub_cap = S.items_cap / SFFK_SUBLISTS;
    int32_t* sub = S.sub + ((size_t)slot * SFFK_SUBLISTS + sub_list) * SFFK_STAR_SUB;
    auto flush = [&]() {
      int base = 0;
      if (lane == 0) base = atomicAdd(sub, n_buf);
      base = __shfl(base, 0);
      for (int o = lane; o < n_buf; o += 64)
        if (base + o < sub_cap) sst<COH>(list + (size_t)sub_list * sub_cap + base + o, buf[o]);
      n_buf = 0;     // (a sub-list that ran over is noticed by k_star_exact: fault)
    };
    const float nxd = (float)env.clear_n[0], nyd = (float)env.clear_n[1], nzd = (float)env.clear_n[2];
    const int pu = lane >> 3, gq = lane & 7;
    for (int w0 = 0; w0 < P; w0 += STAR_TAB) {
      __builtin_amdgcn_wave_barrier();
      for (int cc = 0; cc < my_nch; ++cc) {
        const int pp = excl + cc;
        if (pp >= w0 && pp < w0 + STAR_TAB) {
          const int e = cc < nch_f ? 2 * lane : 2 * lane + 1;
          const int ch = cc < nch_f ? cc : cc - nch_f;
          tab[pp - w0] = (e << 16) | ch;
        }
      }
      __builtin_amdgcn_wave_barrier();
      const int wn = P - w0 < STAR_TAB ? P - w0 : STAR_TAB;
      for (int q0 = 0; q0 < wn; q0 += 8) {
        // eight consecutive samples of an edge lie within 0.4 units of the fifth one (the sample spacing never exceeds
        // the 0.1 of src/problemStruct.h:121) and the bits are built with that reach on top (Ctx::build_clearance): one
        // lookup answers a group of eight samples, a lane takes a group, a step of the wave eight (edge, chunk) pairs
        const bool valid = q0 + pu < wn;
        const int ent = valid ? tab[q0 + pu] : 0;
        const int e = ent >> 16, ch = ent & 0xffff;
        const float* ee = L.edge[wave][e];
        const int ns = __float_as_int(ee[3]);
        const int first = 1 + 64 * ch + 8 * gq;
        bool need = valid && first <= ns;
        const int left = ns - first + 1;
        const int probe = first + 4 <= ns ? first + 4 : ns;
        const uint32_t* wp = nullptr;
        int sh = 0;
        if (need && env.clear_bits_edge) {
          const float td = (float)probe;
          const float fx = __builtin_fmaf(td, ee[4], ee[0]), fy = __builtin_fmaf(td, ee[5], ee[1]), fz = __builtin_fmaf(td, ee[6], ee[2]);
          if (fx >= 0 && fy >= 0 && fz >= 0 && fx < nxd && fy < nyd && fz < nzd) {
            const uint32_t ci = ((uint32_t)(int)fz * (uint32_t)env.clear_n[1] + (uint32_t)(int)fy) * (uint32_t)env.clear_n[0] + (uint32_t)(int)fx;
            wp = env.clear_bits_edge + (ci >> 5);
            sh = (int)(ci & 31u);
          } else if (fx == fx && fy == fy && fz == fz) {
            need = false;                                     // beyond the inflated box of the environment
          }
        }
        const uint32_t word = wp ? *wp : 0u;
        if (wp && ((word >> sh) & 1u)) need = false;
        unsigned long long m = need ? (((left >= 8 ? 0xffULL : ((1ULL << left) - 1ULL))) << (8 * gq)) : 0ULL;
        m |= __shfl_xor(m, 1);
        m |= __shfl_xor(m, 2);
        m |= __shfl_xor(m, 4);
        const bool lead = gq == 0 && m != 0ULL && env.n_tri != 0;
        if (__any(lead)) {
          if (lead) L.any[wave][e] = 1;
          surv_emit(buf, n_buf, lead, lane, (int)(((size_t)i * SFFK_STAR_KC) * 2) + e, ch, m);
          if (n_buf > STAR_SURV - 33) flush();
        }
      }
    }
    if (n_buf) flush();
    STAR_T(3);
    __builtin_amdgcn_wave_barrier();
    if (req_f) {
      if (L.any[wave][2 * lane]) {
        ew_f = -1;
        sst<COH>(S.first_hit + s0, 0x7fffffff); sst<COH>(S.seg_ovf + s0, 0); S.ens[s0] = ns_f; sst<COH>(S.ida + s0, Tb + i); sst<COH>(S.idb + s0, sid);
      } else ew_f = star_pack(ns_f, true);
      S.ew[s0] = ew_f;
    }
    if (req_b) {
      if (L.any[wave][2 * lane + 1]) {
        ew_b = -1;
        sst<COH>(S.first_hit + s0 + 1, 0x7fffffff); sst<COH>(S.seg_ovf + s0 + 1, 0); S.ens[s0 + 1] = ns_b; sst<COH>(S.ida + s0 + 1, sid); sst<COH>(S.idb + s0 + 1, Tb + i);
      } else ew_b = star_pack(ns_b, true);
      S.ew[s0 + 1] = ew_b;
    }
  }
  STAR_T(2);
  const bool pending = __any((need_f && ew_f == -1) || (need_b && ew_b == -1));
  // (an edge still with the exact kernel counts as blocked here; the pass after its answer redoes the sample)
  const bool free_f = ew_f > 0 && ((ew_f >> 1) & 1), free_b = ew_b > 0 && ((ew_b >> 1) & 1);
  const unsigned long long calls_f = ew_f > 0 ? (unsigned long long)(ew_f >> 2) : 0ULL;
  const unsigned long long calls_b = ew_b > 0 ? (unsigned long long)(ew_b >> 2) : 0ULL;
  // ---- choose parent (:320-327): the members in (distance, id) order against the running best
  double best = best0;
  int psel = ex;
  double dcl = pd;
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
  STAR_T(4);
  // ---- write what changed
  // what OTHER samples read of this one is its proposals and its cost (star_view); parent choice, distance and the call counters
  // are outputs only: a pass in which no proposal and no cost changed and no sample waits for an edge has reached the fixed
  // point, whatever else it wrote
  const bool diff_view = (mem && __double_as_longlong(o_prop) != __double_as_longlong(np)) ||
                         (lane == 0 && __double_as_longlong(o_best) != __double_as_longlong(best));
  const bool diff_out = lane == 0 && (o_psel != psel || __double_as_longlong(o_dcl) != __double_as_longlong(dcl) || o_cc != cc || o_pf != pf);
  if (mem && __double_as_longlong(o_prop) != __double_as_longlong(np)) sst<COH>(S.prop + p, np);
  const bool any_view = __any(diff_view);
  const bool any_diff = any_view || __any(diff_out);
  if (lane == 0) {
    if (any_diff) {
      sst<COH>(S.best + i, best); S.psel[i] = psel; S.dcl[i] = dcl;
      S.cnt[2 * (size_t)i] = cc; S.cnt[2 * (size_t)i + 1] = pf;
    }
    if (any_view || pending) sst<COH>(S.changed + slot, 1);
  }
  STAR_T(5);
}

}  // namespace sffk
