// paths.cpp — post-loop path extraction of the SFF solvers (host only, runs once after the loop).
//
// Reference: SpaceForest::getPaths (src/forest.h:420-462) picks, per pair of trees, the border pair
// with the smallest root-to-root cost and strings the two root chains together;
// Solver::getAllPaths (src/problemStruct.h:184-253) closes the matrix over the connected trees by
// joining paths that share a third tree.  Costs feed params.csv / the TSP file
// (src/problemStruct.h:391-468); they are the "path cost" the parity bound is stated on.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "engine.h"
#include "sff_geom.h"

namespace sff {

static bool is_root(const Forest& f, int n) { return f.nodes[n].d_root == 0; }  // Node::IsRoot (primitives.h:476-478)

void Forest::get_paths() {
  nm.assign((size_t)num_roots * num_roots, Holder());
  for (int i = 0; i < num_roots; ++i)
    for (int j = i + 1; j < num_roots; ++j) {
      auto it = borders.find({i, j});
      if (it == borders.end() || it->second.empty()) continue;
      double best = -1;
      for (Border& b : it->second) {
        // DistanceHolder::UpdateDistance (primitives.h:652-654)
        b.dist = nodes[b.n1].d_root + nodes[b.n2].d_root + sffg::dist6(nodes[b.n1].pos, nodes[b.n2].pos);
        if (best == -1 || b.dist < best - SFFG_TOL) {
          best = b.dist;
          Holder h;
          h.n1 = b.n1;
          h.n2 = b.n2;
          h.dist = b.dist;
          NM(i, j) = h;
        }
      }
      Holder& h = NM(i, j);
      std::vector<int> chain;
      for (int n = h.n1;; n = nodes[n].parent) {   // node1 back to its root, then reversed (push_front, :439-444)
        chain.push_back(n);
        if (is_root(*this, n)) break;
      }
      h.plan.assign(chain.rbegin(), chain.rend());
      for (int n = h.n2;; n = nodes[n].parent) {   // node2 forward to its root (push_back, :447-452)
        h.plan.push_back(n);
        if (is_root(*this, n)) break;
      }
      if (cfg.optimize)                             // :454-457 costs may have changed by rewiring
        h.dist = nodes[h.n1].d_root + nodes[h.n2].d_root + sffg::dist6(nodes[h.n1].pos, nodes[h.n2].pos);
    }
}

void Forest::get_all_paths() {
  const int nc = (int)connected.size();
  for (int k = 0; k < nc; ++k) {
    const int id3 = connected[k];
    for (int i = 0; i < nc; ++i) {
      const int id1 = connected[i];
      if (i == k || !NM(id1, id3).exists()) continue;
      for (int j = 0; j < nc; ++j) {
        const int id2 = connected[j];
        if (i == j || !NM(id2, id3).exists()) continue;
        const Holder h1 = NM(id1, id3), h2 = NM(id2, id3);
        std::vector<int> plan1 = h1.plan, plan2 = h2.plan;
        int node1, node2;
        if (nodes[h1.n1].tree == id1) node1 = h1.n1; else { node1 = h1.n2; std::reverse(plan1.begin(), plan1.end()); }
        if (nodes[h2.n1].tree == id2) node2 = h2.n1; else { node2 = h2.n2; std::reverse(plan2.begin(), plan2.end()); }
        int last = -1;
        while (!plan1.empty() && !plan2.empty() && plan1.back() == plan2.back()) {   // strip the shared tail (:224-228)
          last = plan1.back();
          plan1.pop_back();
          plan2.pop_back();
        }
        std::vector<int> fin(plan1.begin(), plan1.end());
        fin.push_back(last);
        fin.insert(fin.end(), plan2.rbegin(), plan2.rend());
        double d = 0;                                // Solver::computeDistance (:170-181)
        for (size_t q = 1; q < fin.size(); ++q) d += sffg::dist6(nodes[fin[q - 1]].pos, nodes[fin[q]].pos);
        if (d < NM(id1, id2).dist - SFFG_TOL) {      // :244-246
          Holder h;
          h.dist = d;
          if (node1 < node2) { h.n1 = node1; h.n2 = node2; h.plan = fin; }
          else { h.n1 = node2; h.n2 = node1; h.plan.assign(fin.rbegin(), fin.rend()); }
          NM(id1, id2) = h;
        }
      }
    }
  }
}

// SpaceForest::smoothPaths (src/forest.h:464-511): walk every path from its far end (index g) and connect
// it to the EARLIEST node t < g-1 whose straight edge is free, dropping the nodes in between.  The
// reference tests t = 0, 1, ... one isPathFree at a time and stops at the first free one; here all
// candidate edges of one g go to the GPU in a single batch and the first free one is taken, which is the
// same choice (isPathFree is pure).  Reference-equivalent call counters follow the early stop.
void Forest::smooth_paths() {
  for (int i = 0; i < num_roots; ++i)
    for (int j = i + 1; j < num_roots; ++j) {
      Holder& h = NM(i, j);
      if (!h.exists()) continue;
      std::vector<int>& plan = h.plan;
      int g = (int)plan.size() - 1;
      double prev_dist = h.dist;
      while (g > 0) {
        const int m = g - 1;   // candidates t = 0 .. g-2
        std::vector<uint8_t> fr(std::max(m, 0));
        std::vector<int32_t> fh(std::max(m, 0)), nsv(std::max(m, 0));
        if (m > 0) {
          std::vector<double> a((size_t)m * 6), b((size_t)m * 6);
          for (int t = 0; t < m; ++t) {
            memcpy(&a[6 * (size_t)t], nodes[plan[t]].pos, 48);
            memcpy(&b[6 * (size_t)t], nodes[plan[g]].pos, 48);
          }
          ctx->collide_segments(a.data(), b.data(), m, fr.data(), fh.data(), nsv.data());
        }
        int p = 0, t = 0;
        double cum = 0;
        bool changed = false;
        while (t < g - 1) {
          if (p != t) cum += sffg::dist6(nodes[plan[t]].pos, nodes[plan[p]].pos);
          st.path_free_calls += 1;
          st.collide_calls += fh[t] > 0 ? (uint64_t)fh[t] : (uint64_t)nsv[t];
          if (fr[t]) { changed = true; break; }
          p = t;
          ++t;
        }
        if (t == g - 1) cum += sffg::dist6(nodes[plan[t]].pos, nodes[plan[p]].pos);
        if (changed) {
          double dif = prev_dist - cum - sffg::dist6(nodes[plan[t]].pos, nodes[plan[g]].pos);
          h.dist -= dif;
          plan.erase(plan.begin() + t + 1, plan.begin() + g);
        }
        prev_dist = cum;
        g = t;
      }
    }
}

}  // namespace sff
