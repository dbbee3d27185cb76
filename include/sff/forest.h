// forest.h — source-compatible SpaceForest<T,R> (reference src/forest.h:31-54): same constructor and
// Solve(); the main loop (:122-202) runs inside libsffgpu as waves of frontier slots on the GPU, the
// post-processing (:203-235) stays here.
#pragma once
#include <cstring>

#include "problemStruct.h"

#ifndef SFF_COMPAT_WAVE
#define SFF_COMPAT_WAVE 8192
#endif

template <class T, class R = Point<T>>
class SpaceForest : public Solver<T, R> {
 public:
  SpaceForest(Problem<T>& problem) : Solver<T, R>(problem) {}

  // src/forest.h:513-568
  void saveFrontiers(const FileStruct file, const std::vector<int32_t>& open_nodes) {
    std::ofstream f;
    if (!this->open(f, file, "Saving frontiers")) return;
    if (file.type == Obj) f << "o Open nodes\n";
    for (int32_t id : open_nodes) {
      const Point<T> p = this->allNodes[id]->Position / this->problem.environment.ScaleFactor;
      if (file.type == Obj) {
        f << "v" << DELIMITER_OUT;
        if (this->usePriority) p.printPosOnly(f); else f << p;
        f << "\n";
      } else {
        f << p << DELIMITER_OUT << "1\n";
      }
    }
  }

  void Solve() override {
    Problem<T>& P = this->problem;
    P.environment.upload();
    sffgpu_forest_cfg cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.dim = P.dimension;
    cfg.optimize = P.optimal;
    cfg.has_goal = P.hasGoal;
    P.goal.toArray(cfg.goal);
    const Range<T>& l = P.environment.limits;
    const double lim[6] = {l.minX, l.maxX, l.minY, l.maxY, l.minZ, l.maxZ};
    std::memcpy(cfg.limits, lim, sizeof lim);
    cfg.dist_tree = this->treeDistance;
    cfg.sampling_dist = Node<T, R>::SamplingDistance;
    cfg.threshold_misses = Node<T, R>::ThresholdMisses;
    cfg.max_iterations = P.maxIterations;
    const char* w = std::getenv("SFF_WAVE");
    cfg.wave = w ? std::atoi(w) : SFF_COMPAT_WAVE;
    const char* s = std::getenv("SFF_SEED");   // the reference seeds from the clock (src/randGen.h:52-55)
    cfg.seed = s ? std::strtoull(s, nullptr, 10)
                 : (uint64_t)std::chrono::high_resolution_clock::now().time_since_epoch().count();
    cfg.rank = 0;
    cfg.world = 1;
    cfg.priority_bias = P.priorityBias;   // != 0: priority frontier heaps (src/heap.h)
    const char* lm = std::getenv("SFF_LIBM");   // 1: samples with the C library's trig, like the reference (parity mode)
    cfg.libm_sampling = (lm && std::atoi(lm)) ? 1 : 0;
    // SFF* + per-iteration tree dumps: rewires change parents later on, so the library keeps the parent history
    cfg.record_parents = (P.optimal && P.saveTreeIter != 0) ? 1 : 0;
    std::vector<double> roots;
    for (const Point<T>& p : P.roots) {
      double a[6];
      p.toArray(a);
      roots.insert(roots.end(), a, a + 6);
    }
    sffgpu_forest* f = nullptr;
    sff_compat::check(sffgpu_forest_create(sff_compat::gpu(), &cfg, roots.data(), (int)P.roots.size(), &f), "forest");

    sffgpu_forest_stats st;
    // upTo >= 0: only the nodes created up to that iteration.  Node ids follow the order of creation and (plain SFF)
    // a parent is always older than its child, so that forest is a prefix of the node arrays.
    auto loadNodes = [&](long upTo = -1) {
      sffgpu_forest_get_stats(f, &st);
      int n = st.n_nodes;
      std::vector<double> pos((size_t)n * 6), cost(n), dpar(n);
      std::vector<int32_t> parent(n), tree(n), iter(n);
      sffgpu_forest_get_nodes(f, pos.data(), parent.data(), tree.data(), iter.data(), cost.data(), dpar.data());
      if (upTo >= 0) {
        while (n > 0 && (long)iter[n - 1] > upTo) --n;
        if (cfg.record_parents) {
          // SFF*: every node's parent as iteration upTo left it = its last history entry up to then (src/forest.h:329, :344)
          const int m = sffgpu_forest_get_parent_history(f, nullptr, nullptr, nullptr, 0);
          sff_compat::check(m < 0 ? m : 0, "parent history");
          std::vector<int32_t> hn(m), hp(m), hi(m);
          sffgpu_forest_get_parent_history(f, hn.data(), hp.data(), hi.data(), m);
          for (int e = 0; e < m && (long)hi[e] <= upTo; ++e)
            if (hn[e] < n) parent[hn[e]] = hp[e];
        }
      }
      this->fillNodes(n, st.n_trees, pos.data(), parent.data(), tree.data(), iter.data(), cost.data(), dpar.data());
    };
    auto loadFrontier = [&](std::vector<int32_t>& open_nodes) {
      int k = sffgpu_forest_get_frontier(f, nullptr, 0);
      open_nodes.resize(k > 0 ? k : 0);
      if (k > 0) sffgpu_forest_get_frontier(f, open_nodes.data(), k);
    };

    auto startingTime = std::chrono::high_resolution_clock::now();   // :117
    if (P.saveTreeIter == 0 && P.saveFrontiersIter == 0) {
      sff_compat::check(sffgpu_forest_run(f, 0), "forest run");
    } else {
      // saveIterCheck (src/problemStruct.h:256-261, src/forest.h:570-578) dumps after every k-th iteration.  The
      // loop advances by waves, the dumps are nevertheless the state after iteration k:
      //  * trees: the nodes created up to k; SFF: their parents never change; SFF*: the parent history gives every
      //    node's parent as iteration k left it (rewires of later iterations undone);
      //  * frontier: a node joins the open list when it is created and leaves it when its slot is exhausted, which the
      //    reference does AFTER the slot's last saveIterCheck (:160-163) - here at the end of the wave: the open list after
      //    iteration k is the list the wave started with plus the nodes the wave has created up to k.
      //    (Priority-frontier mode pops a slot's node off its heap for the duration of the wave: wave-granular there.)
      long nextTree = P.saveTreeIter, nextFront = P.saveFrontiersIter;
      uint64_t wavesBefore = ~0ULL;
      const bool exactFront = P.saveFrontiersIter != 0 && P.priorityBias == 0;
      while (true) {
        sffgpu_forest_get_stats(f, &st);
        if (st.waves == wavesBefore) break;   // terminated: the last call did not start a wave
        wavesBefore = st.waves;
        std::vector<int32_t> open_before;
        const int nodes_before = st.n_nodes;
        if (exactFront) loadFrontier(open_before);
        sff_compat::check(sffgpu_forest_run(f, 1), "forest run");
        sffgpu_forest_get_stats(f, &st);
        while (P.saveTreeIter != 0 && (long)st.iterations >= nextTree) {
          loadNodes(nextTree);
          this->saveTrees(prefixFileName(P.fileNames[SaveTree], "iter_" + std::to_string(nextTree) + "_"));
          nextTree += P.saveTreeIter;
        }
        while (P.saveFrontiersIter != 0 && (long)st.iterations >= nextFront) {
          std::vector<int32_t> open_now;
          if (exactFront) {
            loadNodes(nextFront);
            open_now = open_before;
            for (int id = nodes_before; id < (int)this->allNodes.size(); ++id) open_now.push_back(id);
          } else {
            loadNodes();
            loadFrontier(open_now);
          }
          saveFrontiers(prefixFileName(P.fileNames[SaveFrontiers], "iter_" + std::to_string(nextFront) + "_"), open_now);
          nextFront += P.saveFrontiersIter;
        }
      }
    }
    auto stopTime = std::chrono::high_resolution_clock::now();       // :203

    loadNodes();
    this->pathCost.assign((size_t)st.n_trees * st.n_trees, 0.0);
    std::vector<int32_t> conn(st.n_trees);
    int nc = sffgpu_forest_paths(f, this->pathCost.data(), conn.data(), st.n_trees);   // getPaths + getAllPaths
    const std::vector<int> connected(conn.begin(), conn.begin() + nc);
    auto loadPlans = [&]() {
      this->plans.assign((size_t)st.n_trees * st.n_trees, {});
      for (int i = 0; i < st.n_trees; ++i)
        for (int j = i + 1; j < st.n_trees; ++j) {
          int len = sffgpu_forest_path_plan(f, i, j, nullptr, 0);
          if (len <= 0) continue;
          std::vector<int32_t> ids(len);
          sffgpu_forest_path_plan(f, i, j, ids.data(), len);
          this->plans[(size_t)i * st.n_trees + j].assign(ids.begin(), ids.end());
        }
    };
    loadPlans();
    this->fillPaths(connected);
    // :114-116, :207-235 — same order of outputs as the reference
    if (SaveGoals <= P.saveOptions) this->saveCities(P.fileNames[SaveGoals]);
    if (SaveTree <= P.saveOptions) this->saveTrees(P.fileNames[SaveTree]);
    if (SaveRaw <= P.saveOptions) this->savePaths(P.fileNames[SaveRaw]);
    if (P.smoothing) {                                               // :219-224
      sff_compat::check(sffgpu_forest_smooth_paths(f, this->pathCost.data()), "smooth paths");
      loadPlans();
      this->fillPaths(connected);
      if (SaveSmooth <= P.saveOptions) this->savePaths(P.fileNames[SaveSmooth]);
    }
    std::vector<int32_t> open_nodes;
    if (SaveFrontiers <= P.saveOptions) loadFrontier(open_nodes);
    sffgpu_forest_destroy(f);
    if (SaveParams <= P.saveOptions) this->saveParams(P.fileNames[SaveParams], st.iterations, st.solved != 0, stopTime - startingTime);
    if (SaveTSP <= P.saveOptions) this->saveTsp(P.fileNames[SaveTSP]);
    if (SaveFrontiers <= P.saveOptions) saveFrontiers(P.fileNames[SaveFrontiers], open_nodes);
  }
};
