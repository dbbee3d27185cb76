// rrt.h — source-compatible RapidExpTree<T,R> (reference src/rrt.h:25-44) on top of libsffgpu's
// RRT / RRT* / Multi-T-RRT session.
#pragma once
#include <cstring>

#include "problemStruct.h"

template <class T, class R = Point<T>>
class RapidExpTree : public Solver<T, R> {
 public:
  RapidExpTree(Problem<T>& problem) : Solver<T, R>(problem) {}

  void Solve() override {
    Problem<T>& P = this->problem;
    P.environment.upload();
    sffgpu_rrt_cfg cfg;
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
    cfg.priority_bias = P.priorityBias;
    cfg.max_iterations = P.maxIterations;
    const char* s = std::getenv("SFF_SEED");
    cfg.seed = s ? std::strtoull(s, nullptr, 10)
                 : (uint64_t)std::chrono::high_resolution_clock::now().time_since_epoch().count();
    std::vector<double> roots;
    for (const Point<T>& p : P.roots) {
      double a[6];
      p.toArray(a);
      roots.insert(roots.end(), a, a + 6);
    }
    sffgpu_rrt* r = nullptr;
    sff_compat::check(sffgpu_rrt_create(sff_compat::gpu(), &cfg, roots.data(), (int)P.roots.size(), &r), "rrt");
    sffgpu_rrt_stats st;
    auto loadNodes = [&]() {
      sffgpu_rrt_get_stats(r, &st);
      const int n = st.n_nodes;
      std::vector<double> pos((size_t)n * 6), cost(n), dpar(n);
      std::vector<int32_t> parent(n), tree(n), root(n), iter(n);
      sffgpu_rrt_get_nodes(r, pos.data(), parent.data(), tree.data(), root.data(), iter.data(), cost.data(), dpar.data());
      this->fillNodes(n, P.GetNumRoots(), pos.data(), parent.data(), root.data(), iter.data(), cost.data(), dpar.data(),
                      tree.data());
    };
    auto startingTime = std::chrono::high_resolution_clock::now();   // src/rrt.h:90
    if (P.saveTreeIter == 0) {
      sff_compat::check(sffgpu_rrt_run(r, 0), "rrt run");
    } else {
      // saveIterCheck (src/rrt.h:98, src/problemStruct.h:256-261): the engine runs exactly up to the next multiple
      // of saveTreeIter, so the "iter_<k>_" dumps are the reference's snapshots after iteration k
      while (true) {
        sffgpu_rrt_get_stats(r, &st);
        if (st.solved || st.iterations >= P.maxIterations) break;
        const int before = st.iterations;
        sff_compat::check(sffgpu_rrt_run(r, P.saveTreeIter - before % P.saveTreeIter), "rrt run");
        sffgpu_rrt_get_stats(r, &st);
        if (st.iterations == before) break;
        if (st.iterations % P.saveTreeIter == 0) {
          loadNodes();
          this->saveTrees(prefixFileName(P.fileNames[SaveTree], "iter_" + std::to_string(st.iterations) + "_"));
        }
      }
    }
    auto stopTime = std::chrono::high_resolution_clock::now();       // :100
    loadNodes();
    this->pathCost.assign((size_t)this->numTrees * this->numTrees, 1.7976931348623157e308);
    std::vector<int32_t> conn(this->numTrees);
    int nc = sffgpu_rrt_paths(r, this->pathCost.data(), conn.data(), this->numTrees);   // getConnectedTrees + getPaths
    const std::vector<int> connected(conn.begin(), conn.begin() + (nc > 0 ? nc : 0));
    this->plans.assign((size_t)this->numTrees * this->numTrees, {});
    for (int i = 0; i < this->numTrees; ++i)
      for (int j = i + 1; j < this->numTrees; ++j) {
        int len = sffgpu_rrt_path_plan(r, i, j, nullptr, 0);
        if (len <= 0) continue;
        std::vector<int32_t> ids(len);
        sffgpu_rrt_path_plan(r, i, j, ids.data(), len);
        this->plans[(size_t)i * this->numTrees + j].assign(ids.begin(), ids.end());
      }
    this->fillPaths(connected);
    // smoothPaths (src/rrt.h:113-118, :354-379) shortens the plans held by the central tree's links; the matrix
    // the writers read keeps the copies made in getPaths (:350), so the "smooth" file repeats the raw paths
    if (P.smoothing) sff_compat::check(sffgpu_rrt_smooth_paths(r) < 0 ? -1 : 0, "rrt smoothing");
    sffgpu_rrt_destroy(r);
    if (SaveGoals <= P.saveOptions) this->saveCities(P.fileNames[SaveGoals]);
    if (SaveTree <= P.saveOptions) this->saveTrees(P.fileNames[SaveTree]);
    if (SaveRaw <= P.saveOptions) this->savePaths(P.fileNames[SaveRaw]);
    if (P.smoothing && SaveSmooth <= P.saveOptions) this->savePaths(P.fileNames[SaveSmooth]);
    if (SaveParams <= P.saveOptions) this->saveParams(P.fileNames[SaveParams], st.iterations, st.solved != 0, stopTime - startingTime);
    if (SaveTSP <= P.saveOptions) this->saveTsp(P.fileNames[SaveTSP]);
  }
};
