// lazy.h — source-compatible LazyTSP<T,R> (reference src/lazy.h:25-47) on top of libsffgpu.
//
// The Lazy solver alternates between an external TSP solver run on the current root-to-root distance matrix
// (src/lazy.h:72-118: TSPLIB file out, `system(<tspSolver> ...)`, one result line in) and a single-tree RRT / RRT*
// for every tour edge that has no plan yet (runRRT, :160-284).  The inner planner is the hot part and runs on the GPU
// through the RRT session's lazy_edge mode (include/sffgpu.h); all edges draw from ONE engine stream like the
// reference's single RandGen (rng_skip).  The outer loop, the result parser and the three writers are restated here
// with the reference's file formats and messages.  The TSP binary itself (`obst_tsp`, README.md:14) is not public:
// whatever `<TSP path=...>` names is executed, exactly as the reference does.
#pragma once
#include <cstring>
#include <tuple>

#include "problemStruct.h"

#define TEMP_TSP "tempTsp.tsp"
#define TEMP_RESULT "tempTsp.result"

template <class T, class R = Point<T>>
class LazyTSP : public Solver<T, R> {
 public:
  LazyTSP(Problem<T>& problem) : Solver<T, R>(problem), numRoots{problem.GetNumRoots()} {
    // src/lazy.h:50-62: the roots are nodes 0 .. numRoots-1, the matrix starts with straight-line distances
    for (int i = 0; i < numRoots; ++i) nodePos.push_back(problem.roots[i]);
    edges.resize((size_t)numRoots * numRoots);
    for (int i = 0; i < numRoots; ++i)
      for (int j = i + 1; j < numRoots; ++j) edge(i, j).distance = nodePos[i].distance(nodePos[j]);
  }

  void Solve() override {
    Problem<T>& P = this->problem;
    P.environment.upload();
    const char* s = std::getenv("SFF_SEED");
    seed = s ? std::strtoull(s, nullptr, 10) : (uint64_t)std::chrono::high_resolution_clock::now().time_since_epoch().count();
    T prevDist{-1}, newDist{0};
    std::string resultLine;
    std::deque<std::tuple<int, int>> selectedEdges;
    auto startingTime = std::chrono::high_resolution_clock::now();
    FileStruct tempTsp;
    tempTsp.fileName = TEMP_TSP;
    tempTsp.type = Map;
    bool solved{false};
    int iter{0};
    while (!solved && iter != numRoots * P.maxIterations) {                      // src/lazy.h:87
      selectedEdges.clear();
      prevDist = newDist;
      // run TSP = create file, execute, read output (:91-115)
      std::string id{"id_" + std::to_string(P.iteration) + "_"};
      FileStruct runFile{prefixFileName(tempTsp, id)};
      this->saveTsp(runFile);
      std::string command{P.tspSolver};
      command.append(" --map-type=TSP_FILE --use-path-files-folder=false --use-prm=false --tsp-solver=");
      command.append(P.tspType);
      command.append(" --problem=");
      command.append(runFile.fileName);
      if (system(command.c_str()) != 0) { /* the reference ignores the exit status; the result file decides */ }
      std::string resultName{TEMP_RESULT};
      std::ifstream resFile{resultName.insert(0, id), std::ios::in};
      if (!resFile.good()) {
        std::cout << "Lazy TSP: result file error";
        return;
      }
      getline(resFile, resultLine);
      processResults(resultLine, selectedEdges, newDist);
      newDist = 0;
      // run RRT for selected edges, recompute new distance (:120-131)
      for (auto& pair : selectedEdges) {
        int first, second;
        std::tie(first, second) = pair;
        Edge& e = edge(first, second);
        if (e.plan.empty()) runRRT(first, second, iter);
        newDist += e.distance;
      }
      solved = (newDist >= prevDist - TOLERANCE && newDist <= prevDist + TOLERANCE);
    }
    auto stopTime = std::chrono::high_resolution_clock::now();
    if (SaveRaw <= P.saveOptions) savePaths(P.fileNames[SaveRaw], selectedEdges);
    // (smoothPaths is empty in the reference, src/lazy.h:155-158)
    if (SaveParams <= P.saveOptions) saveParams(P.fileNames[SaveParams], iter, solved, stopTime - startingTime, selectedEdges);
    if (SaveTSP <= P.saveOptions) this->saveTsp(P.fileNames[SaveTSP]);
  }

  // what the tests read
  T edgeDistance(int i, int j) { return edge(i, j).distance; }
  const std::vector<int>& edgePlan(int i, int j) { return edge(i, j).plan; }
  size_t numNodes() const { return nodePos.size(); }

 private:
  struct Edge {                      // DistanceHolder of the root pair: node1 = the lower root (src/primitives.h:618-626)
    T distance{std::numeric_limits<T>::max()};
    std::vector<int> plan;           // global node ids, node1's side first
  };
  int numRoots;
  uint64_t seed{0}, rngDraws{0};     // one engine stream for all edges (the reference's single RandGen)
  std::vector<Point<T>> nodePos;     // allNodes: the roots, then every runRRT's nodes in creation order
  std::vector<Edge> edges;
  Edge& edge(int i, int j) { return edges[(size_t)std::min(i, j) * numRoots + std::max(i, j)]; }

  // src/lazy.h:160-284 on the GPU: one tree from the lower root towards the higher one
  void runRRT(int first, int second, int& iterations) {
    Problem<T>& P = this->problem;
    const int a = std::min(first, second), b = std::max(first, second);
    sffgpu_rrt_cfg cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.dim = P.dimension;
    cfg.optimize = this->optimize;
    nodePos[b].toArray(cfg.goal);
    const Range<T>& l = P.environment.limits;
    const double lim[6] = {l.minX, l.maxX, l.minY, l.maxY, l.minZ, l.maxZ};
    std::memcpy(cfg.limits, lim, sizeof lim);
    cfg.dist_tree = this->treeDistance;
    cfg.sampling_dist = Node<T, R>::SamplingDistance;
    cfg.max_iterations = P.maxIterations;
    cfg.seed = seed;
    cfg.lazy_edge = 1;
    cfg.rng_skip = rngDraws;
    double root[6];
    nodePos[a].toArray(root);
    sffgpu_rrt* r = nullptr;
    sff_compat::check(sffgpu_rrt_create(sff_compat::gpu(), &cfg, root, 1, &r), "lazy rrt");
    sff_compat::check(sffgpu_rrt_run(r, 0), "lazy rrt run");
    sffgpu_rrt_stats st;
    sffgpu_rrt_get_stats(r, &st);
    rngDraws = st.rng_draws;
    // the tree's nodes join allNodes (its own copy of the start first, :166-167)
    const int base = (int)nodePos.size();
    std::vector<double> pos((size_t)st.n_nodes * 6);
    sffgpu_rrt_get_nodes(r, pos.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    for (int k = 0; k < st.n_nodes; ++k)
      nodePos.emplace_back(pos[6 * k], pos[6 * k + 1], pos[6 * k + 2], pos[6 * k + 3], pos[6 * k + 4], pos[6 * k + 5]);
    Edge& e = edge(a, b);
    if (st.solved) {                                                             // :258-273
      e.distance = (T)st.lazy_distance;
      const int len = sffgpu_rrt_lazy_plan(r, nullptr, 0);
      std::vector<int32_t> ids(len);
      sffgpu_rrt_lazy_plan(r, ids.data(), len);
      e.plan.clear();
      for (int32_t k : ids) e.plan.push_back(base + k);
      e.plan.push_back(b);                                                       // the goal itself ends the plan
    } else {
      e.distance = std::numeric_limits<T>::max();                                // :279-281
    }
    sffgpu_rrt_destroy(r);
    iterations += st.iterations;
  }

  // src/lazy.h:286-300: "<length> , <p0> , <p1> , ... , <pN>" -> N = numRoots tour edges
  void processResults(std::string& line, std::deque<std::tuple<int, int>>& edgePairs, T& pathLength) {
    std::string delimiter{" , "}, parsedPart;
    parseString(line, parsedPart, line, delimiter);
    pathLength = std::stod(parsedPart);
    parseString(line, parsedPart, line, delimiter);
    int prevPoint{std::stoi(parsedPart)};
    for (int i = 0; i < numRoots; ++i) {
      parseString(line, parsedPart, line, delimiter);
      int actPoint{std::stoi(parsedPart)};
      edgePairs.push_back(std::tuple<int, int>(prevPoint, actPoint));
      prevPoint = actPoint;
    }
  }

  // src/lazy.h:302-330 (TSPLIB, LOWER_DIAG_ROW over ALL roots, empty COMMENT)
  void saveTsp(const FileStruct file) override {
    std::ofstream f;
    if (!this->open(f, file, "Saving TSP file")) return;
    f << "NAME: " << this->problem.id << "\nCOMMENT:\nTYPE: TSP\nDIMENSION: " << numRoots
      << "\nEDGE_WEIGHT_TYPE : EXPLICIT\nEDGE_WEIGHT_FORMAT : LOWER_DIAG_ROW\nEDGE_WEIGHT_SECTION\n";
    for (int i = 0; i < numRoots; ++i) {
      for (int j = 0; j < i; ++j) f << edge(i, j).distance / this->problem.environment.ScaleFactor << TSP_DELIMITER;
      f << "0\n";
    }
  }
  // src/lazy.h:332-380: the plans of the selected tour edges
  void savePaths(const FileStruct file, const std::deque<std::tuple<int, int>>& selectedPaths) {
    std::ofstream f;
    if (!this->open(f, file, "Saving paths")) return;
    const T sf = this->problem.environment.ScaleFactor;
    if (file.type == Obj) {
      f << "o Paths\n";
      for (const Point<T>& p : nodePos) {
        f << "v" << DELIMITER_OUT;
        (p / sf).printPosOnly(f);
        f << "\n";
      }
    }
    for (auto& pair : selectedPaths) {
      int first, second;
      std::tie(first, second) = pair;
      const std::vector<int>& plan = edge(first, second).plan;
      for (size_t k = 0; k + 1 < plan.size(); ++k) {
        if (file.type == Obj) f << "l" << DELIMITER_OUT << plan[k] + 1 << DELIMITER_OUT << plan[k + 1] + 1 << "\n";
        else f << nodePos[plan[k]] / sf << DELIMITER_OUT << nodePos[plan[k + 1]] / sf << "\n";
      }
      if (file.type != Obj) f << "\n";
    }
  }
  // src/lazy.h:382-425 (append mode): the tour and its edge lengths
  void saveParams(const FileStruct file, const int iterations, const bool solved,
                  const std::chrono::duration<double> elapsedTime, const std::deque<std::tuple<int, int>>& selectedEdges) {
    std::ofstream f;
    if (!this->open(f, file, "Saving parameters", std::ios_base::app)) return;
    f << this->problem.id << CSV_DELIMITER << this->problem.iteration << CSV_DELIMITER << iterations << CSV_DELIMITER
      << (solved ? "solved" : "unsolved") << CSV_DELIMITER << "[";
    int k = 0;
    for (auto& pair : selectedEdges) {
      f << std::get<0>(pair);
      if (++k != numRoots) f << CSV_DELIMITER_2;
    }
    f << "]" << CSV_DELIMITER << "[";
    k = 0;
    for (auto& pair : selectedEdges) {
      f << edge(std::get<0>(pair), std::get<1>(pair)).distance / this->problem.environment.ScaleFactor;
      if (++k != numRoots) f << CSV_DELIMITER_2;
    }
    f << "]" << CSV_DELIMITER << elapsedTime.count() << "\n";
  }
};
