// problemStruct.h — source-compatible Problem<T> / Solver<T,R> (reference src/problemStruct.h) for
// the drop-in header set.  Problem carries exactly the fields src/main.cpp::parseFile writes; the
// Solver base keeps the post-processing surface (tree / goal / path / params / TSP writers, text
// formats per SURVEY.md Appendix D) and fills it from the arrays libsffgpu returns.
#pragma once
#include <chrono>
#include <deque>
#include <fstream>
#include <iostream>
#include <map>
#include <string>
#include <vector>

#include "environment.h"
#include "primitives.h"

enum SaveOptions { None = 0, SaveConcurrent = 1, SaveGoals = 2, SaveTree = 4, SaveRaw = 8, SaveSmooth = 16,
                   SaveParams = 32, SaveTSP = 64, SaveFrontiers = 128, Invalid = 256 };
enum SolverType { SFF, RRT, Lazy };

template <class T>
struct Problem {   // src/problemStruct.h:43-88
  int iteration{0};
  Dimensions dimension{D3};
  SolverType solver;
  bool optimal;
  bool smoothing;
  Environment<T> environment;
  std::deque<Point<T>> roots;
  Point<T> goal;
  bool hasGoal{false};
  bool autoRange{false};
  T distTree;
  T collisionDist;
  int maxIterations;
  T priorityBias{0};
  int saveTreeIter{0};
  int saveFrontiersIter{0};
  SaveOptions saveOptions{None};
  std::map<SaveOptions, FileStruct> fileNames;
  std::string id{"Solver"};
  std::string tspSolver;
  std::string tspType;
  int GetNumRoots() { return hasGoal ? (int)roots.size() + 1 : (int)roots.size(); }
};

inline SaveOptions operator|(SaveOptions a, SaveOptions b) { return static_cast<SaveOptions>(static_cast<int>(a) | static_cast<int>(b)); }
// "flag a is active in b" (src/problemStruct.h:102-104)
inline bool operator<=(SaveOptions a, SaveOptions b) { return (static_cast<int>(b) & static_cast<int>(a)) == static_cast<int>(a); }

template <class T, class R = Point<T>>
class Solver {
 public:
  Solver(Problem<T>& p) : problem{p}, optimize{p.optimal}, usePriority{p.priorityBias != 0}, treeDistance{p.distTree},
                          env{p.environment} {}
  virtual ~Solver() {}
  virtual void Solve() = 0;

 protected:
  Problem<T>& problem;
  bool optimize;
  bool usePriority;
  T treeDistance;
  Environment<T>& env;

  // The reference's view of the result (src/problemStruct.h:124-127): nodes in global creation order, the trees
  // that own them, the root-to-root connections and the connected trees - rebuilt from the arrays the C ABI
  // returns (fillNodes / fillPaths below).  Node ids are the global creation order, so `allNodes[id]` is node id.
  std::deque<Node<T, R>*> allNodes;
  std::deque<Tree<T, Node<T, R>>> trees;
  SymmetricMatrix<DistanceHolder<T, Node<T, R>>> neighboringMatrix{0};
  std::deque<Tree<T, Node<T, R>>*> connectedTrees;

  // flat companions the writers index by tree id
  int numTrees{0};
  std::vector<int> connectedIds;               // tree ids, in the order libsffgpu reports them
  std::vector<double> pathCost;                // numTrees x numTrees path costs, DBL_MAX = none
  std::vector<std::vector<int>> plans;         // node-id plan per (i, j), i < j, row-major

  double costOf(int i, int j) const { return pathCost[(size_t)i * numTrees + j]; }
  const std::vector<int>& planOf(int i, int j) const { return plans[(size_t)std::min(i, j) * numTrees + std::max(i, j)]; }

  // tree = Node::Root of every node; holder (optional) = the tree whose list holds it now (RRT merges, Node::ExpandedRoot)
  void fillNodes(int n, int n_trees, const double* pos, const int32_t* parent, const int32_t* tree, const int32_t* iter,
                 const double* cost, const double* dpar, const int32_t* holder = nullptr) {
    allNodes.clear();
    connectedTrees.clear();
    trees.clear();
    Tree<T, Node<T, R>>::ResetIds();
    Node<T, R>::ResetIds();
    numTrees = n_trees;
    for (int t = 0; t < n_trees; ++t) trees.emplace_back();
    for (int i = 0; i < n; ++i) {     // creation order: Node ids come out as i
      Tree<T, Node<T, R>>& t = trees[tree[i]];
      t.nodes.emplace_back(R(pos[6 * i], pos[6 * i + 1], pos[6 * i + 2], pos[6 * i + 3], pos[6 * i + 4], pos[6 * i + 5]), &t,
                           nullptr, (T)dpar[i], (T)cost[i], (unsigned)iter[i]);
      Node<T, R>& nd = t.nodes.back();
      nd.TreeId = tree[i];
      nd.ParentId = parent[i];
      if (parent[i] < 0 && !t.Root) t.Root = &nd;
      allNodes.push_back(&nd);
    }
    for (int i = 0; i < n; ++i) {     // parents second: SFF* / RRT* rewiring can make a younger node the parent
      Node<T, R>* nd = allNodes[i];
      if (parent[i] >= 0) {
        nd->Closest = allNodes[parent[i]];
        nd->Closest->Children.push_back(nd);
      }
      nd->ExpandedRoot = &trees[holder ? holder[i] : tree[i]];
    }
  }
  // root-to-root connections: cost matrix + plans -> neighboringMatrix / connectedTrees
  void fillPaths(const std::vector<int>& connected) {
    connectedIds = connected;
    connectedTrees.clear();
    for (int t : connected) connectedTrees.push_back(&trees[t]);
    neighboringMatrix = SymmetricMatrix<DistanceHolder<T, Node<T, R>>>(numTrees);
    for (int i = 0; i < numTrees; ++i)
      for (int j = i + 1; j < numTrees; ++j) {
        const std::vector<int>& pl = planOf(i, j);
        if (pl.empty()) continue;
        std::deque<Node<T, R>*> nodes;
        for (int id : pl) nodes.push_back(allNodes[id]);
        neighboringMatrix(i, j) = DistanceHolder<T, Node<T, R>>(nodes.front(), nodes.back(), (T)costOf(i, j), nodes);
      }
  }
  bool open(std::ofstream& f, const FileStruct& file, const char* what, std::ios_base::openmode mode = std::ios_base::out) {
    std::cout << what << "\n";
    f.open(file.fileName.c_str(), mode);
    if (!f.good()) { std::cout << "Cannot create file at: " << file.fileName << "\n"; return false; }
    return true;
  }

  // src/problemStruct.h:264-295: roots (and the goal) of every tree
  virtual void saveCities(const FileStruct file) {
    std::ofstream f;
    if (!open(f, file, "Saving points")) return;
    if (file.type == Obj) f << "o Points\n";
    for (const Node<T, R>* n : allNodes)
      if (n->ParentId < 0) {
        if (file.type == Obj) f << "v" << DELIMITER_OUT;
        f << n->Position / problem.environment.ScaleFactor << "\n";
      }
  }
  // src/problemStruct.h:297-341: trees in creation order, nodes in per-tree insertion order
  virtual void saveTrees(const FileStruct file) {
    std::ofstream f;
    if (!open(f, file, "Saving trees")) return;
    if (file.type == Obj) {
      f << "o Trees\n";
      for (const Node<T, R>* n : allNodes) {
        f << "v" << DELIMITER_OUT;
        (n->Position / problem.environment.ScaleFactor).printPosOnly(f);
        f << "\n";
      }
      for (auto& t : trees)
        for (const Node<T, R>& n : t.nodes)
          if (!n.IsRoot()) f << "l" << DELIMITER_OUT << n.GetId() + 1 << DELIMITER_OUT << n.Closest->GetId() + 1 << "\n";
    } else {
      f << "#X1 Y1 Z1 Yaw1 Pitch1 Roll1 X2 Y2 Z2 Yaw2 Pitch2 Roll2 TreeID IterationOfCreation\n";
      for (auto& t : trees)
        for (const Node<T, R>& n : t.nodes)
          if (!n.IsRoot())
            f << n.Position / problem.environment.ScaleFactor << DELIMITER_OUT
              << n.Closest->Position / problem.environment.ScaleFactor << DELIMITER_OUT << n.Root->GetId() << DELIMITER_OUT
              << n.GetAge() << "\n";
    }
  }
  // src/problemStruct.h:470-527
  virtual void savePaths(const FileStruct file) {
    std::ofstream f;
    if (!open(f, file, "Saving paths")) return;
    if (file.type == Obj) {
      f << "o Paths\n";
      for (const Node<T, R>* n : allNodes) {
        f << "v" << DELIMITER_OUT;
        (n->Position / problem.environment.ScaleFactor).printPosOnly(f);
        f << "\n";
      }
    }
    for (int i = 0; i < numTrees; ++i)
      for (int j = i + 1; j < numTrees; ++j) {
        const std::vector<int>& plan = planOf(i, j);
        if (plan.empty()) continue;
        for (size_t k = 0; k + 1 < plan.size(); ++k) {
          if (file.type == Obj) f << "l" << DELIMITER_OUT << plan[k] + 1 << DELIMITER_OUT << plan[k + 1] + 1 << "\n";
          else f << allNodes[plan[k]]->Position / problem.environment.ScaleFactor << DELIMITER_OUT
                 << allNodes[plan[k + 1]]->Position / problem.environment.ScaleFactor << "\n";
        }
        if (file.type != Obj) f << "\n";
      }
  }
  // src/problemStruct.h:391-429 (append mode)
  virtual void saveParams(const FileStruct file, const int iterations, const bool solved,
                          const std::chrono::duration<double> elapsedTime) {
    std::ofstream f;
    if (!open(f, file, "Saving parameters", std::ios_base::app)) return;
    f << problem.id << CSV_DELIMITER << problem.iteration << CSV_DELIMITER << iterations << CSV_DELIMITER
      << (solved ? "solved" : "unsolved") << CSV_DELIMITER << "[";
    const int nc = (int)connectedIds.size();
    for (int i = 0; i < nc; ++i) {
      f << connectedIds[i];
      if (i + 1 != nc) f << CSV_DELIMITER_2;
    }
    f << "]" << CSV_DELIMITER << "[";
    for (int i = 0; i < nc; ++i)
      for (int j = 0; j < i; ++j) {
        f << costOf(connectedIds[i], connectedIds[j]) / problem.environment.ScaleFactor;
        if (i + 1 != nc || j + 1 != i) f << CSV_DELIMITER_2;
      }
    f << "]" << CSV_DELIMITER << elapsedTime.count() << "\n";
  }
  // src/problemStruct.h:431-468 (TSPLIB, LOWER_DIAG_ROW)
  virtual void saveTsp(const FileStruct file) {
    std::ofstream f;
    if (!open(f, file, "Saving TSP file")) return;
    const int nc = (int)connectedIds.size();
    f << "NAME: " << problem.id << "\nCOMMENT: ";
    for (int i = 0; i < nc; ++i) {
      f << connectedIds[i];
      if (i + 1 != nc) f << TSP_DELIMITER;
    }
    f << "\nTYPE: TSP\nDIMENSION: " << nc << "\nEDGE_WEIGHT_TYPE : EXPLICIT\nEDGE_WEIGHT_FORMAT : LOWER_DIAG_ROW\nEDGE_WEIGHT_SECTION\n";
    for (int i = 0; i < nc; ++i) {
      for (int j = 0; j < i; ++j) f << costOf(connectedIds[i], connectedIds[j]) / problem.environment.ScaleFactor << TSP_DELIMITER;
      f << "0\n";
    }
  }
};
