// lazy.h — LazyTSP<T,R> placeholder (reference src/lazy.h).  The reference's Lazy solver shells out to
// the non-public `obst_tsp` binary (src/lazy.h:93-98, README.md:14), which cannot exist here; the class
// is kept so that src/main.cpp links, and reports the situation in the reference's style.
#pragma once
#include "problemStruct.h"

template <class T, class R = Point<T>>
class LazyTSP : public Solver<T, R> {
 public:
  LazyTSP(Problem<T>& problem) : Solver<T, R>(problem) {}
  void Solve() override {
    std::cout << "LazyTSP: the Lazy solver needs the external TSP binary (" << this->problem.tspSolver
              << ") and is not part of the GPU hot path; use solver=\"sff\" or \"rrt\"\n";
    std::exit(1);
  }
};
