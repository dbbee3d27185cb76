// flann/flann.hpp — the FLANN surface the reference's solvers use (inner seam 1), on top of
// libsffgpu's exact neighbour index.  Put include/sff/ on the include path IN FRONT OF the vendored FLANN and the
// unchanged reference sources (#include <flann/flann.hpp> at src/primitives.h:25, src/forest.h:17, src/rrt.h:16)
// build against it; every call below then runs on the GPU:
//
//   flann::Index<D>(Matrix, KDTreeIndexParams) + buildIndex()    src/forest.h:72-73,99-100  src/rrt.h:57-58,76-77
//   Index::addPoints(Matrix)                                      src/forest.h:367  src/rrt.h:215,298   -> sffgpu_nodes_append
//   Index::radiusSearch(q, indices, dists, r2, SearchParams)      src/forest.h:266-267                 -> sffgpu_radius
//   Index::knnSearch(q, indices, dists, k, SearchParams)          src/forest.h:317  src/rrt.h:143,166,228 -> sffgpu_knn
//   flann::Matrix<T>(ptr, rows, cols), operator[], ptr()          all call sites
//   flann::Accumulator<T>::Type                                   src/primitives.h:409 (the D6Distance functor)
//
// What differs from the vendored library, on purpose: the search is EXACT (FLANN visits at
// most `checks` = 128 leaves of 4 randomised kd-trees) and uses the true 6-D metric with wrapped angle terms in
// fp64 on the float coordinates it is given (the Distance functor passed as template argument is not evaluated:
// the shipped one returns only the squared roll difference, src/primitives.h:416-424).  Results come sorted by
// (distance, index), `dists` holds squared distances like FLANN's L2-style functors, `radius` is a squared radius.
// Every Index is one tree of a process-wide node store (its points keep the order of insertion as their index).
// A boundary deliverable, not parity evidence: nothing in tests/ compares against a build that uses it.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <vector>

#include "../sff_gpu.h"

namespace flann {

template <typename T> struct Accumulator { typedef T Type; };
template <> struct Accumulator<unsigned char> { typedef float Type; };
template <> struct Accumulator<unsigned short> { typedef float Type; };
template <> struct Accumulator<unsigned int> { typedef float Type; };
template <> struct Accumulator<char> { typedef float Type; };
template <> struct Accumulator<short> { typedef float Type; };
template <> struct Accumulator<int> { typedef float Type; };

// caller-owned row-major view (the reference keeps the float rows alive itself: Tree::ptrToDel)
template <typename T>
class Matrix {
 public:
  typedef T type;
  size_t rows{0}, cols{0}, stride{0};
  Matrix() {}
  Matrix(T* data, size_t rows_, size_t cols_, size_t stride_ = 0)
      : rows{rows_}, cols{cols_}, stride{stride_ ? stride_ : cols_ * sizeof(T)}, data_{reinterpret_cast<unsigned char*>(data)} {}
  T* operator[](size_t row) const { return reinterpret_cast<T*>(data_ + row * stride); }
  T* ptr() const { return reinterpret_cast<T*>(data_); }

 private:
  unsigned char* data_{nullptr};
};

struct IndexParams {};
struct KDTreeIndexParams : IndexParams {
  explicit KDTreeIndexParams(int trees_ = 4) : trees{trees_} {}
  int trees;   // ignored: the GPU index is exact
};
struct SearchParams {
  explicit SearchParams(int checks_ = 32, float eps_ = 0, bool sorted_ = true) : checks{checks_}, eps{eps_}, sorted{sorted_} {}
  int checks;  // ignored: every stored point is considered
  float eps;
  bool sorted;
};

namespace sff_detail {
// process-wide store bookkeeping shared by all Index objects: which tree a store entry belongs to and its
// position inside that tree's index
struct Store {
  int next_tree = 0;
  int live = 0;
  std::vector<int32_t> local_of_global;
  bool started = false;
};
inline Store& store() { static Store s; return s; }
}  // namespace sff_detail

template <typename Distance>
class Index {
 public:
  typedef typename Distance::ElementType ElementType;
  typedef typename Distance::ResultType DistanceType;

  Index(const Matrix<ElementType>& features, const IndexParams& /*params*/, Distance = Distance())
      : first{features}, cols{features.cols} {
    sff_detail::Store& s = sff_detail::store();
    if (!s.started || s.live == 0) {   // first index of a solver run: start from an empty store
      sff_compat::check(sffgpu_nodes_reset(sff_compat::gpu(), 0), "nodes_reset");
      s.local_of_global.clear();
      s.next_tree = 0;
      s.started = true;
    }
    tree = s.next_tree++;
    ++s.live;
  }
  Index(const Index&) = delete;
  Index& operator=(const Index&) = delete;
  ~Index() { --sff_detail::store().live; }

  void buildIndex() {
    if (built) return;
    built = true;
    addPoints(first);
  }
  void addPoints(const Matrix<ElementType>& points, float /*rebuild_threshold*/ = 2) {
    const int n = (int)points.rows;
    if (n <= 0) return;
    std::vector<double> pos((size_t)n * 6, 0.0);
    std::vector<int32_t> tr((size_t)n, tree);
    for (int i = 0; i < n; ++i)
      for (size_t k = 0; k < points.cols && k < 6; ++k) pos[6 * (size_t)i + k] = (double)points[i][k];
    sffgpu_ctx* c = sff_compat::gpu();
    const int base = sffgpu_nodes_count(c);
    sff_compat::check(base, "nodes_count");
    sff_compat::check(sffgpu_nodes_append(c, pos.data(), tr.data(), n), "nodes_append");
    sff_detail::Store& s = sff_detail::store();
    s.local_of_global.resize((size_t)base + n, -1);
    for (int i = 0; i < n; ++i) {
      s.local_of_global[(size_t)base + i] = (int32_t)global_of_local.size();
      global_of_local.push_back(base + i);
    }
  }
  size_t size() const { return global_of_local.size(); }
  size_t veclen() const { return cols; }

  // every stored point of this index with squared distance < radius, nearest first; returns the total count
  int radiusSearch(const Matrix<ElementType>& queries, std::vector<std::vector<int>>& indices,
                   std::vector<std::vector<DistanceType>>& dists, float radius, const SearchParams& /*params*/) const {
    const int nq = (int)queries.rows;
    indices.assign(nq, {});
    dists.assign(nq, {});
    if (nq == 0 || global_of_local.empty()) return 0;
    std::vector<double> q = pack(queries);
    std::vector<double> r((size_t)nq, std::sqrt((double)radius));
    std::vector<int32_t> tr((size_t)nq, tree), cnt((size_t)nq, 0);
    int cap = 128, total = 0;
    std::vector<int32_t> idx;
    std::vector<double> dd;
    while (true) {
      idx.assign((size_t)nq * cap, -1);
      dd.assign((size_t)nq * cap, 0.0);
      sff_compat::check(sffgpu_radius(sff_compat::gpu(), q.data(), nq, r.data(), tr.data(), nullptr, idx.data(), dd.data(),
                                      cnt.data(), cap), "radius");
      int most = 0;
      for (int v : cnt) most = v > most ? v : most;
      if (most <= cap) break;
      cap = most;
    }
    for (int i = 0; i < nq; ++i) {
      total += cnt[i];
      unpack(idx.data() + (size_t)i * cap, dd.data() + (size_t)i * cap, cnt[i], indices[i], dists[i]);
    }
    return total;
  }
  // the knn nearest stored points of this index (fewer when it holds fewer)
  int knnSearch(const Matrix<ElementType>& queries, std::vector<std::vector<int>>& indices,
                std::vector<std::vector<DistanceType>>& dists, size_t knn, const SearchParams& /*params*/) const {
    const int nq = (int)queries.rows, k = (int)knn;
    indices.assign(nq, {});
    dists.assign(nq, {});
    if (nq == 0 || k <= 0 || global_of_local.empty()) return 0;
    std::vector<double> q = pack(queries);
    std::vector<int32_t> tr((size_t)nq, tree), cnt((size_t)nq, 0), idx((size_t)nq * k, -1);
    std::vector<double> dd((size_t)nq * k, 0.0);
    sff_compat::check(sffgpu_knn(sff_compat::gpu(), q.data(), nq, k, tr.data(), nullptr, idx.data(), dd.data(), cnt.data()), "knn");
    int total = 0;
    for (int i = 0; i < nq; ++i) {
      total += cnt[i];
      unpack(idx.data() + (size_t)i * k, dd.data() + (size_t)i * k, cnt[i], indices[i], dists[i]);
    }
    return total;
  }

 private:
  Matrix<ElementType> first;
  size_t cols;
  int tree{0};
  bool built{false};
  std::vector<int32_t> global_of_local;

  std::vector<double> pack(const Matrix<ElementType>& m) const {
    std::vector<double> q(m.rows * 6, 0.0);
    for (size_t i = 0; i < m.rows; ++i)
      for (size_t k = 0; k < m.cols && k < 6; ++k) q[6 * i + k] = (double)m[i][k];
    return q;
  }
  void unpack(const int32_t* idx, const double* dd, int n, std::vector<int>& out_i, std::vector<DistanceType>& out_d) const {
    const std::vector<int32_t>& map = sff_detail::store().local_of_global;
    out_i.resize(n);
    out_d.resize(n);
    for (int j = 0; j < n; ++j) {
      out_i[j] = map[(size_t)idx[j]];
      out_d[j] = (DistanceType)(dd[j] * dd[j]);
    }
  }
};

}  // namespace flann
