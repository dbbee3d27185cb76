// primitives.h — source-compatible counterpart of the reference's src/primitives.h for the drop-in
// header set (include/sff/): Point / Range / Dimensions / FileStruct / helpers as src/main.cpp and the
// writers use them, and the node / tree API of the solvers - Node (Closest, Children, Root ...), Tree,
// DistanceHolder, SymmetricMatrix, PathNode, the Heap cost functions (src/primitives.h:440-655, :672-733).
// The growing forest itself lives inside libsffgpu; after (or during) a Solve() these types are the host-side
// view of it, filled from the arrays the C ABI returns (Solver::fillNodes in problemStruct.h).
#pragma once
#include <algorithm>
#include <cmath>
#include <deque>
#include <limits>
#include <map>
#include <regex>
#include <stdexcept>
#include <string>
#include <vector>

#include <flann/flann.hpp>   // include/sff/flann/flann.hpp: the FLANN surface of the solvers on top of libsffgpu

#ifndef DELIMITER_OUT
#define DELIMITER_OUT (" ")
#endif
#define CSV_DELIMITER (",")
#define CSV_DELIMITER_2 (";")
#define TSP_DELIMITER (" ")
#define TOLERANCE 1e-9
#define DEFAULT_THRES_MISS 3
#define DEFAULT_SAMP_DIST 1

enum Dimensions { D2 = 2, D3 = 6 };
enum FileType { Map, Obj };
struct FileStruct {
  std::string fileName;
  FileType type;
};

template <class T>
struct Range {
  T minX, maxX, minY, maxY, minZ, maxZ;
};

// reference src/primitives.h:86-275 (public surface used by the parser and the writers)
template <class T>
class Point {
 public:
  T Yaw{0}, Pitch{0}, Roll{0};
  Point() : c{0, 0, 0} {}
  Point(T x, T y, T z) : c{x, y, z} {}
  Point(T x, T y, T z, T yaw, T pitch, T roll) : Yaw{yaw}, Pitch{pitch}, Roll{roll}, c{x, y, z} {}
  // "[x; y; z]" scaled (src/primitives.h:104-114)
  Point(const std::string& s, T scale = 1) {
    static const std::regex r("\\[(\\-?[\\d.]+);\\s*(\\-?[\\d.]+);\\s*(\\-?[\\d.]+)\\]");
    std::smatch m;
    std::regex_search(s, m, r);
    if (m.size() != 4) throw std::invalid_argument("Unknown format of point");
    for (int i = 0; i < 3; ++i) c[i] = std::stod(m[i + 1]) * scale;
  }
  T x() const { return c[0]; }
  T y() const { return c[1]; }
  T z() const { return c[2]; }
  void set(T x, T y, T z) { c[0] = x; c[1] = y; c[2] = z; }
  void setPosition(unsigned pos, T v) {
    if (pos < 3) c[pos] = v; else if (pos == 3) Yaw = v; else if (pos == 4) Pitch = v; else if (pos == 5) Roll = v;
  }
  const T* operator()() const { return c; }
  T operator[](int i) const { return i < 3 ? c[i] : (i == 3 ? Yaw : (i == 4 ? Pitch : (i == 5 ? Roll : T(1)))); }
  friend bool operator==(const Point& a, const Point& b) {
    return a.c[0] == b.c[0] && a.c[1] == b.c[1] && a.c[2] == b.c[2] && a.Yaw == b.Yaw && a.Pitch == b.Pitch && a.Roll == b.Roll;
  }
  friend bool operator!=(const Point& a, const Point& b) { return !(a == b); }
  // scales the position, NOT the rotation (src/primitives.h:215-222)
  friend Point operator/(const Point& p, const T scale) {
    Point q{p};
    for (int i = 0; i < 3; ++i) q.c[i] /= scale;
    return q;
  }
  void toArray(double out[6]) const { for (int i = 0; i < 6; ++i) out[i] = (*this)[i]; }
  // 6-D metric (src/primitives.h:224-235): position differences this - other, angle differences other - this
  // wrapped ONCE into [-pi, pi) (:278-292); the summation order is the reference's, so the bits are too
  T distance(const Point& other) const {
    T sum = 0;
    for (int i = 0; i < 3; ++i) { const T d = c[i] - other.c[i]; sum += d * d; }
    const T mine[3] = {Yaw, Pitch, Roll}, theirs[3] = {other.Yaw, other.Pitch, other.Roll};
    for (int i = 0; i < 3; ++i) { const T d = wrapOnce(theirs[i] - mine[i]); sum += d * d; }
    return std::sqrt(sum);
  }
  // steer (src/primitives.h:237-250): the point `dist` away from this one towards `other`; angles are not
  // re-normalised
  Point getStateInDistance(const Point& other, const T dist) const {
    const T s = dist / distance(other);
    Point q;
    for (int i = 0; i < 3; ++i) q.c[i] = c[i] + (other.c[i] - c[i]) * s;
    q.Yaw = Yaw + wrapOnce(other.Yaw - Yaw) * s;
    q.Pitch = Pitch + wrapOnce(other.Pitch - Pitch) * s;
    q.Roll = Roll + wrapOnce(other.Roll - Roll) * s;
    return q;
  }
  // R = Rz(Yaw) Ry(Pitch) Rx(Roll) (src/primitives.h:252-262)
  void FillRotationMatrix(T (&m)[3][3]) const {
    const T cy = std::cos(Yaw), sy = std::sin(Yaw), cp = std::cos(Pitch), sp = std::sin(Pitch), cr = std::cos(Roll),
            sr = std::sin(Roll);
    m[0][0] = cy * cp; m[0][1] = cy * sp * sr - sy * cr; m[0][2] = cy * sp * cr + sy * sr;
    m[1][0] = sy * cp; m[1][1] = sy * sp * sr + cy * cr; m[1][2] = sy * sp * cr - cy * sr;
    m[2][0] = -sp;     m[2][1] = cp * sr;                m[2][2] = cp * cr;
  }
  static T wrapOnce(T a) {
    const T pi = (T)3.14159265358979323846;
    if (a < -pi) return a + 2 * pi;
    if (a >= pi) return a - 2 * pi;
    return a;
  }
  void printPosOnly(std::ostream& out) const { out << c[0] << DELIMITER_OUT << c[1] << DELIMITER_OUT << c[2]; }

 private:
  T c[3];
};

template <class T>
std::ostream& operator<<(std::ostream& out, const Point<T>& p) {
  return out << p.x() << DELIMITER_OUT << p.y() << DELIMITER_OUT << p.z() << DELIMITER_OUT << p.Yaw << DELIMITER_OUT
             << p.Pitch << DELIMITER_OUT << p.Roll;
}

template <class T, class R> class Tree;
template <class T, class R> struct DistanceHolder;
template <class T, class R> class Heap;
template <class T, class R = Point<T>> class Node;
template <class T> T Distance(Node<T, Point<T>>& node1, Node<T, Point<T>>& ref);
template <class T> T StarDistance(Node<T, Point<T>>& node1, Node<T, Point<T>>& ref);

// FLANN distance functor of the trees' indices (src/primitives.h:405-438).  The shipped reference functor
// overwrites its accumulator, so FLANN only ever saw the squared roll difference; this one
// is the metric the solvers mean - squared 6-D distance with wrapped angle terms - which is also what
// the index of include/sff/flann/flann.hpp evaluates on the GPU.
template <class T>
struct D6Distance {
  typedef bool is_vector_space_distance;
  typedef T ElementType;
  typedef typename flann::Accumulator<T>::Type ResultType;
  template <typename It1, typename It2>
  ResultType operator()(It1 a, It2 b, size_t size, ResultType /*worst_dist*/ = -1) const {
    ResultType sum = ResultType();
    for (size_t i = 0; i < size; ++i) sum += accum_dist(*a++, *b++, (int)i);
    return sum;
  }
  template <typename U, typename V>
  inline ResultType accum_dist(const U& a, const V& b, int dim) const {
    if (dim < 3) return (a - b) * (a - b);
    ResultType d = (ResultType)b - (ResultType)a;
    const ResultType pi = (ResultType)3.14159265358979323846;
    if (d < -pi) d += 2 * pi; else if (d >= pi) d -= 2 * pi;
    return d * d;
  }
};

// Node of a tree (src/primitives.h:440-498): same members and identity rules - ids come from a process-wide
// counter in creation order, IsRoot() is "DistanceToRoot == 0", a node's ExpandedRoot is inherited from the node
// it was expanded from.  ParentId / TreeId are this header set's additions (the flat view the writers index by).
template <class T, class R>
class Node {
 public:
  inline static char ThresholdMisses = DEFAULT_THRES_MISS;
  inline static double SamplingDistance = DEFAULT_SAMP_DIST;

  R Position;
  Tree<T, Node>* Root{nullptr};
  Tree<T, Node>* ExpandedRoot{nullptr};
  Node* Closest{nullptr};
  std::deque<Node*> Children;
  bool ForceChildren{false};
  T DistanceToClosest{0};
  T DistanceToRoot{0};
  std::map<Node*, T> VisibleNodes;
  int ParentId{-1}, TreeId{0};

  Node(R position, Tree<T, Node>* root, Node* closest, T distanceToClosest, T distanceToRoot, unsigned int iteration)
      : Position{position}, Root{root}, ExpandedRoot{closest ? closest->ExpandedRoot : root}, Closest{closest},
        DistanceToClosest{distanceToClosest}, DistanceToRoot{distanceToRoot}, id{next_id++}, generation{iteration} {
    if (closest) ParentId = closest->GetId();
  }
  friend bool operator==(const Node& a, const Node& b) { return a.id == b.id; }
  friend bool operator<(const Node& a, const Node& b) { return a.id < b.id; }
  const bool IsRoot() const { return DistanceToRoot == 0; }
  const int GetId() const { return (int)id; }
  const int GetNumNodes() const { return (int)next_id; }
  const unsigned int GetAge() const { return generation; }
  // the host-side view is rebuilt from the device's arrays (creation order): restart the id counter first
  static void ResetIds(unsigned first = 0) { next_id = first; }

 private:
  inline static unsigned next_id = 0;
  unsigned id;
  unsigned generation;
};

// src/primitives.h:672-677
template <typename T>
struct PathNode {
  T distanceFromStart{std::numeric_limits<T>::max()};
  int heapPosition{-1};
  Node<T, Point<T>>* previousPoint{nullptr};
};

// One tree (src/primitives.h:500-570): its nodes in insertion order (their position in `nodes` is their index
// in the tree's neighbour index), the index itself, priority frontiers, links to other trees, eaten trees.
template <class T, class R = Node<T>>
class Tree {
 public:
  inline static bool AStar = false;
  std::deque<R> nodes;
  R* Root{nullptr};
  flann::Index<D6Distance<float>>* flannIndex{nullptr};
  std::deque<float*> ptrToDel;
  std::deque<DistanceHolder<T, R>> links;
  std::deque<Heap<T, R>> frontiers;
  std::vector<bool> frontierFilter;
  std::deque<Tree*> eaten;

  Tree() : id{next_id++} {}
  Tree(const Tree&) = delete;              // nodes hold pointers into `nodes`: a tree stays where it is
  Tree& operator=(const Tree&) = delete;
  ~Tree() {
    delete flannIndex;
    for (float* p : ptrToDel) delete[] p;
  }
  friend bool operator==(const Tree& a, const Tree& b) { return a.id == b.id; }
  void AddFrontier(R* goal) {
    frontiers.emplace_back(nodes, goal, true, AStar ? StarDistance<T> : Distance<T>);
    frontierFilter.push_back(false);
  }
  const bool EmptyFrontiers() {
    bool none_left = true, all_filtered = true;
    for (auto& h : frontiers) none_left = none_left && h.empty();
    for (bool f : frontierFilter) all_filtered = all_filtered && f;
    return none_left || all_filtered;
  }
  void EnableFrontier() { std::fill(frontierFilter.begin(), frontierFilter.end(), false); }
  const int GetId() const { return (int)id; }
  static void ResetIds(unsigned first = 0) { next_id = first; }

 private:
  inline static unsigned next_id = 0;
  unsigned id;
};

// Upper-triangular storage of a symmetric table (src/primitives.h:572-596): (i, j) and (j, i) are one cell.
template <class T>
class SymmetricMatrix {
 public:
  SymmetricMatrix(const int size) : n{size}, cells((size_t)size * (size + 1) / 2) {}
  T& operator()(int i, int j) {
    const int r = i <= j ? i : j, c = i <= j ? j : i;       // row r holds columns r .. n-1
    return cells[(size_t)r * n - (size_t)r * (r - 1) / 2 + (c - r)];
  }
  const bool Exists(int i, int j) { return (*this)(i, j).Exists(); }

 private:
  int n;
  std::deque<T> cells;
};

// A connection between two nodes (border pair, link, root-to-root path): lower node id first, the plan runs
// from node1's side to node2's (src/primitives.h:598-655).
template <class T, class R>
struct DistanceHolder {
  R* node1{nullptr};
  R* node2{nullptr};
  T distance{std::numeric_limits<T>::max()};
  std::deque<R*> plan;

  DistanceHolder() {}
  DistanceHolder(R* first, R* second) { order(first, second); UpdateDistance(); }
  DistanceHolder(R* first, R* second, T dist) : distance{dist} { order(first, second); }
  DistanceHolder(R* first, R* second, T dist, std::deque<R*>& p) : distance{dist}, plan{p} {
    if (!order(first, second)) std::reverse(plan.begin(), plan.end());
  }
  friend bool operator<(const DistanceHolder& l, const DistanceHolder& r) { return l.distance < r.distance; }
  friend bool operator==(const DistanceHolder& l, const DistanceHolder& r) { return l.node1 == r.node1 && l.node2 == r.node2; }
  const bool Exists() const { return node1 != nullptr; }
  void UpdateDistance() {
    distance = node1->DistanceToRoot + node2->DistanceToRoot + node1->Position.distance(node2->Position);
  }

 private:
  bool order(R* first, R* second) {     // true when the arguments already came lower id first
    const bool kept = *first < *second;
    node1 = kept ? first : second;
    node2 = kept ? second : first;
    return kept;
  }
};

#include "heap.h"

// cost functions of the priority frontiers (src/primitives.h:726-733)
template <class T>
T Distance(Node<T, Point<T>>& node1, Node<T, Point<T>>& ref) { return node1.Position.distance(ref.Position); }
template <class T>
T StarDistance(Node<T, Point<T>>& node1, Node<T, Point<T>>& ref) {
  return 0.7 * node1.Position.distance(ref.Position) + 0.3 * node1.DistanceToRoot;
}

// src/primitives.h:680-697
inline int parseString(std::string& inp, std::string& outp1, std::string& outp2, std::string& delimiter) {
  size_t pos = inp.find(delimiter);
  if (pos != std::string::npos) {
    outp1 = inp.substr(0, pos);
    outp2 = inp.substr(pos + delimiter.size());
    return (int)pos;
  }
  outp1 = inp;
  outp2 = "";
  return -1;
}

// src/primitives.h:699-710
inline FileStruct prefixFileName(const FileStruct& path, const std::string& insert) {
  FileStruct r{path};
  auto pos = r.fileName.find_last_of("//");
  if (pos != std::string::npos) r.fileName.insert(pos + 1, insert); else r.fileName.insert(0, insert);
  return r;
}

inline std::string trim(const std::string& s) {
  const char* ws = " \n\r\t\f\v";
  size_t b = s.find_first_not_of(ws);
  if (b == std::string::npos) return "";
  return s.substr(b, s.find_last_not_of(ws) - b + 1);
}
