// heap.h — Heap<T, R> of the drop-in header set: the priority frontier of a tree (reference src/heap.h:32-65
// for the interface, :107-238 for the ordering rules that decide which node a biased pick returns).
//
// A binary min-heap of element POINTERS keyed by costFunction(*element, *refPoint).  The interface, the sift
// rules (strict `>` comparisons, left child preferred on ties, pop-at-index moves the LAST element into the hole and
// sifts it up when it is cheaper than the removed one, down otherwise) and therefore the array order after any
// sequence of operations are those of the reference; tests/golden/ref_types.json pins them to the reference's own
// header.  The implementation is this repository's: one flat array, costs cached next to the pointers (the
// reference keeps them in a std::map keyed by pointer and recomputes them on every move).
#pragma once
#include <deque>
#include <limits>
#include <map>
#include <vector>

template <class T>
struct PathNode;   // primitives.h
template <class T>
class Point;
template <class T, class R>
class Node;
template <class T>
T Distance(Node<T, Point<T>>& node1, Node<T, Point<T>>& ref);   // default cost: metric distance to refPoint

template <class T, class R>
class Heap {
 public:
  typedef T (*CostFunction)(R&, R&);
  R* refPoint{nullptr};
  std::map<R*, PathNode<T>> pathPoints;   // position / cost bookkeeping, kept for API compatibility

  Heap() {}
  // heap over the elements of `data` (by address), src/heap.h:72-83
  Heap(std::deque<R>& data, R* goalNode, bool calculateCost = true, CostFunction costFunc = Distance)
      : refPoint{goalNode}, cost_fn{costFunc}, calc{calculateCost} {
    for (R& e : data) items.push_back(&e);
    refresh_all();
    sort();
  }
  // heap over caller-owned pointers; the caller's deque is re-ordered in place (src/heap.h:85-94)
  Heap(std::deque<R*>* data, R* goalNode, bool calculateCost = true, CostFunction costFunc = Distance)
      : refPoint{goalNode}, external{data}, cost_fn{costFunc}, calc{calculateCost} {
    items.assign(data->begin(), data->end());
    refresh_all();
    sort();
  }

  void push(R* v) {
    items.push_back(v);
    note(items.size() - 1);
    sift_up((int)items.size() - 1);
    publish();
  }
  T getCost(int index) {
    auto it = pathPoints.find(items[index]);
    if (it != pathPoints.end()) return it->second.distanceFromStart;
    pathPoints[items[index]] = PathNode<T>();
    return std::numeric_limits<T>::max();
  }
  void sort() {
    for (int i = (int)items.size() - 1; i >= 0; --i) sift_down(i);
    publish();
  }
  void updateCost(int position, T cost) {
    const T before = getCost(position);
    pathPoints[items[position]].distanceFromStart = cost;
    if (before > cost) sift_up(position); else sift_down(position);
    publish();
  }
  void updateCost(R* value, T cost) { updateCost(pathPoints[value].heapPosition, cost); }
  R* pop() {
    if (items.empty()) return nullptr;
    R* top = items.front();
    items.front() = items.back();
    items.pop_back();
    if (!items.empty()) { note(0); sift_down(0); }
    publish();
    return top;
  }
  R* pop(int id) {
    const int n = (int)items.size();
    if (id < 0 || id >= n) return nullptr;
    R* out = items[id];
    if (id == n - 1) {
      items.pop_back();
    } else {
      const T removed = getCost(id), moved = getCost(n - 1);
      items[id] = items.back();
      items.pop_back();
      note(id);
      if (moved < removed) sift_up(id); else sift_down(id);
    }
    publish();
    return out;
  }
  R* get() { return items.front(); }
  R* get(int id) { return items[id]; }
  void replace(int id, R* v) {
    const T before = getCost(id);
    items[id] = v;
    note(id);
    if (getCost(id) < before) sift_up(id); else sift_down(id);
    publish();
  }
  int size() { return (int)items.size(); }
  void clear() { items.clear(); publish(); }
  const bool empty() { return items.empty(); }
  bool checkOrdering() {
    for (int i = 1; i < (int)items.size(); ++i)
      if (getCost(i) < getCost((i - 1) / 2)) return false;
    return true;
  }
  bool check() { return external == nullptr; }
  std::deque<R*>* getHeapVector() {
    mirror.assign(items.begin(), items.end());
    return &mirror;
  }

 private:
  std::vector<R*> items;
  std::deque<R*>* external{nullptr};
  std::deque<R*> mirror;
  CostFunction cost_fn{Distance};
  bool calc{true};

  void note(size_t i) {   // element moved to slot i: position and (when asked to) its cost are refreshed
    PathNode<T>& pn = pathPoints[items[i]];
    pn.heapPosition = (int)i;
    if (calc && cost_fn && refPoint) pn.distanceFromStart = cost_fn(*items[i], *refPoint);
  }
  void refresh_all() { for (size_t i = 0; i < items.size(); ++i) note(i); }
  void swap_slots(int a, int b) {
    R* t = items[a]; items[a] = items[b]; items[b] = t;
    note(a);
    note(b);
  }
  void sift_down(int i) {
    const int n = (int)items.size();
    while (true) {
      const int l = 2 * i + 1, r = l + 1;
      if (l >= n) return;
      int best = i;
      if (getCost(i) > getCost(l)) best = l;
      if (r < n && getCost(best) > getCost(r)) best = r;
      if (best == i) return;
      swap_slots(i, best);
      i = best;
    }
  }
  void sift_up(int i) {
    while (i > 0) {
      const int p = (i - 1) / 2;
      if (!(getCost(p) > getCost(i))) return;
      swap_slots(p, i);
      i = p;
    }
  }
  void publish() { if (external) external->assign(items.begin(), items.end()); }
};
