// types_harness.cpp — one driver, two builds: the node / tree API of the solvers exercised through its PUBLIC
// interface only (Node, Tree, Heap, DistanceHolder, SymmetricMatrix, Point(string, scale)).
//
//   build A (authoring container only, `make -C oracle ref`): against the REFERENCE'S OWN src/primitives.h +
//            src/heap.h where they lie under /root/reference -> oracle/_ref/ref_types_harness; its JSON output is
//            committed as tests/golden/ref_types.json (tests/golden/make_ref_golden.py).
//   build B (everywhere, tests/test_dropin_types.py): against this repository's drop-in headers include/sff/ ->
//            must print the same JSON byte for byte.
//
// Test infrastructure.  No reference source is copied: the same calls are made through whichever header set is on
// the include path.  Doubles are printed as hex floats, so equality is bit equality.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <deque>
#include <map>
#include <queue>
#include <random>
#include <regex>
#include <set>
#include <sstream>
#include <string>
#include <vector>

#include "primitives.h"
#include "heap.h"

typedef Node<double, Point<double>> N;
typedef Tree<double, N> TR;

static void pd(double v) { printf("\"%a\"", v); }

static void dump_pos(const N& n) {
  printf("[");
  for (int k = 0; k < 6; ++k) { if (k) printf(","); pd(n.Position[k]); }
  printf("]");
}

static void dump_heap(Heap<double, N>& h) {
  printf("[");
  for (int i = 0; i < h.size(); ++i) printf("%s%d", i ? "," : "", h.get(i)->GetId());
  printf("]");
}

int main() {
  std::mt19937_64 e(2024);
  std::uniform_real_distribution<double> ux(-50, 50), ua(-3.1, 3.1);
  printf("{\n");

  // ---- Node identity rules
  TR* t0 = new TR();
  TR* t1 = new TR();
  t0->flannIndex = nullptr;   // (the reference leaves the pointer uninitialised)
  t1->flannIndex = nullptr;
  N& r0 = t0->nodes.emplace_back(Point<double>(1, 2, 3, 0.1, 0.2, 0.3), t0, nullptr, 0.0, 0.0, 0u);
  N& r1 = t1->nodes.emplace_back(Point<double>(9, 8, 7, -0.1, -0.2, -0.3), t1, nullptr, 0.0, 0.0, 0u);
  N& c0 = t0->nodes.emplace_back(Point<double>(2, 2, 3, 0.1, 0.2, 0.3), t0, &r0, 1.0, 1.0, 5u);
  N& c1 = t0->nodes.emplace_back(Point<double>(3, 2, 3, 0.1, 0.2, 0.3), t0, &c0, 1.0, 2.0, 9u);
  printf("\"node\":{\"ids\":[%d,%d,%d,%d],\"tree_ids\":[%d,%d],\"is_root\":[%d,%d,%d,%d],\"num_nodes\":%d,\"age\":[%u,%u],"
         "\"expanded_root_is_root\":[%d,%d,%d],\"less\":[%d,%d],\"equal\":[%d,%d]},\n",
         r0.GetId(), r1.GetId(), c0.GetId(), c1.GetId(), t0->GetId(), t1->GetId(), (int)r0.IsRoot(), (int)r1.IsRoot(),
         (int)c0.IsRoot(), (int)c1.IsRoot(), c1.GetNumNodes(), c0.GetAge(), c1.GetAge(), (int)(c1.ExpandedRoot == t0),
         (int)(r1.ExpandedRoot == t1), (int)(c0.ExpandedRoot == c0.Root), (int)(r0 < c1), (int)(c1 < r0), (int)(c0 == c0),
         (int)(c0 == c1));

  // ---- Heap: Tree::AddFrontier (heap over the tree's nodes keyed by the distance to a goal node), then a
  // script of pop / pop-at-index / push
  TR* t2 = new TR();
  t2->flannIndex = nullptr;
  const int first_id = c1.GetNumNodes();
  for (int i = 0; i < 48; ++i)
    t2->nodes.emplace_back(Point<double>(ux(e), ux(e), ux(e), ua(e), ua(e), ua(e)), t2, nullptr, 0.0, (double)i, (unsigned)i);
  // ties: two nodes at the same place as an earlier one
  t2->nodes.emplace_back(t2->nodes[3].Position, t2, nullptr, 0.0, 1.0, 100u);
  t2->nodes.emplace_back(t2->nodes[3].Position, t2, nullptr, 0.0, 1.0, 101u);
  N goal(Point<double>(5, -5, 2, 0.5, 0.0, -0.5), t1, nullptr, 0.0, 0.0, 0u);
  t2->AddFrontier(&goal);
  Heap<double, N>& h = t2->frontiers[0];
  printf("\"heap\":{\"first_id\":%d,\"goal\":", first_id);
  dump_pos(goal);
  printf(",\"positions\":[");
  for (size_t i = 0; i < t2->nodes.size(); ++i) { if (i) printf(","); dump_pos(t2->nodes[i]); }
  printf("],\"initial\":");
  dump_heap(h);
  printf(",\"cost0\":");
  pd(h.getCost(0));
  printf(",\"ops\":[\n");
  std::deque<N> extra;
  for (int step = 0; step < 120; ++step) {
    const int kind = (int)(e() % 4);   // 0 pop, 1/2 pop(id), 3 push
    int arg = -1, ret = -1;
    bool pushed = false;
    if (kind == 0 && h.size() > 0) {
      ret = h.pop()->GetId();
    } else if ((kind == 1 || kind == 2) && h.size() > 0) {
      arg = (int)(e() % (unsigned)h.size());
      if (kind == 2) arg = h.size() - 1 - (int)(e() % 2 == 0 ? 0 : std::min(1, h.size() - 1));
      ret = h.pop(arg)->GetId();
    } else {
      extra.emplace_back(Point<double>(ux(e), ux(e), ux(e), ua(e), ua(e), ua(e)), t2, nullptr, 0.0, 1.0, 200u + step);
      h.push(&extra.back());
      ret = extra.back().GetId();
      pushed = true;
    }
    printf("%s{\"kind\":%d,\"arg\":%d,\"ret\":%d,\"empty\":%d,", step ? ",\n" : "", kind, arg, ret, (int)h.empty());
    if (pushed) { printf("\"pushed\":"); dump_pos(extra.back()); printf(","); }
    printf("\"heap\":");
    dump_heap(h);
    printf("}");
  }
  printf("],\"empty_frontiers\":%d},\n", (int)t2->EmptyFrontiers());

  // ---- DistanceHolder
  {
    std::deque<N*> plan{&c1, &c0, &r0};
    DistanceHolder<double, N> a(&c1, &r1), b(&r1, &c1), c(&c1, &r0, 7.5), d(&c1, &r0, 2.5, plan), f(&r0, &c1, 2.5, plan), none;
    printf("\"holder\":{\"a\":[%d,%d,", a.node1->GetId(), a.node2->GetId());
    pd(a.distance);
    printf("],\"b\":[%d,%d,", b.node1->GetId(), b.node2->GetId());
    pd(b.distance);
    printf("],\"c\":[%d,%d],\"d_plan\":[", c.node1->GetId(), c.node2->GetId());
    for (size_t i = 0; i < d.plan.size(); ++i) printf("%s%d", i ? "," : "", d.plan[i]->GetId());
    printf("],\"f_plan\":[");
    for (size_t i = 0; i < f.plan.size(); ++i) printf("%s%d", i ? "," : "", f.plan[i]->GetId());
    printf("],\"less\":[%d,%d],\"equal\":[%d,%d],\"exists\":[%d,%d],", (int)(d < c), (int)(c < d), (int)(a == b), (int)(a == c),
           (int)a.Exists(), (int)none.Exists());
    c0.DistanceToRoot = 4.25;
    DistanceHolder<double, N> u(&c0, &r1);
    c0.DistanceToRoot = 1.0;
    const double before = u.distance;
    u.UpdateDistance();
    printf("\"update\":[");
    pd(before);
    printf(",");
    pd(u.distance);
    printf("]},\n");
  }

  // ---- SymmetricMatrix: (i, j) and (j, i) are one cell, distinct pairs distinct cells
  {
    const int n = 6;
    SymmetricMatrix<DistanceHolder<double, N>> M(n);
    for (int i = 0; i < n; ++i)
      for (int j = i; j < n; ++j) M(i, j).distance = 100.0 * i + j;
    printf("\"symmetric\":{\"n\":%d,\"read_transposed\":[", n);
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) printf("%s%g", (i || j) ? "," : "", M(j, i).distance);
    M(4, 1) = DistanceHolder<double, N>(&r0, &r1);
    printf("],\"exists\":[%d,%d,%d]},\n", (int)M.Exists(1, 4), (int)M.Exists(4, 1), (int)M.Exists(1, 3));
  }

  // ---- Point(string, scale) (the XML parser's point format) and the metric / steer on those points
  {
    const char* texts[] = {"[1;2;3]", "[-1.5; 4; 3]", "[2.9;  0.3; 7]", "x[10;-20;30.25]y", "[0.001;-.5;5.]"};
    printf("\"points\":[");
    for (int i = 0; i < 5; ++i) {
      Point<double> p(std::string(texts[i]), i % 2 ? 10.0 : 1.0);
      printf("%s[", i ? "," : "");
      for (int k = 0; k < 6; ++k) { if (k) printf(","); pd(p[k]); }
      printf("]");
    }
    int threw = 0;
    try { Point<double> bad(std::string("1 2 3"), 1.0); } catch (const std::invalid_argument&) { threw = 1; }
    printf("],\"bad_format_throws\":%d,\n", threw);
    printf("\"metric\":[");
    for (int i = 0; i < 24; ++i) {
      Point<double> a(ux(e), ux(e), ux(e), 2.2 * ua(e), ua(e), ua(e)), b(ux(e), ux(e), ux(e), ua(e), 2.2 * ua(e), ua(e));
      Point<double> s = a.getStateInDistance(b, 3.0 + i);
      printf("%s[", i ? "," : "");
      pd(a.distance(b));
      printf(",");
      pd(b.distance(a));
      for (int k = 0; k < 6; ++k) { printf(","); pd(s[k]); }
      double R[3][3];
      a.FillRotationMatrix(R);
      for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { printf(","); pd(R[r][c]); }
      printf("]");
    }
    printf("]\n");
  }
  printf("}\n");
  return 0;
}
