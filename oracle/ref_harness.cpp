// ref_harness.cpp — drives the REFERENCE'S OWN headers (src/primitives.h, src/randGen.h and
// the vendored FLANN) to emit golden vectors for the parts of the hot path that compile
// without RAPID.  Test infrastructure; built only in the authoring container into
// oracle/_ref/ (git-ignored); its JSON output is committed as tests/golden/ref_primitives.json
// by tests/golden/make_ref_golden.py.  No reference source is copied: the headers are
// included from where they lie under /root/reference.
//
// The reference seeds RandGen from the clock with no override (src/randGen.h:52-55), so the
// private engine is re-seeded here; the access-specifier macro is applied only after every
// std / FLANN header has been parsed (SURVEY.md §8(c) "Seeding").
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
#include <flann/flann.hpp>

#define private public
#define protected public
#include "primitives.h"
#include "randGen.h"
#undef private
#undef protected

static void pd(double v) { printf("\"%a\"", v); }
static void ppoint(const Point<double>& p) {
  printf("[");
  for (int i = 0; i < 6; ++i) { if (i) printf(","); pd(p[i]); }
  printf("]");
}

int main() {
  Range<double> lim{-60, 2060, -60, 2110, 0, 1000};
  printf("{\n\"limits\":[-60,2060,-60,2110,0,1000],\n\"rng\":[\n");
  const uint64_t seeds[3] = {1ULL, 42ULL, 123456789012345ULL};
  for (int s = 0; s < 3; ++s) {
    printf("{\"seed\":\"%llu\",\n", (unsigned long long)seeds[s]);
    {
      RandGen<double> g(lim);
      g.rndEng = randomEngine(seeds[s]);
      const int his[6] = {1, 2, 9, 100, 12345, 2147483646};
      printf("\"ints\":[");
      for (int h = 0; h < 6; ++h) {
        printf("%s{\"hi\":%d,\"v\":[", h ? "," : "", his[h]);
        for (int i = 0; i < 48; ++i) printf("%s%d", i ? "," : "", g.randomIntMinMax(0, his[h]));
        printf("]}");
      }
      printf("],\n\"probs\":[");
      for (int i = 0; i < 48; ++i) { if (i) printf(","); pd(g.randomProbability()); }
      printf("],\n");
    }
    {
      RandGen<double> g(lim);
      g.rndEng = randomEngine(seeds[s]);
      Point<double> c(100, 200, 300, 0.1, -0.2, 3.0), out;
      printf("\"pid3\":[");
      for (int i = 0; i < 48; ++i) {
        bool ok = g.randomPointInDistance(c, out, i % 2 ? 14.0 : 4000.0, D3);
        printf("%s{\"ok\":%d,\"p\":", i ? "," : "", ok ? 1 : 0);
        ppoint(out);
        printf("}");
        if (i % 3 == 0) c = out;
      }
      printf("],\n");
    }
    {
      RandGen<double> g(lim);
      g.rndEng = randomEngine(seeds[s]);
      Point<double> c(100, 200, 0, 0, 0, 0), out;
      printf("\"pid2\":[");
      for (int i = 0; i < 48; ++i) {
        bool ok = g.randomPointInDistance(c, out, 80.0, D2);
        printf("%s{\"ok\":%d,\"p\":", i ? "," : "", ok ? 1 : 0);
        ppoint(out);
        printf("}");
        c = out;
      }
      printf("],\n");
    }
    {
      RandGen<double> g(lim);
      g.rndEng = randomEngine(seeds[s]);
      Point<double> out;
      printf("\"pis3\":[");
      for (int i = 0; i < 32; ++i) { g.randomPointInSpace(out, D3); if (i) printf(","); ppoint(out); }
      printf("],\n\"pis2\":[");
      for (int i = 0; i < 32; ++i) { g.randomPointInSpace(out, D2); if (i) printf(","); ppoint(out); }
      printf("]\n");
    }
    printf("}%s\n", s < 2 ? "," : "");
  }
  printf("],\n");

  // metric / steer / rotation on seeded random pairs (angles partly outside [-pi, pi))
  std::mt19937_64 e(7);
  std::uniform_real_distribution<double> ux(-2000, 2000), ua(-7, 7), ud(0.1, 300);
  printf("\"metric\":[\n");
  for (int i = 0; i < 200; ++i) {
    Point<double> a(ux(e), ux(e), ux(e), ua(e), ua(e), ua(e)), b(ux(e), ux(e), ux(e), ua(e), ua(e), ua(e));
    if (i % 4 == 0) b = Point<double>(a.x() + ua(e), a.y() + ua(e), a.z() + ua(e), a.Yaw + 0.1 * ua(e), a.Pitch, a.Roll);
    double d = ud(e);
    Point<double> st = a.getStateInDistance(b, d);
    double R[3][3];
    a.FillRotationMatrix(R);
    printf("%s{\"a\":", i ? ",\n" : "");
    ppoint(a);
    printf(",\"b\":");
    ppoint(b);
    printf(",\"dist\":");
    pd(a.distance(b));
    printf(",\"d\":");
    pd(d);
    printf(",\"steer\":");
    ppoint(st);
    printf(",\"R\":[");
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { if (r || c) printf(","); pd(R[r][c]); }
    printf("]}");
  }
  printf("],\n");

  // FLANN functor as shipped (src/primitives.h:405-438): operator() overwrites instead of
  // accumulating, so it returns only the wrapped roll difference squared.
  printf("\"d6\":[\n");
  D6Distance<float> fun;
  std::uniform_real_distribution<float> fx(-100, 100), fa(-3.1f, 3.1f);
  for (int i = 0; i < 32; ++i) {
    float a[6] = {fx(e), fx(e), fx(e), fa(e), fa(e), fa(e)}, b[6] = {fx(e), fx(e), fx(e), fa(e), fa(e), fa(e)};
    float v = fun(a, b, 6);
    printf("%s{\"a\":[", i ? ",\n" : "");
    for (int k = 0; k < 6; ++k) { if (k) printf(","); pd(a[k]); }
    printf("],\"b\":[");
    for (int k = 0; k < 6; ++k) { if (k) printf(","); pd(b[k]); }
    printf("],\"functor\":");
    pd(v);
    printf(",\"accum\":[");
    for (int k = 0; k < 6; ++k) { if (k) printf(","); pd(fun.accum_dist(a[k], b[k], k)); }
    printf("]}");
  }
  printf("],\n");

  // FLANN KDTreeIndex(4) + SearchParams(128) exactly as the solvers call it (src/forest.h:72,266-267,317):
  // recorded to document how far the shipped approximate/buggy search is from the exact neighbours.
  const int N = 1000;
  std::vector<float> data(N * 6);
  std::uniform_real_distribution<float> px(0, 400);
  for (int i = 0; i < N; ++i) {
    for (int k = 0; k < 3; ++k) data[6 * i + k] = px(e);
    for (int k = 3; k < 6; ++k) data[6 * i + k] = fa(e);
  }
  flann::Matrix<float> first(new float[6], 1, 6);
  for (int k = 0; k < 6; ++k) first[0][k] = data[k];
  flann::Index<D6Distance<float>> index(first, flann::KDTreeIndexParams(4));
  index.buildIndex();
  for (int i = 1; i < N; ++i) {
    flann::Matrix<float> m(&data[6 * i], 1, 6);
    index.addPoints(m);
  }
  printf("\"flann\":{\"n\":%d,\"points\":[", N);
  for (int i = 0; i < N * 6; ++i) { if (i) printf(","); pd(data[i]); }
  printf("],\n\"queries\":[\n");
  for (int q = 0; q < 16; ++q) {
    float qv[6] = {px(e), px(e), px(e), fa(e), fa(e), fa(e)};
    flann::Matrix<float> qm(qv, 1, 6);
    std::vector<std::vector<int>> idx;
    std::vector<std::vector<float>> dd;
    int nr = index.radiusSearch(qm, idx, dd, 60.0f * 60.0f, flann::SearchParams(128));
    printf("%s{\"q\":[", q ? ",\n" : "");
    for (int k = 0; k < 6; ++k) { if (k) printf(","); pd(qv[k]); }
    printf("],\"radius_n\":%d,\"radius_idx\":[", nr);
    for (int i = 0; i < nr; ++i) printf("%s%d", i ? "," : "", idx[0][i]);
    printf("],");
    idx.clear(); dd.clear();
    index.knnSearch(qm, idx, dd, 8, flann::SearchParams(128));
    printf("\"knn_idx\":[");
    for (size_t i = 0; i < idx[0].size(); ++i) printf("%s%d", i ? "," : "", idx[0][i]);
    printf("]}");
  }
  printf("]}\n}\n");
  delete[] first.ptr();
  return 0;
}
