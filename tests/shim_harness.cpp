// shim_harness.cpp — drives include/sff/flann/flann.hpp and include/sff/RAPID.H (the two inner seams, SURVEY.md
// section 8(b)) through exactly the calls the reference's solvers make, on seeded data, and prints inputs and
// answers as JSON (hex floats).  tests/test_shims.py checks the answers against the CPU oracle.  This repository's
// own code; the reference is not involved.
//   usage: shim_harness <env.txt> <robot.txt>      (text files: one triangle per line, 9 numbers)
#include <cstdio>
#include <fstream>
#include <random>
#include <string>
#include <vector>

#include "primitives.h"   // Point, D6Distance (and <flann/flann.hpp>)
#include "RAPID.H"

static void pd(double v) { printf("\"%a\"", v); }

static std::vector<double> load(const char* path) {
  std::vector<double> v;
  std::ifstream f(path);
  double x;
  while (f >> x) v.push_back(x);
  return v;
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  std::mt19937_64 e(99);
  std::uniform_real_distribution<float> px(0, 120), pa(-3.1f, 3.1f);
  printf("{\n\"flann\":[\n");
  for (int cols = 6; cols >= 2; cols -= 4) {   // 6-D and 2-D indices (PROBLEM_DIMENSION columns)
    const int T = 3, NP = 300;
    std::vector<flann::Index<D6Distance<float>>*> index(T);
    std::vector<std::vector<float*>> rows(T);
    // every tree starts from one root row and grows point by point (src/forest.h:65-73, :367)
    for (int t = 0; t < T; ++t) {
      float* r = new float[cols];
      for (int k = 0; k < cols; ++k) r[k] = k < 3 ? px(e) : pa(e);
      rows[t].push_back(r);
      flann::Matrix<float> root(r, 1, cols);
      index[t] = new flann::Index<D6Distance<float>>(root, flann::KDTreeIndexParams(4));
      index[t]->buildIndex();
    }
    for (int i = 1; i < NP; ++i)
      for (int t = 0; t < T; ++t) {
        float* r = new float[cols];
        for (int k = 0; k < cols; ++k) r[k] = k < 3 ? px(e) : pa(e);
        if (i % 50 == 0) for (int k = 0; k < cols; ++k) r[k] = rows[t][i - 1][k];   // exact duplicates
        rows[t].push_back(r);
        flann::Matrix<float> m(r, 1, cols);
        index[t]->addPoints(m);
      }
    printf("%s{\"cols\":%d,\"trees\":[", cols == 6 ? "" : ",\n", cols);
    for (int t = 0; t < T; ++t) {
      printf("%s[", t ? "," : "");
      for (size_t i = 0; i < rows[t].size(); ++i)
        for (int k = 0; k < cols; ++k) { if (i || k) printf(","); pd(rows[t][i][k]); }
      printf("]");
    }
    printf("],\n\"queries\":[");
    for (int q = 0; q < 40; ++q) {
      float qv[6] = {0, 0, 0, 0, 0, 0};
      for (int k = 0; k < cols; ++k) qv[k] = k < 3 ? px(e) : pa(e);
      if (q % 8 == 0) for (int k = 0; k < cols; ++k) qv[k] = rows[q % T][q + 1][k];   // a stored point itself
      const int t = q % T;
      const float r2 = (q % 2 ? 30.0f : 55.0f) * (q % 2 ? 30.0f : 55.0f);
      flann::Matrix<float> qm(qv, 1, cols);
      std::vector<std::vector<int>> idx;
      std::vector<std::vector<float>> dd;
      const int nr = index[t]->radiusSearch(qm, idx, dd, r2, flann::SearchParams(128));
      printf("%s{\"tree\":%d,\"q\":[", q ? ",\n" : "", t);
      for (int k = 0; k < 6; ++k) { if (k) printf(","); pd(qv[k]); }
      printf("],\"r2\":");
      pd(r2);
      printf(",\"radius_n\":%d,\"radius_idx\":[", nr);
      for (size_t i = 0; i < idx[0].size(); ++i) printf("%s%d", i ? "," : "", idx[0][i]);
      printf("],\"radius_d\":[");
      for (size_t i = 0; i < dd[0].size(); ++i) { if (i) printf(","); pd(dd[0][i]); }
      printf("],");
      const size_t ks[2] = {1, 9};
      for (int w = 0; w < 2; ++w) {
        index[t]->knnSearch(qm, idx, dd, ks[w], flann::SearchParams(128));
        printf("\"knn%zu_idx\":[", ks[w]);
        for (size_t i = 0; i < idx[0].size(); ++i) printf("%s%d", i ? "," : "", idx[0][i]);
        printf("],\"knn%zu_d\":[", ks[w]);
        for (size_t i = 0; i < dd[0].size(); ++i) { if (i) printf(","); pd(dd[0][i]); }
        printf("]%s", w == 0 ? "," : "");
      }
      printf("}");
    }
    printf("]}");
    for (int t = 0; t < T; ++t) {
      delete index[t];
      for (float* r : rows[t]) delete[] r;
    }
  }
  printf("],\n");

  // ---- RAPID surface: models built triangle by triangle, one RAPID_Collide per pose in the reference's argument
  // order (obstacle at identity / zero first, posed robot second: src/environment.h:268-276)
  const std::vector<double> env = load(argv[1]), rob = load(argv[2]);
  RAPID_model *menv = new RAPID_model(), *mrob = new RAPID_model();
  menv->BeginModel();
  for (size_t t = 0; t < env.size() / 9; ++t) menv->AddTri(&env[9 * t], &env[9 * t + 3], &env[9 * t + 6], (int)t);
  menv->EndModel();
  mrob->BeginModel();
  for (size_t t = 0; t < rob.size() / 9; ++t) mrob->AddTri(&rob[9 * t], &rob[9 * t + 3], &rob[9 * t + 6], (int)t);
  mrob->EndModel();
  double eye[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, zero[3] = {0, 0, 0};
  std::uniform_real_distribution<double> ut(0, 1), ang(-3.14, 3.14);
  std::uniform_int_distribution<size_t> pick(0, env.size() / 9 - 1);
  printf("\"rapid\":[\n");
  for (int i = 0; i < 400; ++i) {
    // poses near the surface of the environment, so that both outcomes occur
    const size_t t = pick(e);
    double w0 = ut(e), w1 = ut(e) * (1 - w0), w2 = 1 - w0 - w1;
    Point<double> p(w0 * env[9 * t] + w1 * env[9 * t + 3] + w2 * env[9 * t + 6] + 8 * (ut(e) - 0.5),
                    w0 * env[9 * t + 1] + w1 * env[9 * t + 4] + w2 * env[9 * t + 7] + 8 * (ut(e) - 0.5),
                    w0 * env[9 * t + 2] + w1 * env[9 * t + 5] + w2 * env[9 * t + 8] + 8 * (ut(e) - 0.5), ang(e), ang(e), ang(e));
    double R[3][3], T[3] = {p.x(), p.y(), p.z()};
    p.FillRotationMatrix(R);
    RAPID_Collide(eye, zero, menv, R, T, mrob);
    const int forward = RAPID_num_contacts;
    RAPID_Collide(R, T, mrob, eye, zero, menv, RAPID_FIRST_CONTACT);   // the other argument order
    printf("%s{\"p\":[", i ? ",\n" : "");
    for (int k = 0; k < 6; ++k) { if (k) printf(","); pd(p[k]); }
    printf("],\"hit\":%d,\"hit_swapped\":%d}", forward != 0, RAPID_num_contacts != 0);
  }
  printf("]\n}\n");
  return 0;
}
