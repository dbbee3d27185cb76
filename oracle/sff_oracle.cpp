// sff_oracle.cpp — CPU ORACLE: test infrastructure only, never part of the shipped path.
// Plain single-threaded restatement of the reference hot path; every block cites the
// reference file:line it follows (paths relative to /root/reference).  See sff_oracle.h
// for the parity status ("parity unpinned" at the RAPID boundary and for the solver loop).
//
// Build: g++ -std=c++17 -O2 -ffp-contract=off -fPIC -shared (oracle/Makefile).
#include "sff_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <limits>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

// portable trig: the DEFINITION shared with the device kernels (see that header)
#include "../space_filling_forest_star_amd/csrc/sff_pmath.h"

namespace {

constexpr double TOLERANCE = 1e-9;  // src/primitives.h:45

// ------------------------------------------------------------------ trig dispatch
inline double tsin(double x, int trig) { return trig == SFFO_TRIG_LIBM ? std::sin(x) : sffp::psin(x); }
inline double tcos(double x, int trig) { return trig == SFFO_TRIG_LIBM ? std::cos(x) : sffp::pcos(x); }
inline double tacos(double x, int trig) { return trig == SFFO_TRIG_LIBM ? std::acos(x) : sffp::pacos(x); }

// ------------------------------------------------------------------ primitives
// src/primitives.h:278-286
inline double normalize_angle(double a) {
  if (a < -M_PI) return a + 2 * M_PI;
  if (a >= M_PI) return a - 2 * M_PI;
  return a;
}
// src/primitives.h:288-292  AngleDifference(a1, a2) = Normalize(a2 - a1)
inline double angle_diff(double a1, double a2) { return normalize_angle(a2 - a1); }

// src/primitives.h:224-235
inline double distance6(const double* a, const double* b) {
  double sum = 0;
  for (int i = 0; i < 3; ++i) {
    double d = a[i] - b[i];
    sum += d * d;
  }
  for (int i = 3; i < 6; ++i) {
    double d = angle_diff(a[i], b[i]);
    sum += d * d;
  }
  return std::sqrt(sum);
}

// src/primitives.h:237-250
inline void steer6(const double* from, const double* to, double dist, double* out) {
  double real = distance6(from, to);
  double dir[3], adir[3];
  for (int i = 0; i < 3; ++i) dir[i] = to[i] - from[i];
  for (int i = 0; i < 3; ++i) adir[i] = angle_diff(from[i + 3], to[i + 3]);
  for (int i = 0; i < 3; ++i) out[i] = from[i] + dir[i] * (dist / real);
  for (int i = 0; i < 3; ++i) out[i + 3] = from[i + 3] + adir[i] * (dist / real);
}

// src/primitives.h:252-262  R = Rz(yaw) Ry(pitch) Rx(roll)
inline void rotation(const double* p, int trig, double R[9]) {
  double cy = tcos(p[3], trig), sy = tsin(p[3], trig);
  double cp = tcos(p[4], trig), sp = tsin(p[4], trig);
  double cr = tcos(p[5], trig), sr = tsin(p[5], trig);
  R[0] = cy * cp;
  R[1] = cy * sp * sr - sy * cr;
  R[2] = cy * sp * cr + sy * sr;
  R[3] = sy * cp;
  R[4] = sy * sp * sr + cy * cr;
  R[5] = sy * sp * cr - cy * sr;
  R[6] = -sp;
  R[7] = cp * sr;
  R[8] = cp * cr;
}

// ------------------------------------------------------------------ RNG
// std::mt19937_64 is fully specified by ISO C++ [rand.predef]; restated so that the
// stream does not depend on the GPU box's STL.  Seeding: src/randGen.h:53-55.
struct Mt64 {
  uint64_t mt[312];
  int idx;
  uint64_t n_drawn = 0;   // outputs produced since the last reseed
  explicit Mt64(uint64_t seed = 5489ULL) { reseed(seed); }
  void reseed(uint64_t seed) {
    mt[0] = seed;
    for (int i = 1; i < 312; ++i) mt[i] = 6364136223846793005ULL * (mt[i - 1] ^ (mt[i - 1] >> 62)) + (uint64_t)i;
    idx = 312;
    n_drawn = 0;
  }
  uint64_t next() {
    ++n_drawn;
    if (idx >= 312) {
      const uint64_t UM = 0xFFFFFFFF80000000ULL, LM = 0x7FFFFFFFULL;
      for (int i = 0; i < 312; ++i) {
        uint64_t x = (mt[i] & UM) | (mt[(i + 1) % 312] & LM);
        uint64_t xa = x >> 1;
        if (x & 1ULL) xa ^= 0xB5026F5AA96619E9ULL;
        mt[i] = mt[(i + 156) % 312] ^ xa;
      }
      idx = 0;
    }
    uint64_t y = mt[idx++];
    y ^= (y >> 29) & 0x5555555555555555ULL;
    y ^= (y << 17) & 0x71D67FFFEDA60000ULL;
    y ^= (y << 37) & 0xFFF7EEE000000000ULL;
    y ^= (y >> 43);
    return y;
  }
};

// libstdc++ generate_canonical<double,53> with a 64-bit engine: one draw, double(x)/2^64,
// clamped below 1 (/usr/include/c++/11/bits/random.tcc:3348-3380)
inline double canonical_from_word(uint64_t w) {
  double r = (double)w / 18446744073709551616.0;
  if (r >= 1.0) r = std::nextafter(1.0, 0.0);
  return r;
}
// uniform_real_distribution<double>(a,b): canonical*(b-a)+a
inline double uniform_real_from_word(uint64_t w, double a, double b) { return canonical_from_word(w) * (b - a) + a; }

struct Rng {
  Mt64 eng;
  double lim[6];
  int trig;
  uint64_t raw() { return eng.next(); }
  // src/randGen.h:149-152 + libstdc++ uniform_int_distribution (Lemire, bits/uniform_int_dist.h:243-310)
  int rand_int(int lo, int hi) {
    uint64_t range = (uint64_t)((int64_t)hi - (int64_t)lo) + 1ULL;
    unsigned __int128 prod = (unsigned __int128)eng.next() * range;
    uint64_t low = (uint64_t)prod;
    if (low < range) {
      uint64_t thr = (0ULL - range) % range;
      while (low < thr) {
        prod = (unsigned __int128)eng.next() * range;
        low = (uint64_t)prod;
      }
    }
    return lo + (int)(uint64_t)(prod >> 64);
  }
  double prob() { return uniform_real_from_word(eng.next(), 0.0, 1.0); }  // src/randGen.h:155-157
};

// src/randGen.h:160-170
inline bool in_limits(const double* p, const double* lim) {
  bool v = true;
  v &= p[0] >= lim[0];
  v &= p[0] <= lim[1];
  v &= p[1] >= lim[2];
  v &= p[1] <= lim[3];
  v &= p[2] >= lim[4];
  v &= p[2] <= lim[5];
  return v;
}

// src/randGen.h:70-109, fed with raw engine words in the reference's draw order
// (phi, theta, yaw, pitch-u, flip-u, roll).
inline bool sample_from_words(const uint64_t* w, const double* center, double dist, int dim, const double* lim,
                              int trig, double* out) {
  double temp[6];
  double phi = uniform_real_from_word(w[0], -M_PI, M_PI);
  if (dim == 2) {
    temp[0] = center[0] + tcos(phi, trig) * dist;
    temp[1] = center[1] + tsin(phi, trig) * dist;
    temp[2] = temp[3] = temp[4] = temp[5] = 0;
    for (int i = 0; i < 6; ++i) out[i] = temp[i];
  } else {
    double theta = uniform_real_from_word(w[1], -M_PI, M_PI);
    temp[0] = center[0] + tcos(theta, trig) * tsin(phi, trig) * dist;
    temp[1] = center[1] + tsin(theta, trig) * tsin(phi, trig) * dist;
    temp[2] = center[2] + tcos(phi, trig) * dist;
    temp[3] = uniform_real_from_word(w[2], -M_PI, M_PI);
    phi = tacos(1 - 2 * uniform_real_from_word(w[3], 0.0, 1.0), trig) + M_PI_2;
    if (uniform_real_from_word(w[4], 0.0, 1.0) < 0.5) {
      if (phi < 0) phi += M_PI; else phi -= M_PI;
    }
    temp[4] = phi;
    temp[5] = uniform_real_from_word(w[5], -M_PI, M_PI);
    steer6(center, temp, dist, out);
  }
  return in_limits(out, lim);
}

// ------------------------------------------------------------------ triangle contact
// RAPID 2.01 (UNC GAMMA; NOT in the reference tree) decides leaf contacts with a 17-axis
// separating-axis test: the two face normals, the 9 edge x edge directions and the 6
// in-plane edge normals; an axis separates only on a STRICT gap, so touching counts as
// contact.  Restated from the published algorithm.  This build's
// definition adds one exact pre-condition — the two triangles' axis-aligned boxes overlap
// (closed intervals) — which true contacts always satisfy and which makes hierarchical
// culling with exact boxes provably result-preserving.
inline void cross3(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
inline double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline bool project6(const double* ax, const double* p1, const double* p2, const double* p3, const double* q1,
                     const double* q2, const double* q3) {
  double P1 = dot3(ax, p1), P2 = dot3(ax, p2), P3 = dot3(ax, p3);
  double Q1 = dot3(ax, q1), Q2 = dot3(ax, q2), Q3 = dot3(ax, q3);
  double mx1 = P1 > P2 ? P1 : P2; if (P3 > mx1) mx1 = P3;
  double mn1 = P1 < P2 ? P1 : P2; if (P3 < mn1) mn1 = P3;
  double mx2 = Q1 > Q2 ? Q1 : Q2; if (Q3 > mx2) mx2 = Q3;
  double mn2 = Q1 < Q2 ? Q1 : Q2; if (Q3 < mn2) mn2 = Q3;
  if (mn1 > mx2) return false;
  if (mn2 > mx1) return false;
  return true;
}
inline bool aabb_overlap_tri(const double* P, const double* Q) {
  for (int a = 0; a < 3; ++a) {
    double pmin = std::min(P[a], std::min(P[3 + a], P[6 + a])), pmax = std::max(P[a], std::max(P[3 + a], P[6 + a]));
    double qmin = std::min(Q[a], std::min(Q[3 + a], Q[6 + a])), qmax = std::max(Q[a], std::max(Q[3 + a], Q[6 + a]));
    if (pmin > qmax || qmin > pmax) return false;
  }
  return true;
}
bool sat17(const double* P, const double* Q) {
  double p1[3], p2[3], p3[3], q1[3], q2[3], q3[3];
  for (int i = 0; i < 3; ++i) {
    p1[i] = P[i] - P[i];
    p2[i] = P[3 + i] - P[i];
    p3[i] = P[6 + i] - P[i];
    q1[i] = Q[i] - P[i];
    q2[i] = Q[3 + i] - P[i];
    q3[i] = Q[6 + i] - P[i];
  }
  double e1[3], e2[3], e3[3], f1[3], f2[3], f3[3];
  for (int i = 0; i < 3; ++i) {
    e1[i] = p2[i] - p1[i];
    e2[i] = p3[i] - p2[i];
    e3[i] = p1[i] - p3[i];
    f1[i] = q2[i] - q1[i];
    f2[i] = q3[i] - q2[i];
    f3[i] = q1[i] - q3[i];
  }
  double n1[3], m1[3], ax[3];
  cross3(e1, e2, n1);
  cross3(f1, f2, m1);
  if (!project6(n1, p1, p2, p3, q1, q2, q3)) return false;
  if (!project6(m1, p1, p2, p3, q1, q2, q3)) return false;
  const double* E[3] = {e1, e2, e3};
  const double* F[3] = {f1, f2, f3};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      cross3(E[i], F[j], ax);
      if (!project6(ax, p1, p2, p3, q1, q2, q3)) return false;
    }
  for (int i = 0; i < 3; ++i) {
    cross3(E[i], n1, ax);
    if (!project6(ax, p1, p2, p3, q1, q2, q3)) return false;
  }
  for (int i = 0; i < 3; ++i) {
    cross3(F[i], m1, ax);
    if (!project6(ax, p1, p2, p3, q1, q2, q3)) return false;
  }
  return true;
}
inline bool tri_contact(const double* P, const double* Q) { return aabb_overlap_tri(P, Q) && sat17(P, Q); }

// world = R * v + T, evaluated ((R0*v0 + R1*v1) + R2*v2) + T
inline void xform(const double* R, const double* T, const double* v, double* w) {
  for (int i = 0; i < 3; ++i) w[i] = ((R[3 * i] * v[0] + R[3 * i + 1] * v[1]) + R[3 * i + 2] * v[2]) + T[i];
}

// ------------------------------------------------------------------ collision world
struct Box {
  double lo[3], hi[3];
};
inline bool box_overlap(const Box& a, const Box& b) {
  for (int i = 0; i < 3; ++i)
    if (a.lo[i] > b.hi[i] || b.lo[i] > a.hi[i]) return false;
  return true;
}
struct BvhNode {
  Box box;
  int left, right;  // children, or left = -1 - firstTri, right = count for leaves
};

struct World {
  std::vector<double> env;    // n_env * 9, world frame (obstacles are posed at identity: src/environment.h:274)
  std::vector<double> robot;  // n_robot * 9, model frame
  std::vector<Box> env_box;
  std::vector<int> order;     // bvh leaf order -> env tri index
  std::vector<BvhNode> nodes;
  int trig;
  uint64_t collide_calls = 0;
  bool has_map;

  int build(int begin, int end) {
    BvhNode nd;
    for (int i = 0; i < 3; ++i) { nd.box.lo[i] = 1e300; nd.box.hi[i] = -1e300; }
    for (int k = begin; k < end; ++k) {
      const Box& b = env_box[order[k]];
      for (int i = 0; i < 3; ++i) {
        nd.box.lo[i] = std::min(nd.box.lo[i], b.lo[i]);
        nd.box.hi[i] = std::max(nd.box.hi[i], b.hi[i]);
      }
    }
    int id = (int)nodes.size();
    nodes.push_back(nd);
    if (end - begin <= 4) {
      nodes[id].left = -1 - begin;
      nodes[id].right = end - begin;
      return id;
    }
    int ax = 0;
    double ext = -1;
    for (int i = 0; i < 3; ++i)
      if (nd.box.hi[i] - nd.box.lo[i] > ext) { ext = nd.box.hi[i] - nd.box.lo[i]; ax = i; }
    int mid = (begin + end) / 2;
    std::nth_element(order.begin() + begin, order.begin() + mid, order.begin() + end, [&](int a, int b) {
      return env_box[a].lo[ax] + env_box[a].hi[ax] < env_box[b].lo[ax] + env_box[b].hi[ax];
    });
    int l = build(begin, mid);
    int r = build(mid, end);
    nodes[id].left = l;
    nodes[id].right = r;
    return id;
  }

  void init() {
    int n = (int)env.size() / 9;
    has_map = n > 0;
    env_box.resize(n);
    order.resize(n);
    for (int t = 0; t < n; ++t) {
      order[t] = t;
      for (int a = 0; a < 3; ++a) {
        const double* P = &env[9 * t];
        env_box[t].lo[a] = std::min(P[a], std::min(P[3 + a], P[6 + a]));
        env_box[t].hi[a] = std::max(P[a], std::max(P[3 + a], P[6 + a]));
      }
    }
    if (n) build(0, n);
    int nv = (int)robot.size() / 3;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int v = 0; v < nv; ++v)
      for (int i = 0; i < 3; ++i) { lo[i] = std::min(lo[i], robot[3 * v + i]); hi[i] = std::max(hi[i], robot[3 * v + i]); }
    for (int i = 0; i < 3; ++i) rc[i] = nv ? 0.5 * (lo[i] + hi[i]) : 0;
    rrad = 0;
    for (int v = 0; v < nv; ++v) {
      double d2 = 0;
      for (int i = 0; i < 3; ++i) d2 += (robot[3 * v + i] - rc[i]) * (robot[3 * v + i] - rc[i]);
      rrad = std::max(rrad, std::sqrt(d2));
    }
  }

  // src/environment.h:268-276 + :306-316 — boolean "robot at pose p touches any obstacle triangle".
  // Brute force over all pairs: the ground truth the accelerated paths must equal.
  bool collide_brute(const double* p) {
    double R[9];
    rotation(p, trig, R);
    int nr = (int)robot.size() / 9, ne = (int)env.size() / 9;
    bool hit = false;
    for (int r = 0; r < nr; ++r) {
      double Q[9];
      for (int v = 0; v < 3; ++v) xform(R, p, &robot[9 * r + 3 * v], &Q[3 * v]);
      for (int e = 0; e < ne; ++e)
        if (tri_contact(&env[9 * e], Q)) hit = true;
    }
    return hit;
  }

  // Same boolean through an exact-box hierarchy (the oracle's own accelerator; it culls
  // only pairs whose boxes do not overlap, which tri_contact rejects anyway).
  bool collide(const double* p) {
    ++collide_calls;
    if (!has_map) return false;  // src/environment.h:307-309
    double R[9];
    const bool ident = p[3] == 0 && p[4] == 0 && p[5] == 0;
    if (ident) {
      // cos(0) = 1 and sin(0) = 0 exactly in both trig modes, so FillRotationMatrix yields the
      // identity (up to the sign of zeros, which no product below can observe)
      R[0] = R[4] = R[8] = 1; R[1] = R[2] = R[3] = R[5] = R[6] = R[7] = 0;
    } else {
      rotation(p, trig, R);
    }
    // conservative pre-test: a box around the robot's bounding sphere (slightly inflated for
    // rounding) that misses every leaf box proves that no triangle boxes overlap
    {
      double c[3];
      xform(R, p, rc, c);
      double rr = rrad * (1 + 1e-9) + 1e-9 * (std::fabs(c[0]) + std::fabs(c[1]) + std::fabs(c[2]) + 1);
      Box sb;
      for (int i = 0; i < 3; ++i) { sb.lo[i] = c[i] - rr; sb.hi[i] = c[i] + rr; }
      if (!any_leaf_overlap(sb)) return false;
    }
    int nr = (int)robot.size() / 9;
    std::vector<double>& W = scratch;
    W.resize(robot.size());
    Box rb;
    for (int i = 0; i < 3; ++i) { rb.lo[i] = 1e300; rb.hi[i] = -1e300; }
    for (int v = 0; v < nr * 3; ++v) {
      xform(R, p, &robot[3 * v], &W[3 * v]);
      for (int i = 0; i < 3; ++i) {
        rb.lo[i] = std::min(rb.lo[i], W[3 * v + i]);
        rb.hi[i] = std::max(rb.hi[i], W[3 * v + i]);
      }
    }
    int stack[128], sp = 0;
    stack[sp++] = 0;
    while (sp) {
      const BvhNode& nd = nodes[stack[--sp]];
      if (!box_overlap(nd.box, rb)) continue;
      if (nd.left < 0) {
        int first = -1 - nd.left;
        for (int k = first; k < first + nd.right; ++k) {
          int e = order[k];
          if (!box_overlap(env_box[e], rb)) continue;
          for (int r = 0; r < nr; ++r)
            if (tri_contact(&env[9 * e], &W[9 * r])) return true;
        }
      } else {
        stack[sp++] = nd.left;
        stack[sp++] = nd.right;
      }
    }
    return false;
  }
  bool any_leaf_overlap(const Box& b) const {
    int stack[128], sp = 0;
    stack[sp++] = 0;
    while (sp) {
      const BvhNode& nd = nodes[stack[--sp]];
      if (!box_overlap(nd.box, b)) continue;
      if (nd.left < 0) {
        int first = -1 - nd.left;
        for (int k = first; k < first + nd.right; ++k)
          if (box_overlap(env_box[order[k]], b)) return true;
      } else {
        stack[sp++] = nd.left;
        stack[sp++] = nd.right;
      }
    }
    return false;
  }
  double rc[3] = {0, 0, 0}, rrad = 0;  // robot bounding sphere (model frame)
  std::vector<double> scratch;

  // src/problemStruct.h:154-168 — samples index = 1 .. < parts, rotation fixed at zero,
  // endpoints excluded, early exit on the first hit.
  bool path_free(const double* a, const double* b, int* first_hit, int* n_samples) {
    double total = distance6(a, b);
    double parts = total / 0.1;  // collisionSampleSize{0.1}, src/problemStruct.h:121
    double dir[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
    double pos[6] = {0, 0, 0, 0, 0, 0};
    bool is_free = true;
    int fh = -1, ns = 0;
    if (parts > 1) ns = (int)std::ceil(parts) - 1;  // number of integers index >= 1 with index < parts
    for (unsigned int index = 1; index < parts && is_free; ++index) {
      for (int i = 0; i < 3; ++i) pos[i] = a[i] + index * dir[i] / parts;
      if (collide(pos)) { is_free = false; fh = (int)index; }
    }
    if (first_hit) *first_hit = fh;
    if (n_samples) *n_samples = ns;
    return is_free;
  }
};

// ------------------------------------------------------------------ mesh parsing
// src/primitives.h:680-697 (single-delimiter split; the multi-char look-ahead quirk at
// :685-687 only triggers when the delimiter is the last character of the line)
bool split_token(std::string& line, std::string& tok, const std::string& delim) {
  size_t pos = line.find(delim);
  if (pos != std::string::npos) {
    tok = line.substr(0, pos);
    line = line.substr(pos + delim.size());
    return true;
  }
  tok = line;
  line.clear();
  return false;
}
std::string trim_ws(const std::string& s) {
  const char* ws = " \n\r\t\f\v";
  size_t b = s.find_first_not_of(ws);
  if (b == std::string::npos) return "";
  size_t e = s.find_last_not_of(ws);
  return s.substr(b, e - b + 1);
}

// src/environment.h:125-166 (ParseOBJFile), :197-223 (addPoint/addFacet)
int parse_obj(const char* path, const double* pos, double scale, std::vector<double>& tris) {
  std::ifstream f(path);
  if (!f) return -1;
  std::vector<double> pts;
  std::string line, tok;
  while (std::getline(f, line)) {
    split_token(line, tok, " ");
    char c = tok.empty() ? '\0' : tok[0];
    if (c == 'v') {  // NB: "vn"/"vt" lines also land here (src/environment.h:134-144)
      double p[3];
      for (int i = 0; i < 3; ++i) {
        split_token(line, tok, " ");
        p[i] = std::stod(tok) + pos[i];   // position added BEFORE scaling (:140, :198-202)
      }
      for (int i = 0; i < 3; ++i) pts.push_back(p[i] * scale);
    } else if (c == 'f') {
      for (int i = 0; i < 3; ++i) {
        split_token(line, tok, " ");
        int k = std::stoi(tok);            // "1//1" -> 1 (:145-153)
        int at = k - 0 - 1;                // offset never advances (:155-159)
        if (at < 0 || (size_t)at * 3 + 2 >= pts.size()) return -1;
        for (int j = 0; j < 3; ++j) tris.push_back(pts[3 * at + j]);
      }
    }
  }
  return (int)tris.size() / 9;
}

// src/environment.h:169-195 (ParseMapFile): rows of 3 x (x y), z = 0
int parse_tri2d(const char* path, const double* pos, double scale, std::vector<double>& tris) {
  std::ifstream f(path);
  if (!f) return -1;
  std::string line, tok;
  while (std::getline(f, line)) {
    line = trim_ws(line);
    if (line.empty()) continue;
    for (int i = 0; i < 3; ++i) {
      double p[3] = {0, 0, 0};
      for (int j = 0; j < 2; ++j) {
        split_token(line, tok, " ");
        p[j] = std::stod(tok) + pos[j];
      }
      // pointCache[2] stays 0 and is scaled in place by addPoint (src/environment.h:173,198-202)
      for (int j = 0; j < 3; ++j) tris.push_back(p[j] * scale);
    }
  }
  return (int)tris.size() / 9;
}

// ------------------------------------------------------------------ exact neighbours
struct Hit {
  double d;
  int idx;
  bool operator<(const Hit& o) const { return d < o.d || (d == o.d && idx < o.idx); }
};

// ------------------------------------------------------------------ SFF forest
struct FNode {
  double pos[6];
  int tree;
  int parent;  // global id, -1 for roots
  int idx_in_tree;
  bool force_children = false;
  double d_closest, d_root;
  unsigned iter;
};
struct Border {
  int n1, n2;
  double dist;
};

// uniform hash grid over xyz for exact radius queries (cell = query radius)
struct Grid {
  double cell = 1;
  std::unordered_map<uint64_t, std::vector<int>> cells;
  static uint64_t key(int64_t x, int64_t y, int64_t z) {
    return ((uint64_t)(x & 0x1FFFFF) << 42) | ((uint64_t)(y & 0x1FFFFF) << 21) | (uint64_t)(z & 0x1FFFFF);
  }
  void coords(const double* p, int64_t* c) const {
    for (int i = 0; i < 3; ++i) c[i] = (int64_t)std::floor(p[i] / cell);
  }
  void insert(const double* p, int id) {
    int64_t c[3];
    coords(p, c);
    cells[key(c[0], c[1], c[2])].push_back(id);
  }
};


// Priority frontier (src/heap.h): binary min-heap of node ids keyed by Distance(node, refPoint)
// (src/primitives.h:726-729), with the reference's BubbleUp / BubbleDown and its pop-at-index.
struct PHeap {
  std::vector<int> v;
  const std::vector<FNode>* nodes = nullptr;
  double ref[6];
  double cost(int i) const { return distance6((*nodes)[v[i]].pos, ref); }   // HEAPCOST_GET
  void bubble_down(int index) {                                              // src/heap.h:122-153
    const int size = (int)v.size();
    const int l = 2 * index + 1, r = 2 * index + 2;
    if (l >= size) return;
    int mi = index;
    if (cost(index) > cost(l)) mi = l;
    if (r < size && cost(mi) > cost(r)) mi = r;
    if (mi != index) { std::swap(v[index], v[mi]); bubble_down(mi); }
  }
  void bubble_up(int index) {                                                // src/heap.h:155-173
    if (index == 0) return;
    const int p = (index - 1) / 2;
    if (cost(p) > cost(index)) { std::swap(v[p], v[index]); bubble_up(p); }
  }
  void push(int n) { v.push_back(n); bubble_up((int)v.size() - 1); }         // :175-187
  int pop() {                                                                // :189-207
    int mn = v[0];
    v[0] = v.back();
    v.pop_back();
    bubble_down(0);
    return mn;
  }
  int pop_at(int id) {                                                       // :209-238
    const int size = (int)v.size();
    const double old_cost = cost(id);
    int val = -1;
    if (id == size - 1) { val = v.back(); v.pop_back(); }
    else if (size > id) {
      const double new_cost = cost(size - 1);
      val = v[id];
      v[id] = v[size - 1];
      v.pop_back();
      if (new_cost < old_cost) bubble_up(id); else bubble_down(id);
    }
    return val;
  }
};

struct Forest {
  World* w;
  sffo_forest_cfg cfg;
  Rng rng;
  std::vector<FNode> nodes;                 // global creation order (allNodes)
  std::vector<std::vector<int>> trees;      // per tree: node ids in insertion order
  std::vector<int> frontier, closed;
  std::map<std::pair<int, int>, std::vector<Border>> borders;
  std::vector<int> connected;
  int num_roots;  // problem.GetNumRoots(): roots + goal
  int goal_node = -1;
  int iter = 0;
  bool solved = false, empty_frontier = false;
  uint64_t path_free_calls = 0, nn_queries = 0, waves = 0, collide_base = 0;
  Grid grid;
  std::vector<std::vector<PHeap>> heaps;    // Tree::frontiers, only with priorityBias != 0
  bool use_priority() const { return cfg.priority_bias != 0; }
  bool tree_frontiers_empty(int t) const {   // Tree::EmptyFrontiers (src/primitives.h:542-556; the filter is never set)
    for (const PHeap& h : heaps[t]) if (!h.v.empty()) return false;
    return true;
  }
  bool all_frontiers_empty() const {
    for (size_t t = 0; t < heaps.size(); ++t) if (!tree_frontiers_empty((int)t)) return false;
    return true;
  }

  bool path_free(const double* a, const double* b) {
    ++path_free_calls;
    return w->path_free(a, b, nullptr, nullptr);
  }
  std::vector<Border>& border(int i, int j) {  // SymmetricMatrix, src/primitives.h:572-596
    if (i > j) std::swap(i, j);
    return borders[{i, j}];
  }
  int add_node(const double* pos, int tree, int parent, double dclosest, double droot, unsigned it) {
    FNode n;
    memcpy(n.pos, pos, sizeof n.pos);
    n.tree = tree;
    n.parent = parent;
    n.d_closest = dclosest;
    n.d_root = droot;
    n.iter = it;
    n.idx_in_tree = (int)trees[tree].size();
    int id = (int)nodes.size();
    nodes.push_back(n);
    trees[tree].push_back(id);
    grid.insert(pos, id);
    return id;
  }

  // all nodes with 6-D distance < r, grouped per tree (ascending tree id), each group
  // sorted by (distance, index in tree).  Replaces the per-tree FLANN radiusSearch of
  // src/forest.h:262-267 with an exact query (true metric, double).
  void radius_all(const double* q, double r, std::vector<std::vector<Hit>>& per_tree) {
    per_tree.assign(trees.size(), {});
    int64_t c[3];
    grid.coords(q, c);
    int reach = (int)std::ceil(r / grid.cell);
    for (int64_t x = c[0] - reach; x <= c[0] + reach; ++x)
      for (int64_t y = c[1] - reach; y <= c[1] + reach; ++y)
        for (int64_t z = c[2] - reach; z <= c[2] + reach; ++z) {
          auto it = grid.cells.find(Grid::key(x, y, z));
          if (it == grid.cells.end()) continue;
          for (int id : it->second) {
            // reference: realDist = neighbour->Position.distance(newPoint)  (src/forest.h:274)
            double d = distance6(nodes[id].pos, q);
            if (d < r) per_tree[nodes[id].tree].push_back({d, nodes[id].idx_in_tree});
          }
        }
    for (auto& v : per_tree) std::sort(v.begin(), v.end());
    nn_queries += trees.size();
  }

  // k nearest of one tree, sorted by (distance, index in tree)  (src/forest.h:317)
  void knn_tree(int tree, const double* q, size_t k, std::vector<Hit>& out) {
    out.clear();
    for (int id : trees[tree]) out.push_back({distance6(q, nodes[id].pos), nodes[id].idx_in_tree});
    if (out.size() > k) {
      std::partial_sort(out.begin(), out.begin() + k, out.end());
      out.resize(k);
    } else {
      std::sort(out.begin(), out.end());
    }
    ++nn_queries;
  }

  // src/forest.h:240-376 — returns true on rejection.  `words` are the raw engine draws.
  bool expand_node(int expanded, unsigned iteration, const uint64_t* words) {
    double np[6];
    bool result = sample_from_words(words, nodes[expanded].pos, cfg.sampling_dist, cfg.dim, cfg.limits, cfg.trig, np);
    if (!result || w->collide(np) || !path_free(nodes[expanded].pos, np)) return true;  // :246

    double parent_dist = distance6(nodes[expanded].pos, np);  // :250
    int my_tree = nodes[expanded].tree;
    // Reference radius is treeDistance + 2*SamplingDistance (:261) but only neighbours with
    // realDist < parentDistance - TOL (same tree) or < treeDistance - TOL (other tree) can act,
    // so the exact query uses r = max of the two thresholds; decisions are identical.
    double r = std::max(parent_dist, cfg.dist_tree);
    std::vector<std::vector<Hit>> per_tree;
    radius_all(np, r, per_tree);
    for (int j = 0; j < (int)trees.size(); ++j) {            // :262
      for (const Hit& h : per_tree[j]) {                      // :270
        int nb = trees[j][h.idx];
        double real = distance6(nodes[nb].pos, np);           // :274
        if (!nodes[expanded].force_children && real < (parent_dist - TOLERANCE) && j == my_tree &&
            path_free(nodes[nb].pos, np)) {                   // :276
          return true;
        }
        if (j != my_tree && real < (cfg.dist_tree - TOLERANCE)) {  // :283
          if (cfg.has_goal && pos_equal(nodes[nb].pos, cfg.goal)) {  // :286
            double g[6];
            memcpy(g, cfg.goal, sizeof g);
            solved = path_free(np, g);                        // :287
          } else if (!cfg.has_goal && path_free(nodes[expanded].pos, nodes[nb].pos)) {  // :288
            std::vector<Border>& bp = border(j, my_tree);
            int a = std::min(nb, expanded), b = std::max(nb, expanded);  // DistanceHolder orders by id (:609-616)
            bool found = false;
            for (const Border& x : bp) if (x.n1 == a && x.n2 == b) { found = true; break; }
            if (!found) {
              double d = nodes[nb].d_root + nodes[expanded].d_root + distance6(nodes[nb].pos, nodes[expanded].pos);
              bp.push_back({a, b, d});
            }
          }
          if (!solved) return true;                           // :296-299
        }
      }
    }

    int new_id;
    if (cfg.optimize) {                                        // :307-351
      double best = distance6(np, nodes[expanded].pos) + nodes[expanded].d_root;
      double ksff = 2 * M_E * std::log10((double)nodes.size());  // Node::globId (:309, primitives.h:484-486)
      std::vector<Hit> knn;
      knn_tree(my_tree, np, (size_t)ksff, knn);
      for (const Hit& h : knn) {                               // :320-327
        int nb = trees[my_tree][h.idx];
        double nd = distance6(np, nodes[nb].pos) + nodes[nb].d_root;
        if (nd < best - TOLERANCE && path_free(np, nodes[nb].pos)) {
          best = nd;
          expanded = nb;
        }
      }
      new_id = add_node(np, my_tree, expanded, distance6(np, nodes[expanded].pos), best, iteration);  // :329
      for (const Hit& h : knn) {                               // :332-350
        int nb = trees[my_tree][h.idx];
        double npd = distance6(nodes[nb].pos, np);
        double proposed = best + npd;
        if (proposed < nodes[nb].d_root - TOLERANCE && path_free(nodes[nb].pos, np)) {
          nodes[nb].parent = new_id;
          nodes[nb].d_closest = npd;
          nodes[nb].d_root = proposed;   // descendants are NOT updated (Appendix A.6)
        }
      }
    } else {
      new_id = add_node(np, my_tree, expanded, parent_dist, parent_dist + nodes[expanded].d_root, iteration);  // :353
    }
    if (use_priority()) {                                      // :360-363
      for (PHeap& h : heaps[my_tree]) h.push(new_id);
    } else {
      frontier.push_back(new_id);                              // :365
    }
    if (solved) {                                              // :369-372
      double d = distance6(np, cfg.goal);
      border(num_roots - 1, my_tree).push_back({std::min(new_id, goal_node), std::max(new_id, goal_node),
                                                nodes[new_id].d_root + d});
    }
    return false;
  }
  static bool pos_equal(const double* a, const double* b) {
    for (int i = 0; i < 6; ++i) if (a[i] != b[i]) return false;
    return true;
  }

  // src/forest.h:379-418
  int max_connected() {
    int max_conn = 0, remaining = num_roots;
    std::vector<char> conn(num_roots, 0);
    int unconnected = 0;
    while (max_conn < remaining) {
      connected.clear();
      std::vector<int> stack{unconnected};
      conn[unconnected] = 1;
      while (!stack.empty()) {
        int root = stack.front();
        stack.erase(stack.begin());
        connected.push_back(root);
        for (int i = 0; i < num_roots; ++i) {
          if (root == i) continue;
          auto it = borders.find({std::min(root, i), std::max(root, i)});
          bool nonempty = it != borders.end() && !it->second.empty();
          if (nonempty && !conn[i]) {
            conn[i] = 1;
            stack.insert(stack.begin(), i);
          }
        }
      }
      max_conn = (int)connected.size();
      for (int i = 0; i < num_roots; ++i)
        if (!conn[i]) { unconnected = i; break; }
      remaining -= max_conn;
    }
    return max_conn;
  }


  // ---- path extraction (post-loop; src/forest.h:420-462 getPaths, src/problemStruct.h:184-253 getAllPaths)
  struct Holder {            // DistanceHolder, src/primitives.h:598-655
    int n1 = -1, n2 = -1;
    double dist = std::numeric_limits<double>::max();
    std::vector<int> plan;
    bool exists() const { return n1 >= 0; }
  };
  std::vector<Holder> nm;    // neighboringMatrix (symmetric, num_roots x num_roots)
  Holder& NM(int i, int j) { return nm[(size_t)std::min(i, j) * num_roots + std::max(i, j)]; }
  bool is_root(int n) const { return nodes[n].d_root == 0; }   // Node::IsRoot, src/primitives.h:476-478

  void get_paths() {
    nm.assign((size_t)num_roots * num_roots, Holder());
    for (int i = 0; i < num_roots; ++i)
      for (int j = i + 1; j < num_roots; ++j) {
        auto it = borders.find({i, j});
        if (it == borders.end() || it->second.empty()) continue;
        double best = -1;
        for (Border& b : it->second) {
          b.dist = nodes[b.n1].d_root + nodes[b.n2].d_root + distance6(nodes[b.n1].pos, nodes[b.n2].pos);  // UpdateDistance
          if (best == -1 || b.dist < best - TOLERANCE) {
            best = b.dist;
            Holder h;
            h.n1 = b.n1; h.n2 = b.n2; h.dist = b.dist;
            NM(i, j) = h;
          }
        }
        Holder& h = NM(i, j);
        std::vector<int> front;
        int n = h.n1;
        front.push_back(n);
        while (!is_root(n)) { n = nodes[n].parent; front.push_back(n); }
        std::reverse(front.begin(), front.end());
        h.plan = front;
        n = h.n2;
        h.plan.push_back(n);
        while (!is_root(n)) { n = nodes[n].parent; h.plan.push_back(n); }
        if (cfg.optimize)
          h.dist = nodes[h.n1].d_root + nodes[h.n2].d_root + distance6(nodes[h.n1].pos, nodes[h.n2].pos);
      }
  }
  double plan_length(const std::vector<int>& plan) const {   // Solver::computeDistance, src/problemStruct.h:170-181
    double d = 0;
    for (size_t k = 1; k < plan.size(); ++k) d += distance6(nodes[plan[k - 1]].pos, nodes[plan[k]].pos);
    return d;
  }
  Holder make_holder(int a, int b, double dist, std::vector<int> plan) {  // DistanceHolder(first, second, dist, plan)
    Holder h;
    h.dist = dist;
    if (a < b) { h.n1 = a; h.n2 = b; h.plan = plan; }
    else { h.n1 = b; h.n2 = a; std::reverse(plan.begin(), plan.end()); h.plan = plan; }
    return h;
  }
  void get_all_paths() {
    const int nc = (int)connected.size();
    for (int k = 0; k < nc; ++k) {
      int id3 = connected[k];
      for (int i = 0; i < nc; ++i) {
        int id1 = connected[i];
        if (i == k || !NM(id1, id3).exists()) continue;
        for (int j = 0; j < nc; ++j) {
          int id2 = connected[j];
          if (i == j || !NM(id2, id3).exists()) continue;
          const Holder h1 = NM(id1, id3), h2 = NM(id2, id3);
          std::vector<int> plan1 = h1.plan, plan2 = h2.plan;
          int node1, node2;
          if (nodes[h1.n1].tree == id1) node1 = h1.n1; else { node1 = h1.n2; std::reverse(plan1.begin(), plan1.end()); }
          if (nodes[h2.n1].tree == id2) node2 = h2.n1; else { node2 = h2.n2; std::reverse(plan2.begin(), plan2.end()); }
          int last = -1;
          while (!plan1.empty() && !plan2.empty() && plan1.back() == plan2.back()) {
            last = plan1.back();
            plan1.pop_back();
            plan2.pop_back();
          }
          std::vector<int> fin(plan1.begin(), plan1.end());
          fin.push_back(last);
          for (size_t q = plan2.size(); q-- > 0;) fin.push_back(plan2[q]);
          double d = plan_length(fin);
          if (d < NM(id1, id2).dist - TOLERANCE) NM(id1, id2) = make_holder(node1, node2, d, fin);
        }
      }
    }
  }

  // src/forest.h:464-511 smoothPaths: shortcut every path from its far end; indices restate the reference's
  // reverse-iterator walk (tempGoal = plan index g, testNode = index t running from the start of the plan)
  void smooth_paths() {
    for (int i = 0; i < num_roots; ++i)
      for (int j = i + 1; j < num_roots; ++j) {
        Holder& h = NM(i, j);
        if (!h.exists()) continue;
        std::vector<int>& plan = h.plan;
        int g = (int)plan.size() - 1;
        double prev_dist = h.dist;
        while (g > 0) {
          int p = 0, t = 0;
          double cum = 0;
          bool changed = false;
          while (t < g - 1) {
            if (p != t) cum += distance6(nodes[plan[t]].pos, nodes[plan[p]].pos);
            if (path_free(nodes[plan[t]].pos, nodes[plan[g]].pos)) { changed = true; break; }
            p = t;
            ++t;
          }
          if (t == g - 1) cum += distance6(nodes[plan[t]].pos, nodes[plan[p]].pos);
          if (changed) {
            double dif = prev_dist - cum - distance6(nodes[plan[t]].pos, nodes[plan[g]].pos);
            h.dist -= dif;
            plan.erase(plan.begin() + t + 1, plan.begin() + g);
          }
          prev_dist = cum;
          g = t;
        }
      }
  }

  bool budget_hit() const { return cfg.node_budget > 0 && (int)nodes.size() >= cfg.node_budget; }

  // src/forest.h:122-202, generalised to waves of cfg.wave slots; wave == 1 is the
  // reference loop verbatim (same RNG consumption order).
  void run(int max_waves) {
    struct Slot { int node; bool from_closed; bool failing; int tree = -1, heap = -1; };
    int done = 0;
    const int words_per = cfg.dim == 2 ? 1 : 6;
    while (!(solved || iter >= cfg.max_iterations || budget_hit())) {
      if (max_waves > 0 && done >= max_waves) break;
      ++done;
      ++waves;
      std::vector<Slot> slots;
      if (use_priority() && !empty_frontier) {                 // :126-147 priority frontier
        int pool = 0;
        for (auto& hs : heaps) if (!hs.empty()) pool += (int)hs[0].v.size();
        const int n_slots = std::max(1, std::min(cfg.wave, pool));
        for (int s = 0; s < n_slots; ++s) {
          if (all_frontiers_empty()) break;                    // every frontier node is already in a slot
          int t = -1;
          while (t < 0 || tree_frontiers_empty(t)) t = rng.rand_int(0, (int)trees.size() - 1);
          int hp = -1;
          while (hp < 0 || heaps[t][hp].v.empty()) hp = rng.rand_int(0, (int)heaps[t].size() - 1);
          PHeap& prior = heaps[t][hp];
          Slot sl;
          if (rng.prob() <= cfg.priority_bias) sl.node = prior.pop();
          else sl.node = prior.pop_at(rng.rand_int(0, (int)prior.v.size() - 1));
          sl.from_closed = false;
          sl.failing = true;
          sl.tree = t;
          sl.heap = hp;
          slots.push_back(sl);
        }
      } else {
      // a wave never holds more slots than the pool it draws from (wave == 1 unaffected)
      const int pool = (!closed.empty() && empty_frontier) ? (int)closed.size() : (int)frontier.size();
      const int n_slots = std::max(1, std::min(cfg.wave, pool));
      for (int s = 0; s < n_slots; ++s) {
        Slot sl;
        if (!closed.empty() && empty_frontier) {              // :138-141
          int pos = rng.rand_int(0, (int)closed.size() - 1);
          sl.node = closed[pos];
          sl.from_closed = true;
        } else {                                               // :149-150
          int pos = rng.rand_int(0, (int)frontier.size() - 1);
          sl.node = frontier[pos];
          sl.from_closed = false;
        }
        sl.failing = true;
        slots.push_back(sl);
      }
      }
      for (int round = 0; round < cfg.threshold_misses && !solved; ++round) {  // :155
        for (Slot& sl : slots) {
          if (!sl.failing || iter >= cfg.max_iterations || solved) continue;
          uint64_t words[6];
          for (int k = 0; k < words_per; ++k) words[k] = rng.raw();
          ++iter;
          sl.failing = expand_node(sl.node, (unsigned)iter, words);
        }
      }
      for (Slot& sl : slots) {                                 // :160-181
        if (use_priority() && sl.tree >= 0) {
          if (sl.failing) {                                    // exhausted: drop it from the tree's other heaps too
            for (int i = (int)heaps[sl.tree].size() - 1; i > -1; --i) {
              if (i == sl.heap) continue;
              PHeap& p = heaps[sl.tree][i];
              for (int j = (int)p.v.size() - 1; j > -1; --j)
                if (j < (int)p.v.size() && p.v[j] == sl.node) p.pop_at(j);
            }
            if (!nodes[sl.node].force_children) {
              nodes[sl.node].force_children = true;
              closed.push_back(sl.node);
            }
          } else {
            heaps[sl.tree][sl.heap].push(sl.node);             // :179-181 back onto the heap it was taken from
          }
          continue;
        }
        if (sl.failing && !sl.from_closed) {
          auto it = std::find(frontier.begin(), frontier.end(), sl.node);
          if (it != frontier.end()) {
            frontier.erase(it);
            nodes[sl.node].force_children = true;
            closed.push_back(sl.node);
          }
        }
      }
      if (!solved && use_priority()) empty_frontier = all_frontiers_empty();  // :184-191
      else
      empty_frontier = frontier.empty();                       // :193
      if (!solved) {                                           // :196-201
        bool conn = max_connected() == num_roots;
        solved = (!cfg.has_goal && empty_frontier && conn);
      } else {
        max_connected();
      }
    }
  }
};


// ------------------------------------------------------------------ RRT / RRT* / Multi-T-RRT
// src/rrt.h:47-83 (constructor), :86-99 (Solve loop), :128-322 (expandNode incl. tree merging).
// A node keeps ONE identity (its global id); the reference's merge copies the `from` tree's nodes
// into `to` (:243-249) and re-resolves parents / children / links by id (:251-296) — restated as
// moving the ids to the end of `to`'s list in `from` order.  Neighbour queries are exact (true
// metric); the reference's merge fills only 2 of 6 FLANN columns (:243-245), which is not kept.
struct RNode {
  double pos[6];
  int root_tree;   // Node::Root (tree it was created in; RRT* rewire copies it, :195)
  int tree;        // Node::ExpandedRoot: the tree whose list currently holds it
  int parent;
  int idx_in_tree;
  double d_closest, d_root;
  unsigned iter;
};
struct RLink { int n1, n2; double dist; };

struct Rrt {
  World* w;
  sffo_rrt_cfg cfg;
  Rng rng;
  std::vector<RNode> nodes;
  std::vector<std::vector<int>> trees;       // tree id -> node ids in list order
  std::vector<std::vector<RLink>> links;     // Tree::links
  std::vector<std::vector<int>> eaten;       // Tree::eaten
  std::vector<int> tree_frontier;            // treeFrontier (tree ids)
  int num_trees = 0;                          // numTrees
  int goal_node = -1;
  int iter = 0;
  bool solved = false;
  uint64_t path_free_calls = 0, nn_queries = 0, collide_base = 0, merges = 0;
  int lazy_last = -1;                                   // lazy_edge: the node that reached the goal
  double lazy_distance = std::numeric_limits<double>::max();

  bool path_free(const double* a, const double* b) {
    ++path_free_calls;
    return w->path_free(a, b, nullptr, nullptr);
  }
  int add_node(const double* pos, int root_tree, int tree, int parent, double dc, double dr, unsigned it) {
    RNode n;
    memcpy(n.pos, pos, sizeof n.pos);
    n.root_tree = root_tree; n.tree = tree; n.parent = parent; n.d_closest = dc; n.d_root = dr; n.iter = it;
    n.idx_in_tree = (int)trees[tree].size();
    int id = (int)nodes.size();
    nodes.push_back(n);
    trees[tree].push_back(id);
    return id;
  }
  void knn(int tree, const double* q, size_t k, std::vector<Hit>& out) {
    out.clear();
    for (int id : trees[tree]) out.push_back({distance6(q, nodes[id].pos), nodes[id].idx_in_tree});
    if (out.size() > k) { std::partial_sort(out.begin(), out.begin() + k, out.end()); out.resize(k); }
    else std::sort(out.begin(), out.end());
    ++nn_queries;
  }
  RLink make_link(int a, int b) {   // DistanceHolder(first, second), src/primitives.h:609-618
    double d = nodes[a].d_root + nodes[b].d_root + distance6(nodes[a].pos, nodes[b].pos);
    return {std::min(a, b), std::max(a, b), d};
  }

  void expand(int tree_to_expand, unsigned iteration) {
    double rnd[6], np[6];
    if (cfg.priority_bias != 0 && rng.prob() <= cfg.priority_bias) {          // :130-131
      memcpy(rnd, nodes[goal_node].pos, sizeof rnd);
    } else {                                                                    // :133, src/randGen.h:124-146
      double y = uniform_real_from_word(rng.raw(), cfg.limits[2], cfg.limits[3]);
      double x = uniform_real_from_word(rng.raw(), cfg.limits[0], cfg.limits[1]);
      rnd[0] = x; rnd[1] = y; rnd[2] = 0; rnd[3] = rnd[4] = rnd[5] = 0;
      if (cfg.dim == 6) {
        rnd[2] = uniform_real_from_word(rng.raw(), cfg.limits[4], cfg.limits[5]);
        rnd[3] = uniform_real_from_word(rng.raw(), -M_PI, M_PI);
        double phi = tacos(1 - 2 * uniform_real_from_word(rng.raw(), 0.0, 1.0), cfg.trig) + M_PI_2;
        if (uniform_real_from_word(rng.raw(), 0.0, 1.0) < 0.5) { if (phi < 0) phi += M_PI; else phi -= M_PI; }
        rnd[4] = phi;
        rnd[5] = uniform_real_from_word(rng.raw(), -M_PI, M_PI);
      }
    }
    std::vector<Hit> hits;
    knn(tree_to_expand, rnd, 1, hits);                                          // :143
    int nearest = trees[tree_to_expand][hits[0].idx];
    steer6(nodes[nearest].pos, rnd, cfg.sampling_dist, np);                     // :148
    if (w->collide(np) || !path_free(nodes[nearest].pos, np)) return;           // :149-151
    int new_id;
    if (cfg.optimize) {                                                         // :156-201
      double best = distance6(np, nodes[nearest].pos) + nodes[nearest].d_root;
      double krrt = 2 * M_E * std::log10((double)nodes.size() + (cfg.lazy_edge ? 1.0 : 0.0));   // src/lazy.h:199
      std::vector<Hit> kn;
      knn(tree_to_expand, np, (size_t)krrt, kn);
      for (const Hit& h : kn) {
        int nb = trees[tree_to_expand][h.idx];
        double nd = distance6(np, nodes[nb].pos) + nodes[nb].d_root;
        if (nd < best - TOLERANCE && path_free(np, nodes[nb].pos)) { best = nd; nearest = nb; }
      }
      new_id = add_node(np, nodes[nearest].root_tree, tree_to_expand, nearest, distance6(nodes[nearest].pos, np), best, iteration);
      for (const Hit& h : kn) {
        int nb = trees[tree_to_expand][h.idx];
        double npd = distance6(nodes[nb].pos, np);
        double proposed = best + npd;
        if (proposed < nodes[nb].d_root - TOLERANCE && path_free(nodes[nb].pos, np)) {
          nodes[nb].parent = new_id;
          nodes[nb].root_tree = nodes[new_id].root_tree;                        // :195
          nodes[nb].d_closest = npd;
          nodes[nb].d_root = proposed;
        }
      }
    } else {                                                                    // :203
      new_id = add_node(np, nodes[nearest].root_tree, tree_to_expand, nearest, cfg.sampling_dist,
                        nodes[nearest].d_root + cfg.sampling_dist, iteration);
    }
    if (cfg.lazy_edge) {                                                        // src/lazy.h:258-273
      const double gd = distance6(cfg.goal, nodes[new_id].pos);
      if (gd < cfg.dist_tree) {
        solved = true;
        lazy_distance = gd + nodes[new_id].d_root;
        lazy_last = new_id;
      }
      return;
    }
    // :219-319 connect to / merge with the other live trees
    for (int i = 0; i < (int)tree_frontier.size(); ++i) {
      int tree = tree_frontier[i];
      if (tree == tree_to_expand) continue;
      knn(tree, np, 1, hits);
      int nb = trees[tree][hits[0].idx];
      double nd = distance6(nodes[nb].pos, np);
      if (nd < cfg.dist_tree && path_free(np, nodes[nb].pos)) {                  // :231 (no TOLERANCE here)
        links[tree_to_expand].push_back(make_link(new_id, nb));                  // :233
        int nbt = nodes[nb].tree;
        int to = tree_to_expand < nbt ? tree_to_expand : nbt;
        int from = tree_to_expand < nbt ? nbt : tree_to_expand;
        for (int id : trees[from]) {                                             // :240-250
          nodes[id].tree = to;
          nodes[id].idx_in_tree = (int)trees[to].size();
          trees[to].push_back(id);
        }
        for (RLink& l : links[to]) l = make_link(l.n1, l.n2);                    // :278-289 re-created -> distance recomputed
        for (const RLink& l : links[from]) links[to].push_back(make_link(l.n1, l.n2));  // :291-299
        eaten[to].push_back(from);                                               // :305-308
        for (int t : eaten[from]) eaten[to].push_back(t);
        tree_frontier.erase(std::find(tree_frontier.begin(), tree_frontier.end(), from));  // :310-315
        tree_to_expand = to;
        solved = tree_frontier.size() == 1;
        --num_trees;
        --i;
        ++merges;
      }
    }
  }

  // RapidExpTree::smoothPaths (src/rrt.h:354-379): the same far-end shortcutting as the forest's, but on the
  // plans stored in the central tree's LINKS.  neighboringMatrix received copies of those plans in getPaths
  // (:350) and every writer reads the matrix, so the reference's outputs do not change; the shortened link
  // plans are what this restatement exposes.  Reverse-iterator walk restated with forward indices
  // (tempGoal = g, testNode = t); the erase is [t+1, g).
  std::vector<std::vector<int>> link_plans;
  uint64_t smooth_path_free_calls = 0;
  void smooth_paths() {
    for (std::vector<int>& plan : link_plans) {
      int g = (int)plan.size() - 1;
      while (g > 0) {
        int t = 0;
        bool changed = false;
        while (t < g - 1) {
          ++smooth_path_free_calls;
          if (path_free(nodes[plan[t]].pos, nodes[plan[g]].pos)) { changed = true; break; }
          ++t;
        }
        if (changed) plan.erase(plan.begin() + t + 1, plan.begin() + g);
        g = t;
      }
    }
  }

  // post-loop: getConnectedTrees (src/rrt.h:381-393), getPaths (:324-352), Solver::getAllPaths (problemStruct.h:184-253)
  struct PH { int n1 = -1, n2 = -1; double dist = std::numeric_limits<double>::max(); std::vector<int> plan; };
  std::vector<PH> nm;
  std::vector<int> connected;
  PH& NM(int i, int j) { int nt = (int)trees.size(); return nm[(size_t)std::min(i, j) * nt + std::max(i, j)]; }
  void get_paths() {
    const int nt = (int)trees.size();
    nm.assign((size_t)nt * nt, PH());
    connected.clear();
    size_t max_conn = 0;
    int central = 0;
    const int num_roots = cfg.has_goal ? num_trees + 2 : num_trees + 1;
    for (int i = 0; i < num_roots && i < nt; ++i)
      if (eaten[i].size() > max_conn) { max_conn = eaten[i].size(); central = i; connected = eaten[i]; connected.push_back(i); }
    link_plans.clear();
    for (const RLink& link : links[central]) {
      PH h;
      h.n1 = link.n1; h.n2 = link.n2; h.dist = link.dist;
      std::vector<int> chain;
      for (int n = link.n1;; n = nodes[n].parent) { chain.push_back(n); if (nodes[n].d_root == 0) break; }
      h.plan.assign(chain.rbegin(), chain.rend());
      for (int n = link.n2;; n = nodes[n].parent) { h.plan.push_back(n); if (nodes[n].d_root == 0) break; }
      NM(nodes[link.n1].root_tree, nodes[link.n2].root_tree) = h;
      link_plans.push_back(h.plan);   // DistanceHolder::plan of the link itself (the matrix holds a COPY, :350)
    }
    const int nc = (int)connected.size();
    for (int k = 0; k < nc; ++k) {
      int id3 = connected[k];
      for (int i = 0; i < nc; ++i) {
        int id1 = connected[i];
        if (i == k || NM(id1, id3).n1 < 0) continue;
        for (int j = 0; j < nc; ++j) {
          int id2 = connected[j];
          if (i == j || NM(id2, id3).n1 < 0) continue;
          const PH h1 = NM(id1, id3), h2 = NM(id2, id3);
          std::vector<int> plan1 = h1.plan, plan2 = h2.plan;
          int node1, node2;
          if (nodes[h1.n1].root_tree == id1) node1 = h1.n1; else { node1 = h1.n2; std::reverse(plan1.begin(), plan1.end()); }
          if (nodes[h2.n1].root_tree == id2) node2 = h2.n1; else { node2 = h2.n2; std::reverse(plan2.begin(), plan2.end()); }
          int last = -1;
          while (!plan1.empty() && !plan2.empty() && plan1.back() == plan2.back()) { last = plan1.back(); plan1.pop_back(); plan2.pop_back(); }
          std::vector<int> fin(plan1.begin(), plan1.end());
          fin.push_back(last);
          for (size_t q = plan2.size(); q-- > 0;) fin.push_back(plan2[q]);
          double d = 0;
          for (size_t q = 1; q < fin.size(); ++q) d += distance6(nodes[fin[q - 1]].pos, nodes[fin[q]].pos);
          if (d < NM(id1, id2).dist - TOLERANCE) {
            PH h;
            h.dist = d;
            if (node1 < node2) { h.n1 = node1; h.n2 = node2; h.plan = fin; }
            else { h.n1 = node2; h.n2 = node1; h.plan.assign(fin.rbegin(), fin.rend()); }
            NM(id1, id2) = h;
          }
        }
      }
    }
  }

  void run(int max_iters) {
    int done = 0;
    while (!(solved || iter == cfg.max_iterations)) {                            // :93
      if (max_iters > 0 && done >= max_iters) break;
      ++done;
      ++iter;
      int tree = cfg.lazy_edge ? 0 : tree_frontier[rng.rand_int(0, num_trees)];  // :95 (src/lazy.h:181 draws no tree)
      expand(tree, (unsigned)iter);
    }
  }
};

}  // namespace

// =================================================================== C interface
extern "C" {

int sffo_parse_obj(const char* path, const double pos[3], double scale, double* tri9, int cap) {
  std::vector<double> t;
  int n;
  try { n = parse_obj(path, pos, scale, t); } catch (...) { return -1; }
  if (n < 0) return -1;
  if (n > cap) return -2;
  memcpy(tri9, t.data(), t.size() * sizeof(double));
  return n;
}
int sffo_parse_tri2d(const char* path, const double pos[3], double scale, double* tri9, int cap) {
  std::vector<double> t;
  int n;
  try { n = parse_tri2d(path, pos, scale, t); } catch (...) { return -1; }
  if (n < 0) return -1;
  if (n > cap) return -2;
  memcpy(tri9, t.data(), t.size() * sizeof(double));
  return n;
}

double sffo_distance(const double a[6], const double b[6]) { return distance6(a, b); }
void sffo_steer(const double from[6], const double to[6], double dist, double out[6]) { steer6(from, to, dist, out); }
void sffo_rotation(const double p[6], int trig, double R[9]) { rotation(p, trig, R); }
double sffo_sin(double x, int trig) { return tsin(x, trig); }
double sffo_cos(double x, int trig) { return tcos(x, trig); }
double sffo_acos(double x, int trig) { return tacos(x, trig); }

struct sffo_rng { Rng r; };
sffo_rng* sffo_rng_create(uint64_t seed, const double limits[6], int trig) {
  sffo_rng* g = new sffo_rng;
  g->r.eng.reseed(seed);
  memcpy(g->r.lim, limits, sizeof g->r.lim);
  g->r.trig = trig;
  return g;
}
void sffo_rng_destroy(sffo_rng* g) { delete g; }
uint64_t sffo_rng_raw(sffo_rng* g) { return g->r.raw(); }
int sffo_rng_int(sffo_rng* g, int lo, int hi) { return g->r.rand_int(lo, hi); }
double sffo_rng_prob(sffo_rng* g) { return g->r.prob(); }
int sffo_rng_point_in_distance(sffo_rng* g, const double center[6], double dist, int dim, double out[6]) {
  uint64_t w[6];
  int n = dim == 2 ? 1 : 6;
  for (int i = 0; i < n; ++i) w[i] = g->r.raw();
  return sample_from_words(w, center, dist, dim, g->r.lim, g->r.trig, out) ? 1 : 0;
}
// src/randGen.h:124-146.  g++ evaluates the two arguments of point.set(uniSpaceX(..), uniSpaceY(..), 0)
// right to left: Y is drawn first (pinned by tests/golden/ref_primitives.json).
void sffo_rng_point_in_space(sffo_rng* g, int dim, double out[6]) {
  Rng& r = g->r;
  double y = uniform_real_from_word(r.raw(), r.lim[2], r.lim[3]);
  double x = uniform_real_from_word(r.raw(), r.lim[0], r.lim[1]);
  out[0] = x; out[1] = y; out[2] = 0; out[3] = out[4] = out[5] = 0;
  if (dim == 6) {
    out[2] = uniform_real_from_word(r.raw(), r.lim[4], r.lim[5]);
    out[3] = uniform_real_from_word(r.raw(), -M_PI, M_PI);
    double phi = tacos(1 - 2 * uniform_real_from_word(r.raw(), 0.0, 1.0), r.trig) + M_PI_2;
    if (uniform_real_from_word(r.raw(), 0.0, 1.0) < 0.5) {
      if (phi < 0) phi += M_PI; else phi -= M_PI;
    }
    out[4] = phi;
    out[5] = uniform_real_from_word(r.raw(), -M_PI, M_PI);
  }
}
int sffo_sample_from_words(const uint64_t* words, const double center[6], double dist, int dim,
                           const double limits[6], int trig, double out[6]) {
  return sample_from_words(words, center, dist, dim, limits, trig, out) ? 1 : 0;
}

struct sffo_world { World w; };
sffo_world* sffo_world_create(const double* env_tri9, int n_env, const double* robot_tri9, int n_robot, int trig) {
  sffo_world* h = new sffo_world;
  h->w.env.assign(env_tri9, env_tri9 + (size_t)n_env * 9);
  h->w.robot.assign(robot_tri9, robot_tri9 + (size_t)n_robot * 9);
  h->w.trig = trig;
  h->w.init();
  return h;
}
void sffo_world_destroy(sffo_world* h) { delete h; }
int sffo_tri_contact(const double P[9], const double Q[9]) { return tri_contact(P, Q) ? 1 : 0; }
int sffo_collide_pose_brute(sffo_world* h, const double p[6]) { return h->w.collide_brute(p) ? 1 : 0; }
int sffo_collide_pose(sffo_world* h, const double p[6]) { return h->w.collide(p) ? 1 : 0; }
int sffo_path_free(sffo_world* h, const double a[6], const double b[6], int* first_hit, int* n_samples) {
  return h->w.path_free(a, b, first_hit, n_samples) ? 1 : 0;
}
uint64_t sffo_world_collide_calls(sffo_world* h) { return h->w.collide_calls; }

int sffo_radius(const double* pts, int n, const double q[6], double r, int32_t* idx, double* dist, int cap) {
  std::vector<Hit> hits;
  for (int i = 0; i < n; ++i) {
    double d = distance6(pts + 6 * i, q);
    if (d < r) hits.push_back({d, i});
  }
  std::sort(hits.begin(), hits.end());
  int m = std::min((int)hits.size(), cap);
  for (int i = 0; i < m; ++i) { idx[i] = hits[i].idx; if (dist) dist[i] = hits[i].d; }
  return (int)hits.size();
}
// The priority-frontier heap on its own (test entry): a heap over nodes 0 .. n_initial-1 keyed by the distance to
// `ref` (Tree::AddFrontier, src/primitives.h:530-540 -> Heap ctor + sort, src/heap.h:72-83,107-113), then a script of
// operations - kind 0 pop (:189-207), 1/2 pop at index arg (:209-238), 3 push of the next unused node (:175-187).
// ret[i] = node returned / pushed, state[i * cap ..] = heap array after the operation (-1 padded).
int sffo_heap_script(const double* pos6, int n_total, int n_initial, const double ref[6], const int32_t* ops, int n_ops,
                     int32_t* initial, int32_t* ret, int32_t* state, int cap) {
  std::vector<FNode> nodes((size_t)n_total);
  for (int i = 0; i < n_total; ++i) memcpy(nodes[i].pos, pos6 + 6 * (size_t)i, sizeof nodes[i].pos);
  PHeap hp;
  hp.nodes = &nodes;
  memcpy(hp.ref, ref, sizeof hp.ref);
  for (int i = 0; i < n_initial; ++i) hp.v.push_back(i);
  for (int k = (int)hp.v.size() - 1; k >= 0; --k) hp.bubble_down(k);
  for (int j = 0; j < cap; ++j) initial[j] = j < (int)hp.v.size() ? hp.v[j] : -1;
  int next = n_initial;
  for (int i = 0; i < n_ops; ++i) {
    const int kind = ops[2 * i], arg = ops[2 * i + 1];
    int r = -1;
    if (kind == 0) { if (hp.v.empty()) return -1; r = hp.pop(); }
    else if (kind == 1 || kind == 2) { if (arg < 0 || arg >= (int)hp.v.size()) return -1; r = hp.pop_at(arg); }
    else { if (next >= n_total) return -1; r = next; hp.push(next++); }
    ret[i] = r;
    for (int j = 0; j < cap; ++j) state[(size_t)i * cap + j] = j < (int)hp.v.size() ? hp.v[j] : -1;
  }
  return 0;
}

int sffo_knn(const double* pts, int n, const double q[6], int k, int32_t* idx, double* dist) {
  std::vector<Hit> hits(n);
  for (int i = 0; i < n; ++i) hits[i] = {distance6(q, pts + 6 * i), i};
  int m = std::min(n, k);
  std::partial_sort(hits.begin(), hits.begin() + m, hits.end());
  for (int i = 0; i < m; ++i) { idx[i] = hits[i].idx; if (dist) dist[i] = hits[i].d; }
  return m;
}

struct sffo_forest { Forest f; };
// src/forest.h:57-110 (constructor): one tree per root, roots on the frontier; the goal is
// an extra single-node tree that is searched but never expanded.
sffo_forest* sffo_forest_create(sffo_world* w, const sffo_forest_cfg* cfg, const double* roots6, int n_roots) {
  sffo_forest* h = new sffo_forest;
  Forest& f = h->f;
  f.w = &w->w;
  f.collide_base = w->w.collide_calls;
  f.cfg = *cfg;
  if (f.cfg.wave < 1) f.cfg.wave = 1;
  f.rng.eng.reseed(cfg->seed);
  memcpy(f.rng.lim, cfg->limits, sizeof f.rng.lim);
  f.rng.trig = cfg->trig;
  f.grid.cell = std::max(cfg->sampling_dist, cfg->dist_tree);
  f.num_roots = n_roots + (cfg->has_goal ? 1 : 0);
  f.trees.resize(f.num_roots);
  for (int j = 0; j < n_roots; ++j) {
    int id = f.add_node(roots6 + 6 * j, j, -1, 0, 0, 0);
    f.frontier.push_back(id);
  }
  if (cfg->has_goal) f.goal_node = f.add_node(cfg->goal, n_roots, -1, 0, 0, 0);
  if (f.use_priority()) {                      // Tree::AddFrontier (src/primitives.h:530-540) as called at src/forest.h:78-88,104-108
    f.heaps.resize(f.num_roots);
    auto add_heap = [&](int tree, int ref_node) {
      PHeap hp;
      hp.nodes = &f.nodes;
      memcpy(hp.ref, f.nodes[ref_node].pos, sizeof hp.ref);
      for (int id : f.trees[tree]) hp.v.push_back(id);
      for (int k = (int)hp.v.size() - 1; k >= 0; --k) hp.bubble_down(k);   // Heap::sort
      f.heaps[tree].push_back(hp);
    };
    if (!cfg->has_goal) {
      for (int i = 0; i < n_roots; ++i)
        for (int j = 0; j < n_roots; ++j)
          if (i != j) add_heap(i, f.trees[j][0]);
    } else {
      for (int i = 0; i < n_roots; ++i) add_heap(i, f.goal_node);
    }
  }
  return h;
}
void sffo_forest_destroy(sffo_forest* h) { delete h; }
void sffo_forest_run(sffo_forest* h, int max_waves) { h->f.run(max_waves); }
void sffo_forest_get_stats(sffo_forest* h, sffo_forest_stats* s) {
  Forest& f = h->f;
  s->iterations = f.iter;
  bool solved = f.solved;
  // src/forest.h:204-206
  if (!solved && !f.cfg.has_goal) solved = f.max_connected() == f.num_roots;
  s->solved = solved;
  s->n_nodes = (int)f.nodes.size();
  s->n_trees = (int)f.trees.size();
  s->frontier_size = (int)f.frontier.size();
  s->closed_size = (int)f.closed.size();
  s->n_connected = (int)f.connected.size();
  int nb = 0;
  for (auto& kv : f.borders) nb += (int)kv.second.size();
  s->n_borders = nb;
  s->collide_calls = f.w->collide_calls - f.collide_base;
  s->path_free_calls = f.path_free_calls;
  s->nn_queries = f.nn_queries;
  s->waves = f.waves;
}
void sffo_forest_get_nodes(sffo_forest* h, double* pos6, int32_t* parent, int32_t* tree, int32_t* iter,
                           double* cost, double* dpar) {
  Forest& f = h->f;
  for (size_t i = 0; i < f.nodes.size(); ++i) {
    const FNode& n = f.nodes[i];
    if (pos6) memcpy(pos6 + 6 * i, n.pos, sizeof n.pos);
    if (parent) parent[i] = n.parent;
    if (tree) tree[i] = n.tree;
    if (iter) iter[i] = (int32_t)n.iter;
    if (cost) cost[i] = n.d_root;
    if (dpar) dpar[i] = n.d_closest;
  }
}
int sffo_forest_get_borders(sffo_forest* h, int32_t* ta, int32_t* tb, int32_t* n1, int32_t* n2, double* dist, int cap) {
  int k = 0;
  for (auto& kv : h->f.borders)
    for (const Border& b : kv.second) {
      if (k < cap) {
        ta[k] = kv.first.first; tb[k] = kv.first.second; n1[k] = b.n1; n2[k] = b.n2; dist[k] = b.dist;
      }
      ++k;
    }
  return k;
}
/* post-loop path extraction: dist = num_roots x num_roots matrix (max double where no path); returns num_roots */
int sffo_forest_paths(sffo_forest* h, double* dist) {
  Forest& f = h->f;
  if (!f.solved && !f.cfg.has_goal) f.solved = f.max_connected() == f.num_roots; else f.max_connected();
  f.get_paths();
  f.get_all_paths();
  for (int i = 0; i < f.num_roots; ++i)
    for (int j = 0; j < f.num_roots; ++j) dist[(size_t)i * f.num_roots + j] = i == j ? 0.0 : f.NM(i, j).dist;
  return f.num_roots;
}
int sffo_forest_smooth(sffo_forest* h, double* dist) {
  Forest& f = h->f;
  if (f.nm.empty()) return -1;
  f.smooth_paths();
  for (int i = 0; i < f.num_roots; ++i)
    for (int j = 0; j < f.num_roots; ++j) dist[(size_t)i * f.num_roots + j] = i == j ? 0.0 : f.NM(i, j).dist;
  return f.num_roots;
}
int sffo_forest_path_plan(sffo_forest* h, int i, int j, int32_t* node_ids, int cap) {
  Forest& f = h->f;
  if (f.nm.empty() || i == j) return 0;
  const auto& p = f.NM(i, j).plan;
  for (size_t k = 0; k < p.size() && (int)k < cap; ++k) node_ids[k] = p[k];
  return (int)p.size();
}
uint64_t sffo_forest_fingerprint(sffo_forest* h) {
  uint64_t x = 1469598103934665603ULL;
  auto mix = [&](const void* p, size_t n) {
    const unsigned char* c = (const unsigned char*)p;
    for (size_t i = 0; i < n; ++i) { x ^= c[i]; x *= 1099511628211ULL; }
  };
  for (const FNode& n : h->f.nodes) {
    int32_t v[3] = {n.parent, n.tree, (int32_t)n.iter};
    mix(v, sizeof v);
    mix(n.pos, sizeof n.pos);
  }
  return x;
}

struct sffo_rrt { Rrt r; };
sffo_rrt* sffo_rrt_create(sffo_world* w, const sffo_rrt_cfg* cfg, const double* roots6, int n_roots) {
  sffo_rrt* h = new sffo_rrt;
  Rrt& r = h->r;
  r.w = &w->w;
  r.collide_base = w->w.collide_calls;
  r.cfg = *cfg;
  r.rng.eng.reseed(cfg->seed);
  for (uint64_t k = 0; k < cfg->rng_skip; ++k) (void)r.rng.raw();
  memcpy(r.rng.lim, cfg->limits, sizeof r.rng.lim);
  r.rng.trig = cfg->trig;
  int nt = n_roots + (cfg->has_goal ? 1 : 0);
  r.trees.resize(nt); r.links.resize(nt); r.eaten.resize(nt);
  for (int j = 0; j < n_roots; ++j) {                                            // src/rrt.h:48-62
    r.add_node(roots6 + 6 * j, j, j, -1, 0, 0, 0);
    r.tree_frontier.push_back(j);
  }
  r.num_trees = n_roots - 1;                                                     // :63
  if (cfg->has_goal) {                                                           // :66-82
    r.goal_node = r.add_node(cfg->goal, n_roots, n_roots, -1, 0, 0, 0);
    r.tree_frontier.push_back(n_roots);
  }
  return h;
}
void sffo_rrt_destroy(sffo_rrt* h) { delete h; }
void sffo_rrt_run(sffo_rrt* h, int max_iters) { h->r.run(max_iters); }
void sffo_rrt_get_stats(sffo_rrt* h, sffo_rrt_stats* s) {
  Rrt& r = h->r;
  s->iterations = r.iter;
  s->solved = r.solved;
  s->n_nodes = (int)r.nodes.size();
  s->n_live_trees = (int)r.tree_frontier.size();
  s->merges = (int)r.merges;
  int nl = 0;
  for (auto& l : r.links) nl += (int)l.size();
  s->n_links = nl;
  s->collide_calls = r.w->collide_calls - r.collide_base;
  s->path_free_calls = r.path_free_calls;
  s->nn_queries = r.nn_queries;
  s->rng_draws = r.rng.eng.n_drawn;
  s->lazy_distance = r.lazy_distance;
}
int sffo_rrt_lazy_plan(sffo_rrt* h, int32_t* node_ids, int cap) {
  Rrt& r = h->r;
  if (r.lazy_last < 0) return 0;
  std::vector<int> chain;
  for (int n = r.lazy_last;; n = r.nodes[n].parent) { chain.push_back(n); if (r.nodes[n].d_root == 0) break; }   // IsRoot(), src/primitives.h:476
  std::reverse(chain.begin(), chain.end());
  for (int k = 0; k < (int)chain.size() && k < cap; ++k) node_ids[k] = chain[k];
  return (int)chain.size();
}
void sffo_rrt_get_nodes(sffo_rrt* h, double* pos6, int32_t* parent, int32_t* tree, int32_t* root_tree, int32_t* iter,
                        double* cost, double* dpar) {
  Rrt& r = h->r;
  for (size_t i = 0; i < r.nodes.size(); ++i) {
    const RNode& n = r.nodes[i];
    if (pos6) memcpy(pos6 + 6 * i, n.pos, sizeof n.pos);
    if (parent) parent[i] = n.parent;
    if (tree) tree[i] = n.tree;
    if (root_tree) root_tree[i] = n.root_tree;
    if (iter) iter[i] = (int32_t)n.iter;
    if (cost) cost[i] = n.d_root;
    if (dpar) dpar[i] = n.d_closest;
  }
}
int sffo_rrt_paths(sffo_rrt* h, double* dist) {
  Rrt& r = h->r;
  r.get_paths();
  const int nt = (int)r.trees.size();
  for (int i = 0; i < nt; ++i)
    for (int j = 0; j < nt; ++j) dist[(size_t)i * nt + j] = i == j ? 0.0 : r.NM(i, j).dist;
  return (int)r.connected.size();
}
int sffo_rrt_path_plan(sffo_rrt* h, int i, int j, int32_t* node_ids, int cap) {
  Rrt& r = h->r;
  if (r.nm.empty() || i == j) return 0;
  const auto& p = r.NM(i, j).plan;
  for (size_t k = 0; k < p.size() && (int)k < cap; ++k) node_ids[k] = p[k];
  return (int)p.size();
}
int sffo_rrt_smooth(sffo_rrt* h) {
  h->r.smooth_paths();
  return (int)h->r.link_plans.size();
}
int sffo_rrt_link_plan(sffo_rrt* h, int k, int32_t* node_ids, int cap) {
  Rrt& r = h->r;
  if (k < 0 || k >= (int)r.link_plans.size()) return -1;
  const auto& p = r.link_plans[k];
  for (size_t q = 0; q < p.size() && (int)q < cap; ++q) node_ids[q] = p[q];
  return (int)p.size();
}
int sffo_rrt_get_links(sffo_rrt* h, int32_t* tree, int32_t* n1, int32_t* n2, double* dist, int cap) {
  int k = 0;
  for (size_t t = 0; t < h->r.links.size(); ++t)
    for (const RLink& l : h->r.links[t]) {
      if (k < cap) { tree[k] = (int32_t)t; n1[k] = l.n1; n2[k] = l.n2; dist[k] = l.dist; }
      ++k;
    }
  return k;
}

}  // extern "C"
