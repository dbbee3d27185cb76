// sff_geom.h — geometry of the SFF/RRT hot path shared by the gfx950 kernels and the host
// engine of the shipped library.  Everything is written with plain IEEE-754 double
// arithmetic in a fixed evaluation order (build with -ffp-contract=off) so that a value
// computed in a kernel and the same value computed on the host carry identical bits.
//
// Reference semantics restated here (paths relative to the reference tree):
//   metric / steer / rotation ........ src/primitives.h:224-292
//   sample around a node ............. src/randGen.h:70-109 (+ libstdc++ distributions)
//   local-planner sample positions ... src/problemStruct.h:154-168
//   triangle contact ................. RAPID 2.01 leaf test (library not in the tree)
#pragma once
#include <stdint.h>
#include "sff_pmath.h"

namespace sffg {

#define SFFG_PI 3.14159265358979323846
#define SFFG_PI_2 1.57079632679489661923
#define SFFG_TOL 1e-9  // TOLERANCE, src/primitives.h:45

SFF_HD double wrap_angle(double a) {  // NormalizeAngle: ONE correction only (primitives.h:278-286)
  if (a < -SFFG_PI) return a + 2 * SFFG_PI;
  if (a >= SFFG_PI) return a - 2 * SFFG_PI;
  return a;
}

// Point::distance(this=a, other=b): xyz differences a-b, angle terms wrap(b-a)
SFF_HD double dist6(const double* a, const double* b) {
  double sum = 0;
  for (int i = 0; i < 3; ++i) {
    double d = a[i] - b[i];
    sum += d * d;
  }
  for (int i = 3; i < 6; ++i) {
    double d = wrap_angle(b[i] - a[i]);
    sum += d * d;
  }
  return __builtin_sqrt(sum);
}

// Point::getStateInDistance(other, dist)
SFF_HD void steer(const double* from, const double* to, double dist, double* out) {
  double real = dist6(from, to);
  double s = dist / real;
  for (int i = 0; i < 3; ++i) out[i] = from[i] + (to[i] - from[i]) * s;
  for (int i = 3; i < 6; ++i) out[i] = from[i] + wrap_angle(to[i] - from[i]) * s;
}

// Point::FillRotationMatrix — R = Rz(yaw) Ry(pitch) Rx(roll), portable trig
SFF_HD void rotation(const double* p, double* R) {
  double cy = sffp::pcos(p[3]), sy = sffp::psin(p[3]);
  double cp = sffp::pcos(p[4]), sp = sffp::psin(p[4]);
  double cr = sffp::pcos(p[5]), sr = sffp::psin(p[5]);
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

// world = R v + T, evaluated ((R0 v0 + R1 v1) + R2 v2) + T
SFF_HD void xform(const double* R, const double* T, const double* v, double* w) {
  for (int i = 0; i < 3; ++i) w[i] = ((R[3 * i] * v[0] + R[3 * i + 1] * v[1]) + R[3 * i + 2] * v[2]) + T[i];
}

// libstdc++ generate_canonical<double,53> on one mt19937_64 word, then uniform_real(a,b)
SFF_HD double canonical(uint64_t w) {
  double r = (double)w / 18446744073709551616.0;
  if (r >= 1.0) r = 0.99999999999999988898;  // nextafter(1, 0)
  return r;
}
SFF_HD double uniform_real(uint64_t w, double a, double b) { return canonical(w) * (b - a) + a; }

SFF_HD bool in_limits(const double* p, const double* lim) {
  return p[0] >= lim[0] && p[0] <= lim[1] && p[1] >= lim[2] && p[1] <= lim[3] && p[2] >= lim[4] && p[2] <= lim[5];
}

// RandGen::randomPointInDistance from pre-drawn engine words (draw order: phi, theta, yaw,
// pitch-u, flip-u, roll; 2-D uses phi only).  Returns the limits test.
// The five transcendental values of a sample come in from outside: the kernels evaluate them with the portable trig
// (sample_point below), the libm parity mode with the C library's cos / sin / acos - on the host, per engine word,
// because that arithmetic is the reference's own (src/randGen.h:70-109) and exists nowhere else.
struct SampleTrig {
  double c_phi, s_phi;      // cos / sin of phi   = uniform_real(w[0], -pi, pi)
  double c_theta, s_theta;  // cos / sin of theta = uniform_real(w[1], -pi, pi)
  double acos_u;            // acos(1 - 2 uniform_real(w[3], 0, 1))
};
SFF_HD double sample_angle(uint64_t w) { return uniform_real(w, -SFFG_PI, SFFG_PI); }
SFF_HD double sample_acos_arg(uint64_t w) { return 1 - 2 * uniform_real(w, 0.0, 1.0); }
SFF_HD bool sample_point_with(const uint64_t* w, const double* center, double dist, int dim, const double* lim,
                              double* out, const SampleTrig& t) {
  if (dim == 2) {
    out[0] = center[0] + t.c_phi * dist;
    out[1] = center[1] + t.s_phi * dist;
    out[2] = 0; out[3] = 0; out[4] = 0; out[5] = 0;
  } else {
    double temp[6];
    double sphi = t.s_phi;
    temp[0] = center[0] + t.c_theta * sphi * dist;
    temp[1] = center[1] + t.s_theta * sphi * dist;
    temp[2] = center[2] + t.c_phi * dist;
    temp[3] = uniform_real(w[2], -SFFG_PI, SFFG_PI);
    double pitch = t.acos_u + SFFG_PI_2;
    if (uniform_real(w[4], 0.0, 1.0) < 0.5) {
      if (pitch < 0) pitch += SFFG_PI; else pitch -= SFFG_PI;
    }
    temp[4] = pitch;
    temp[5] = uniform_real(w[5], -SFFG_PI, SFFG_PI);
    steer(center, temp, dist, out);
  }
  return in_limits(out, lim);
}
SFF_HD bool sample_point(const uint64_t* w, const double* center, double dist, int dim, const double* lim,
                         double* out) {
  SampleTrig t;
  const double phi = sample_angle(w[0]);
  t.c_phi = sffp::pcos(phi);
  t.s_phi = sffp::psin(phi);
  t.c_theta = 0; t.s_theta = 0; t.acos_u = 0;
  if (dim != 2) {
    const double theta = sample_angle(w[1]);
    t.c_theta = sffp::pcos(theta);
    t.s_theta = sffp::psin(theta);
    t.acos_u = sffp::pacos(sample_acos_arg(w[3]));
  }
  return sample_point_with(w, center, dist, dim, lim, out, t);
}

// ---- local planner (Solver::isPathFree): parts = dist/0.1, samples index = 1 .. < parts
SFF_HD double edge_parts(const double* a, const double* b) { return dist6(a, b) / 0.1; }
SFF_HD int edge_samples(double parts) {
  if (!(parts > 1.0)) return 0;
  double c = (double)(long long)parts;  // trunc; parts > 0
  if (c < parts) c += 1.0;              // ceil
  return (int)c - 1;
}
SFF_HD void edge_sample_pos(const double* a, const double* dir, double parts, int index, double* p) {
  for (int i = 0; i < 3; ++i) p[i] = a[i] + (double)index * dir[i] / parts;
}

// ---- triangle contact: closed-interval box overlap AND the 17-axis separating-axis test
SFF_HD void cross(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
SFF_HD double dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// p1 is the origin after the shift, so its projection is exactly 0*ax0 + 0*ax1 + 0*ax2 = 0
// for finite axes; it is still evaluated through dot() to keep the bits of the general form.
SFF_HD bool axis_overlap(const double* ax, const double* p1, const double* p2, const double* p3, const double* q1,
                         const double* q2, const double* q3) {
  double P1 = dot(ax, p1), P2 = dot(ax, p2), P3 = dot(ax, p3);
  double Q1 = dot(ax, q1), Q2 = dot(ax, q2), Q3 = dot(ax, q3);
  double mx1 = P1 > P2 ? P1 : P2; if (P3 > mx1) mx1 = P3;
  double mn1 = P1 < P2 ? P1 : P2; if (P3 < mn1) mn1 = P3;
  double mx2 = Q1 > Q2 ? Q1 : Q2; if (Q3 > mx2) mx2 = Q3;
  double mn2 = Q1 < Q2 ? Q1 : Q2; if (Q3 < mn2) mn2 = Q3;
  if (mn1 > mx2) return false;
  if (mn2 > mx1) return false;
  return true;
}

SFF_HD bool sat17(const double* P, const double* Q) {
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
  cross(e1, e2, n1);
  cross(f1, f2, m1);
  if (!axis_overlap(n1, p1, p2, p3, q1, q2, q3)) return false;
  if (!axis_overlap(m1, p1, p2, p3, q1, q2, q3)) return false;
#define SFFG_AXIS(A, B)                                                   \
  cross(A, B, ax);                                                        \
  if (!axis_overlap(ax, p1, p2, p3, q1, q2, q3)) return false;
  SFFG_AXIS(e1, f1) SFFG_AXIS(e1, f2) SFFG_AXIS(e1, f3)
  SFFG_AXIS(e2, f1) SFFG_AXIS(e2, f2) SFFG_AXIS(e2, f3)
  SFFG_AXIS(e3, f1) SFFG_AXIS(e3, f2) SFFG_AXIS(e3, f3)
  SFFG_AXIS(e1, n1) SFFG_AXIS(e2, n1) SFFG_AXIS(e3, n1)
  SFFG_AXIS(f1, m1) SFFG_AXIS(f2, m1) SFFG_AXIS(f3, m1)
#undef SFFG_AXIS
  return true;
}

SFF_HD double min3(double a, double b, double c) { double m = a < b ? a : b; return c < m ? c : m; }
SFF_HD double max3(double a, double b, double c) { double m = a > b ? a : b; return c > m ? c : m; }

// box of an env triangle is precomputed (lo/hi); Q is the posed robot triangle
SFF_HD bool tri_box_overlap(const double* lo, const double* hi, const double* Q) {
  for (int a = 0; a < 3; ++a) {
    double qmin = min3(Q[a], Q[3 + a], Q[6 + a]), qmax = max3(Q[a], Q[3 + a], Q[6 + a]);
    if (lo[a] > qmax || qmin > hi[a]) return false;
  }
  return true;
}

}  // namespace sffg
