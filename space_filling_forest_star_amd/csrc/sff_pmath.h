// sff_pmath.h — portable, bit-reproducible double-precision sin / cos / acos.
//
// The reference draws its samples and builds its rotation matrices with glibc's
// cos/sin/acos (reference src/randGen.h:77-102, src/primitives.h:252-262).  A GPU
// libm differs from glibc in the last ulp, and one flipped threshold compare would
// diverge a whole planner run (SURVEY.md §7 "Bit-reproducible sampling").  These
// routines are therefore written once, with only IEEE-754 +,-,*,/ and sqrt, and are
// compiled with -ffp-contract=off for BOTH the gfx950 kernels and the host, so the
// HIP path and the CPU checker produce identical bits.  Algorithms: Cody–Waite
// three-part pi/2 reduction and the classic (Sun fdlibm-lineage) minimax kernels;
// accuracy < 1 ulp, verified against glibc in tests/test_pmath.py.
#pragma once
#include <string.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define SFF_HD __host__ __device__ inline
#else
#define SFF_HD inline
#endif

namespace sffp {

SFF_HD double rint_even(double v) {
  // round-to-nearest-even for |v| < 2^51 without touching the FP environment
  const double big = 6755399441055744.0;  // 1.5 * 2^52
  return (v + big) - big;
}

SFF_HD double kernel_sin(double x, double y, int have_tail) {
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  double z = x * x;
  double w = z * z;
  double r = (S2 + z * (S3 + z * S4)) + (z * w) * (S5 + z * S6);
  double v = z * x;
  if (!have_tail) return x + v * (S1 + z * r);
  return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

SFF_HD double kernel_cos(double x, double y) {
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double z = x * x;
  double w = z * z;
  double r = z * (C1 + z * (C2 + z * C3)) + (w * w) * (C4 + z * (C5 + z * C6));
  double hz = 0.5 * z;
  w = 1.0 - hz;
  return w + (((1.0 - w) - hz) + (z * r - x * y));
}

// x = k*pi/2 + (y0 + y1), |y0| <= pi/4 (+ a hair).  Valid for |x| < ~1e6.
SFF_HD int rem_pio2(double x, double* y0, double* y1) {
  const double INVPIO2 = 6.36619772367581382433e-01;   // 2/pi
  const double P1 = 1.57079632673412561417e+00;        // first 33 bits of pi/2
  const double P2 = 6.07710050630396597660e-11;        // next 33 bits
  const double P3 = 2.02226624879595063154e-21;        // remainder
  double k = rint_even(x * INVPIO2);
  double t = x - k * P1;      // k*P1 exact (33+20 bits)
  double w2 = k * P2;         // exact
  double r = t - w2;
  // exact rounding error of r = t - w2 (Knuth TwoSum on t + (-w2))
  double bb = r - t;
  double e = (t - (r - bb)) + ((-w2) - bb);
  double tail = e - k * P3;
  double s = r + tail;
  *y1 = (r - s) + tail;
  *y0 = s;
  return (int)k;
}

SFF_HD double psin(double x) {
  double ax = x < 0 ? -x : x;
  if (ax <= 0.78539816339744827900) return kernel_sin(x, 0.0, 0);
  double y0, y1;
  int n = rem_pio2(x, &y0, &y1) & 3;
  if (n == 0) return kernel_sin(y0, y1, 1);
  if (n == 1) return kernel_cos(y0, y1);
  if (n == 2) return -kernel_sin(y0, y1, 1);
  return -kernel_cos(y0, y1);
}

SFF_HD double pcos(double x) {
  double ax = x < 0 ? -x : x;
  if (ax <= 0.78539816339744827900) return kernel_cos(x, 0.0);
  double y0, y1;
  int n = rem_pio2(x, &y0, &y1) & 3;
  if (n == 0) return kernel_cos(y0, y1);
  if (n == 1) return -kernel_sin(y0, y1, 1);
  if (n == 2) return -kernel_cos(y0, y1);
  return kernel_sin(y0, y1, 1);
}

SFF_HD double clear_low32(double s) {
  unsigned long long u;
  memcpy(&u, &s, 8);
  u &= 0xFFFFFFFF00000000ULL;
  memcpy(&s, &u, 8);
  return s;
}

SFF_HD double acos_R(double z) {
  const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01,
               pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
               pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
               qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
               qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
  double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
  double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
  return p / q;
}

// acos for x in [-1, 1] (callers pass 1-2u with u in [0,1)).
SFF_HD double pacos(double x) {
  const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17;
  const double pi = 3.14159265358979311600e+00;
  double ax = x < 0 ? -x : x;
  if (ax >= 1.0) {
    if (x >= 1.0) return 0.0;
    return pi + 2.0 * pio2_lo;
  }
  if (ax < 0.5) {
    if (ax < 6.938893903907228e-18) return pio2_hi + pio2_lo;  // 2^-57
    double z = x * x;
    double r = acos_R(z);
    return pio2_hi - (x - (pio2_lo - x * r));
  }
  if (x < 0) {
    double z = (1.0 + x) * 0.5;
    double s = __builtin_sqrt(z);
    double w = acos_R(z) * s - pio2_lo;
    return pi - 2.0 * (s + w);
  }
  double z = (1.0 - x) * 0.5;
  double s = __builtin_sqrt(z);
  double df = clear_low32(s);
  double c = (z - df * df) / (s + df);
  double w = acos_R(z) * s + c;
  return 2.0 * (df + w);
}

}  // namespace sffp
