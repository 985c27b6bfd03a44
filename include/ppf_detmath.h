/*
 * ppf_detmath.h — deterministic fp64 elementary functions for the PPF engine.
 *
 * WHY THIS EXISTS
 *   The reference's PPF path (OpenCV-contrib surface_matching, called from
 *   /root/reference/include/CloudProcessing.h:442,495) quantises acos()/atan2()
 *   results with (int) casts and builds rotations from sin()/cos().  Vote counts
 *   are therefore a function of the last bit of those libm calls.  glibc (host)
 *   and OCML (gfx950) disagree in the last ulp, so a GPU engine that has to be
 *   *bit-exact* in its vote counts against a CPU oracle cannot call either.
 *   This header is the single numeric spec both sides evaluate: every function
 *   below uses only IEEE-754 fp64 + - * / and sqrt (all correctly rounded on
 *   x86-64 and on gfx950), integer bit tests, and no fused multiply-add.
 *
 *   Build rule: every translation unit including this file MUST be compiled
 *   with -ffp-contract=off (hipcc and g++ alike).  The pragmas below are a
 *   second line of defence.
 *
 *   Algorithms: classic argument-reduction + minimax-polynomial forms (the
 *   coefficient sets are the public-domain Sun fdlibm ones); accuracy is
 *   checked against glibc in tests/test_detmath.py (<= 1 ulp on the tested
 *   ranges, 2 ulp worst case near zeros of sin/cos).
 *
 *   This is a product header (part of the boundary).  The oracle may include
 *   it; nothing here includes or calls anything under oracle/.
 */
#ifndef PPF_DETMATH_H
#define PPF_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define PPF_HD __host__ __device__ __forceinline__
#else
#define PPF_HD static inline
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#elif defined(__GNUC__)
#pragma GCC optimize("fp-contract=off")
#endif

#define PPF_PI 3.14159265358979311600e+00 /* == M_PI, 0x400921FB54442D18 */

PPF_HD uint64_t ppf_d2bits(double x) {
  uint64_t u;
  __builtin_memcpy(&u, &x, 8);
  return u;
}
PPF_HD double ppf_bits2d(uint64_t u) {
  double x;
  __builtin_memcpy(&x, &u, 8);
  return x;
}
PPF_HD int32_t ppf_hi(double x) { return (int32_t)(ppf_d2bits(x) >> 32); }
PPF_HD uint32_t ppf_lo(double x) { return (uint32_t)ppf_d2bits(x); }
PPF_HD double ppf_fabs(double x) { return ppf_bits2d(ppf_d2bits(x) & 0x7fffffffffffffffULL); }
PPF_HD int ppf_isnan(double x) { return x != x; }
PPF_HD double ppf_sqrt(double x) { return __builtin_sqrt(x); }

/* (int)double with the x86-64 cvttsd2si result for NaN / out-of-range inputs
 * (0x80000000), which is what the reference's build computes for
 * (int)(acos(1.0000001)/step).  Spelled out so host and device agree. */
PPF_HD int32_t ppf_d2i(double x) {
  if (!(x > -2147483649.0 && x < 2147483648.0)) return (int32_t)0x80000000;
  return (int32_t)x;
}
PPF_HD int32_t ppf_f2i(float x) {
  if (!(x >= -2147483648.0f && x < 2147483648.0f)) return (int32_t)0x80000000;
  return (int32_t)x;
}

/* ------------------------------------------------------------------ acos */
PPF_HD double ppf_acos_rational(double z) {
  const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01,
               pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
               pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
               qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
               qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
  double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
  double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
  return p / q;
}

PPF_HD double ppf_acos(double x) {
  const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17;
  int32_t hx = ppf_hi(x);
  int32_t ix = hx & 0x7fffffff;
  if (ix >= 0x3ff00000) { /* |x| >= 1 or NaN */
    if (((uint32_t)(ix - 0x3ff00000) | ppf_lo(x)) == 0) return (hx > 0) ? 0.0 : PPF_PI + 2.0 * pio2_lo;
    return ppf_bits2d(0x7ff8000000000000ULL); /* NaN */
  }
  if (ix < 0x3fe00000) { /* |x| < 0.5 */
    if (ix <= 0x3c600000) return pio2_hi + pio2_lo;
    double r = ppf_acos_rational(x * x);
    return pio2_hi - (x - (pio2_lo - r * x));
  }
  if (hx < 0) { /* x <= -0.5 */
    double z = (1.0 + x) * 0.5;
    double r = ppf_acos_rational(z);
    double s = ppf_sqrt(z);
    double w = r * s - pio2_lo;
    return PPF_PI - 2.0 * (s + w);
  }
  /* x >= 0.5 */
  double z = (1.0 - x) * 0.5;
  double s = ppf_sqrt(z);
  double df = ppf_bits2d(ppf_d2bits(s) & 0xffffffff00000000ULL);
  double c = (z - df * df) / (s + df);
  double r = ppf_acos_rational(z);
  double w = r * s + c;
  return 2.0 * (df + w);
}

/* ------------------------------------------------------------------ atan / atan2 */
PPF_HD double ppf_atan(double x) {
  const double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01,
               aT2 = 1.42857142725034663711e-01, aT3 = -1.11111104054623557880e-01,
               aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02,
               aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02,
               aT8 = 4.97687799461593236017e-02, aT9 = -3.65315727442169155270e-02,
               aT10 = 1.62858201153657823623e-02;
  int32_t hx = ppf_hi(x);
  int32_t ix = hx & 0x7fffffff;
  if (ix >= 0x44100000) { /* |x| >= 2^66, inf or NaN */
    if (ppf_isnan(x)) return x + x;
    double v = 1.57079632679489655800e+00 + 6.12323399573676603587e-17;
    return (hx > 0) ? v : -v;
  }
  if (ix < 0x3e200000) return x; /* |x| < 2^-29 */
  /* The argument reduction of the five ranges, written without branches: every range's numerator and denominator are the
   * expressions fdlibm evaluates in that range, chosen by selects, followed by ONE division (|x| < 0.4375: x / 1.0 == x).
   * Same operations on the same operands as the branching form, hence the same bits; on a GPU a wave whose lanes fall
   * into different ranges no longer walks through four divisions one after the other. */
  const double ax = ppf_fabs(x);
  const int r0 = ix < 0x3fdc0000, r1 = ix < 0x3fe60000, r2 = ix < 0x3ff30000, r3 = ix < 0x40038000;
  const double num = r0 ? x : r1 ? (2.0 * ax - 1.0) : r2 ? (ax - 1.0) : r3 ? (ax - 1.5) : -1.0;
  const double den = r0 ? 1.0 : r1 ? (2.0 + ax) : r2 ? (ax + 1.0) : r3 ? (1.0 + 1.5 * ax) : ax;
  const double hi = r0 ? 0.0 : r1 ? 4.63647609000806093515e-01 : r2 ? 7.85398163397448278999e-01
                    : r3 ? 9.82793723247329054082e-01 : 1.57079632679489655800e+00;
  const double lo = r0 ? 0.0 : r1 ? 2.26987774529616870924e-17 : r2 ? 3.06161699786838301793e-17
                    : r3 ? 1.39033110312309984516e-17 : 6.12323399573676603587e-17;
  const double t = num / den;
  double z = t * t;
  double w = z * z;
  double s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
  double s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
  if (r0) return t - t * (s1 + s2);
  z = hi - ((t * (s1 + s2) - lo) - t);
  return (hx < 0) ? -z : z;
}

/* atan2 for finite or NaN arguments (infinities are not produced by the PPF path;
 * they fall through the generic division and still give a defined result). */
PPF_HD double ppf_atan2(double y, double x) {
  const double pi_lo = 1.2246467991473531772E-16;
  const double pi_o_2 = 1.5707963267948965580E+00;
  const double tiny = 1.0e-300;
  if (ppf_isnan(x) || ppf_isnan(y)) return x + y;
  int32_t hx = ppf_hi(x), hy = ppf_hi(y);
  uint32_t lx = ppf_lo(x), ly = ppf_lo(y);
  int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
  if ((((uint32_t)(hx - 0x3ff00000)) | lx) == 0) return ppf_atan(y); /* x == 1.0 */
  int m = ((hy >> 31) & 1) | ((hx >> 30) & 2); /* 2*sign(x) + sign(y) */
  if (((uint32_t)iy | ly) == 0) { /* y == 0 */
    switch (m) {
      case 0:
      case 1: return y;
      case 2: return PPF_PI + tiny;
      default: return -PPF_PI - tiny;
    }
  }
  if (((uint32_t)ix | lx) == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny; /* x == 0 */
  int32_t k = (iy - ix) >> 20;
  double z;
  if (k > 60) z = pi_o_2 + 0.5 * pi_lo;
  else if (hx < 0 && k < -60) z = 0.0;
  else z = ppf_atan(ppf_fabs(y / x));
  switch (m) {
    case 0: return z;
    case 1: return -z;
    case 2: return PPF_PI - (z - pi_lo);
    default: return (z - pi_lo) - PPF_PI;
  }
}

/* ------------------------------------------------------------------ sin / cos */
PPF_HD double ppf_ksin(double x, double y, int iy) {
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  double z = x * x;
  double v = z * x;
  double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  if (iy == 0) return x + v * (S1 + z * r);
  return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}
PPF_HD double ppf_kcos(double x, double y) {
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

/* Reduce x (|x| <= ~1e5, far more than the PPF path's |x| <= 2*pi) to y0+y1 in
 * [-pi/4, pi/4]; returns the quadrant.  Two-stage Cody-Waite, always both stages. */
PPF_HD int ppf_rem_pio2(double x, double* y0, double* y1) {
  const double invpio2 = 6.36619772367581382433e-01;
  const double pio2_1 = 1.57079632673412561417e+00;  /* first 33 bits of pi/2 */
  const double pio2_2 = 6.07710050630396597660e-11;  /* next 33 bits */
  const double pio2_2t = 2.02226624879595063154e-21; /* tail */
  if (ppf_fabs(x) <= 7.85398163397448278999e-01) {
    *y0 = x; *y1 = 0.0;
    return 0;
  }
  double t = x * invpio2;
  int32_t n = (int32_t)(t + (t >= 0.0 ? 0.5 : -0.5));
  double fn = (double)n;
  double r1 = x - fn * pio2_1;
  double w1 = fn * pio2_2;
  double r2 = r1 - w1;
  double w = fn * pio2_2t - ((r1 - r2) - w1);
  *y0 = r2 - w;
  *y1 = (r2 - *y0) - w;
  return n;
}
PPF_HD double ppf_sin(double x) {
  if (ppf_isnan(x)) return x;
  double y0, y1;
  int n = ppf_rem_pio2(x, &y0, &y1);
  switch (n & 3) {
    case 0: return ppf_ksin(y0, y1, 1);
    case 1: return ppf_kcos(y0, y1);
    case 2: return -ppf_ksin(y0, y1, 1);
    default: return -ppf_kcos(y0, y1);
  }
}
PPF_HD double ppf_cos(double x) {
  if (ppf_isnan(x)) return x;
  double y0, y1;
  int n = ppf_rem_pio2(x, &y0, &y1);
  switch (n & 3) {
    case 0: return ppf_kcos(y0, y1);
    case 1: return -ppf_ksin(y0, y1, 1);
    case 2: return -ppf_kcos(y0, y1);
    default: return ppf_ksin(y0, y1, 1);
  }
}

#endif /* PPF_DETMATH_H */
