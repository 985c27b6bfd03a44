/*
 * ppf_core.h — pair feature, key hash, reference-frame transform and pose algebra of the PPF
 * path, written once for host and gfx950 device code (every function is PPF_HD).
 *
 * What is computed is fixed by the library the reference calls,
 * cv::ppf_match_3d::PPF3DDetector (/root/reference/include/CloudProcessing.h:236,442,495;
 * SURVEY.md §8a rows A3, A4, A5, A8).  Elementary functions come from include/ppf_detmath.h so the
 * integer vote counts are identical on every machine; compile with -ffp-contract=off.
 *
 * This is product code: it does not include or call anything under oracle/.
 */
#ifndef PPF_CORE_H
#define PPF_CORE_H

#include "../../include/ppf_detmath.h"

#define PPF_EPS 1.192092896e-07 /* FLT_EPSILON, the library's EPS */
#define PPF_HASH_SEED 42u

struct ppf_vec3 {
  double x, y, z;
};

PPF_HD ppf_vec3 ppf_mk3(double x, double y, double z) {
  ppf_vec3 v;
  v.x = x; v.y = y; v.z = z;
  return v;
}
PPF_HD double ppf_dot3(const ppf_vec3& a, const ppf_vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

/* R (row-major 3x3) * p */
PPF_HD ppf_vec3 ppf_mul33(const double* R, const ppf_vec3& p) {
  return ppf_mk3(R[0] * p.x + R[1] * p.y + R[2] * p.z, R[3] * p.x + R[4] * p.y + R[5] * p.z,
                 R[6] * p.x + R[7] * p.y + R[8] * p.z);
}

/* ---- MurmurHash3_x64_128 of one 16-byte key, seed 42, low 32 bits of h1 (row A3) ---------- */
PPF_HD uint64_t ppf_rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
PPF_HD uint64_t ppf_fmix64(uint64_t k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL;
  k ^= k >> 33;
  return k;
}
PPF_HD uint32_t ppf_murmur_key16(int32_t k0, int32_t k1i, int32_t k2i, int32_t k3) {
  const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
  uint64_t h1 = PPF_HASH_SEED, h2 = PPF_HASH_SEED;
  uint64_t k1 = (uint64_t)(uint32_t)k0 | ((uint64_t)(uint32_t)k1i << 32);
  uint64_t k2 = (uint64_t)(uint32_t)k2i | ((uint64_t)(uint32_t)k3 << 32);
  k1 *= c1; k1 = ppf_rotl64(k1, 31); k1 *= c2; h1 ^= k1;
  h1 = ppf_rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
  k2 *= c2; k2 = ppf_rotl64(k2, 33); k2 *= c1; h2 ^= k2;
  h2 = ppf_rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
  h1 ^= 16; h2 ^= 16;
  h1 += h2; h2 += h1;
  h1 = ppf_fmix64(h1); h2 = ppf_fmix64(h2);
  h1 += h2;
  return (uint32_t)h1;
}

/* ---- pair feature (row A3): f = {acos(n1.d), acos(n2.d), acos(n1.n2), |d|}; f must come in
 * zeroed and is left untouched apart from f[3] when |d| <= EPS. */
PPF_HD void ppf_pair_feature(const ppf_vec3& p1, const ppf_vec3& n1, const ppf_vec3& p2, const ppf_vec3& n2,
                             double* f) {
  ppf_vec3 d = ppf_mk3(p2.x - p1.x, p2.y - p1.y, p2.z - p1.z);
  f[3] = ppf_sqrt(d.x * d.x + d.y * d.y + d.z * d.z);
  if (f[3] <= PPF_EPS) return;
  double s = 1.0 / f[3];
  d.x *= s; d.y *= s; d.z *= s;
  f[0] = ppf_acos(ppf_dot3(n1, d));
  f[1] = ppf_acos(ppf_dot3(n2, d));
  f[2] = ppf_acos(ppf_dot3(n1, n2));
}

PPF_HD uint32_t ppf_hash_feature(const double* f, double angle_step, double dist_step) {
  return ppf_murmur_key16(ppf_d2i(f[0] / angle_step), ppf_d2i(f[1] / angle_step), ppf_d2i(f[2] / angle_step),
                          ppf_d2i(f[3] / dist_step));
}

/* ---- PCL's pair feature (policy switch feature = PPF_FEATURE_DARBOUX; pcl::computePairFeatures as PPFEstimation
 * uses it): with d = p2 - p1, a1 = n1.d/|d|, a2 = n2.d/|d|, the point whose normal is closer to the line takes the
 * role of the source (u = its normal; PCL tests acos|a1| > acos|a2|, i.e. |a1| < |a2|), v = d x u / |d x u|, w = u x v:
 *   f[0] = atan2(w.n, u.n)   f[1] = v.n   f[2] = a1 or -a2   f[3] = |d|      (n = the other point's normal)
 * Returns 0 for degenerate pairs (|d| = 0 or d parallel to u), which PCL leaves out.  Keys are floor(f / step) --
 * PCL divides the two cosines by the ANGLE step as well. */
PPF_HD int ppf_pair_feature_darboux(const ppf_vec3& p1, const ppf_vec3& n1, const ppf_vec3& p2, const ppf_vec3& n2, double* f) {
  ppf_vec3 d = ppf_mk3(p2.x - p1.x, p2.y - p1.y, p2.z - p1.z);
  const double f4 = ppf_sqrt(d.x * d.x + d.y * d.y + d.z * d.z);
  if (!(f4 > 0.0)) return 0;
  const double a1 = ppf_dot3(n1, d) / f4, a2 = ppf_dot3(n2, d) / f4;
  ppf_vec3 u = n1, n = n2;
  double f3 = a1;
  if (ppf_fabs(a1) < ppf_fabs(a2)) {
    u = n2; n = n1;
    d = ppf_mk3(-d.x, -d.y, -d.z);
    f3 = -a2;
  }
  ppf_vec3 v = ppf_mk3(d.y * u.z - d.z * u.y, d.z * u.x - d.x * u.z, d.x * u.y - d.y * u.x);
  const double vn = ppf_sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
  if (!(vn > 0.0)) return 0;
  v = ppf_mk3(v.x / vn, v.y / vn, v.z / vn);
  const ppf_vec3 w = ppf_mk3(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x);
  f[0] = ppf_atan2(ppf_dot3(w, n), ppf_dot3(u, n));
  f[1] = ppf_dot3(v, n);
  f[2] = f3;
  f[3] = f4;
  return 1;
}
PPF_HD int32_t ppf_floor_key(double x) { return ppf_d2i(__builtin_floor(x)); }

/* the key table's shape: keys (k0 + o0, k1 + o1, k2 + o2, k3) inside n0 x n1 x n2 x nd are tabulated.  Offsets are zero
 * for the reference's feature (three acos bins, never negative); PCL's feature has signed keys. */
struct KeyDims {
  int n0, n1, n2, nd, o0, o1, o2;
};
PPF_HD bool key_index(const KeyDims& d, const int32_t k0, const int32_t k1, const int32_t k2, const int32_t k3, size_t* idx) {
  const uint32_t a = (uint32_t)(k0 + d.o0), b = (uint32_t)(k1 + d.o1), c = (uint32_t)(k2 + d.o2);
  if (!((a < (uint32_t)d.n0) & (b < (uint32_t)d.n1) & (c < (uint32_t)d.n2) & ((uint32_t)k3 < (uint32_t)d.nd))) return false;
  *idx = (size_t)(((a * (uint32_t)d.n1 + b) * (uint32_t)d.n2 + c) * (uint32_t)d.nd + (uint32_t)k3); /* the table holds at most 2^26 keys: 32-bit arithmetic */
  return true;
}
PPF_HD size_t key_table_size(const KeyDims& d) { return (size_t)d.n0 * d.n1 * d.n2 * d.nd; }
/* PPF_KEY_EXACT with the reference's feature: an acos of a dot product a rounding above 1 (parallel normals on a flat face) is
 * NaN and its bin INT_MIN (ppf_d2i).  The reference's table hashes such a key like any other, and so does the hash path
 * here; under exact keys they are keys like any other too -- a model pair and a scene pair with NaN in the same places and
 * equal bins elsewhere match -- and take the last bin of their dimension, which no angle reaches (bins run to
 * floor(pi / angle_step), the dimension one further).  Off the fast path: only keys key_index() turned down come here. */
PPF_HD bool key_index_nan(const KeyDims& d, int32_t k0, int32_t k1, int32_t k2, const int32_t k3, size_t* idx) {
  if ((d.o0 | d.o1 | d.o2) != 0) return false; /* the Darboux feature has no NaN keys (degenerate pairs are left out) */
  const int32_t nan_bin = (int32_t)0x80000000;
  if (k0 == nan_bin) k0 = d.n0 - 1;
  if (k1 == nan_bin) k1 = d.n1 - 1;
  if (k2 == nan_bin) k2 = d.n2 - 1;
  return key_index(d, k0, k1, k2, k3, idx);
}

/* A pair record names a model row by a code: bits 0..17 the byte offset of the row's bin 0 in the LDS accumulator (a multiple
 * of 4) with the half of the word its 16-bit cells live in as bit 0; bits 18..22 and 23..29 the entry's X and cell q of the
 * count-table path (agg_cell: a function of alpha_m and numAngles alone, so the table build evaluates it once instead
 * of k_vote once per entry and range of hits: 3 % of the kernel) */
constexpr uint32_t ROW_OFFSET_MASK = 0x3FFFCu, ROW_CODE_MASK = 0x3FFFFu;
constexpr int ROW_X_SHIFT = 18, ROW_Q_SHIFT = 23;
constexpr uint32_t ROW_Q_MASK = 127u; /* the cell field: bits 23..29 (0 .. AGG_Q, AGG_Q <= 64) */
#ifndef PPF_AGG_Q
#define PPF_AGG_Q 64
#endif
constexpr int AGG_Q = PPF_AGG_Q;      /* count-table cells per alpha bin (a multiple of 16, at most 64: one cell per lane in the table build) */
static_assert(AGG_Q % 16 == 0 && AGG_Q >= 16 && AGG_Q <= 64, "count-table cells");

/* ---- reference frame (row A4): R rotates n onto +x (Rodrigues about (0, n.z, -n.y)), t = -R p */
PPF_HD void ppf_transform_rt(const ppf_vec3& p, const ppf_vec3& n, double* R, double* t) {
  double angle = ppf_acos(n.x);
  double ax[3] = {0.0, n.z, -n.y};
  if (n.y == 0 && n.z == 0) {
    ax[1] = 1; ax[2] = 0;
  } else {
    double norm = ppf_sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
    if (norm > PPF_EPS) {
      double s = 1.0 / norm;
      ax[0] *= s; ax[1] *= s; ax[2] *= s;
    }
  }
  const double sinA = ppf_sin(angle), cosA = ppf_cos(angle), cos1A = 1.0 - cosA;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double v = (i == j) ? cosA : 0.0;
      if (i != j) v += (((i + 1) % 3 == j) ? -1.0 : 1.0) * sinA * ax[3 - i - j];
      v += cos1A * ax[i] * ax[j];
      R[i * 3 + j] = v;
    }
  ppf_vec3 rp = ppf_mul33(R, p);
  t[0] = -rp.x; t[1] = -rp.y; t[2] = -rp.z;
}

/* alpha of a point already moved into the reference frame (q = t + R p2).  Returns 0 where the
 * library skips the pair (NaN). */
PPF_HD int ppf_alpha_in_frame(double qy, double qz, double* alpha) {
  double a = ppf_atan2(-qz, qy);
  if (a != a) return 0;
  /* the library's sign fix-up: if (sin(a) * qz < 0) a = -a.  For |a| <= pi the sine has the sign of a (its true value is at
   * least 1.2e-16 away from zero wherever a != 0), so away from the underflow range the product's sign is that of a * qz
   * and no sine is needed; the tiny arguments keep the literal expression. */
  const double aa = ppf_fabs(a), aq = ppf_fabs(qz);
  const int flip = (aa > 1e-100 && aq > 1e-100) ? ((a < 0.0) != (qz < 0.0)) : (ppf_sin(a) * qz < 0.0);
  if (flip) a = -a;
  *alpha = -a;
  return 1;
}

/* Whether ppf_alpha_in_frame has an alpha at all, without computing it: ppf_atan2 returns NaN exactly when an argument is
 * NaN or both are infinite (every other path ends in a finite value), which finite clouds never produce. */
PPF_HD int ppf_alpha_exists(double qy, double qz) {
  const double big = 1.7976931348623157e308;
  return !(qy != qy) && !(qz != qz) && !(ppf_fabs(qy) > big && ppf_fabs(qz) > big);
}

/* training-side alpha (computeAlpha): NaN -> 0 */
PPF_HD double ppf_model_alpha(const double* R, const double* t, const ppf_vec3& p2) {
  ppf_vec3 rp = ppf_mul33(R, p2);
  double a;
  if (!ppf_alpha_in_frame(t[1] + rp.y, t[2] + rp.z, &a)) return 0.0;
  return a;
}

/* alpha bin of one vote, the exact chain: (int)(numAngles*(alpha_m - alpha_s + 2pi)/(4pi)) */
PPF_HD int32_t ppf_alpha_bin_exact(float alpha_m, double alpha_s, int num_angles) {
  double alpha = (double)alpha_m - alpha_s;
  return ppf_d2i(num_angles * (alpha + 2 * PPF_PI) / (4 * PPF_PI));
}

/* PCL's binning (policy switch alpha_range_2pi): the difference wrapped into [-pi, pi], numAngles bins of 2pi/numAngles */
PPF_HD int32_t ppf_alpha_bin_exact_2pi(float alpha_m, double alpha_s, int num_angles) {
  double alpha = (double)alpha_m - alpha_s;
  if (alpha < -PPF_PI) alpha += 2 * PPF_PI;
  else if (alpha > PPF_PI) alpha -= 2 * PPF_PI;
  const int32_t k = ppf_d2i(num_angles * (alpha + PPF_PI) / (2 * PPF_PI));
  return k >= num_angles ? num_angles - 1 : k; /* alpha == +pi exactly */
}

/* ---- pose algebra (row A8) ------------------------------------------------------------------ */
PPF_HD void ppf_mat44_mul(const double* A, const double* B, double* C) {
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      double s = 0;
      for (int k = 0; k < 4; k++) s += A[i * 4 + k] * B[k * 4 + j];
      C[i * 4 + j] = s;
    }
}
PPF_HD void ppf_rt_to_pose(const double* R, const double* t, double* P) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) P[i * 4 + j] = R[i * 3 + j];
  P[3] = t[0]; P[7] = t[1]; P[11] = t[2];
  P[12] = 0; P[13] = 0; P[14] = 0; P[15] = 1;
}
PPF_HD void ppf_dcm_to_quat(const double* R, double* q) {
  double n4;
  const double tr = R[0] + R[4] + R[8];
  if (tr > 0.0) {
    q[1] = R[5] - R[7]; q[2] = R[6] - R[2]; q[3] = R[1] - R[3]; q[0] = tr + 1.0; n4 = q[0];
  } else if ((R[0] > R[4]) && (R[0] > R[8])) {
    q[1] = 1.0 + R[0] - R[4] - R[8]; q[2] = R[3] + R[1]; q[3] = R[6] + R[2]; q[0] = R[5] - R[7]; n4 = q[1];
  } else if (R[4] > R[8]) {
    q[1] = R[3] + R[1]; q[2] = 1.0 + R[4] - R[0] - R[8]; q[3] = R[7] + R[5]; q[0] = R[6] - R[2]; n4 = q[2];
  } else {
    q[1] = R[6] + R[2]; q[2] = R[7] + R[5]; q[3] = 1.0 + R[8] - R[0] - R[4]; q[0] = R[1] - R[3]; n4 = q[3];
  }
  const double factor = 0.5 / ppf_sqrt(n4);
  q[0] *= factor; q[1] *= factor; q[2] *= factor; q[3] *= factor;
}
PPF_HD void ppf_quat_to_dcm(const double* q, double* R) {
  double sqw = q[0] * q[0], sqx = q[1] * q[1], sqy = q[2] * q[2], sqz = q[3] * q[3];
  double tmp1, tmp2;
  R[0] = sqx - sqy - sqz + sqw;
  R[4] = -sqx + sqy - sqz + sqw;
  R[8] = -sqx - sqy + sqz + sqw;
  tmp1 = q[1] * q[2]; tmp2 = q[3] * q[0];
  R[1] = 2.0 * (tmp1 + tmp2); R[3] = 2.0 * (tmp1 - tmp2);
  tmp1 = q[1] * q[3]; tmp2 = q[2] * q[0];
  R[2] = 2.0 * (tmp1 - tmp2); R[6] = 2.0 * (tmp1 + tmp2);
  tmp1 = q[2] * q[3]; tmp2 = q[1] * q[0];
  R[5] = 2.0 * (tmp1 + tmp2); R[7] = 2.0 * (tmp1 - tmp2);
}
PPF_HD double ppf_angle_from_trace(double trace) {
  if (ppf_fabs(trace - 3) <= PPF_EPS) return 0;
  if (ppf_fabs(trace + 1) <= PPF_EPS) return PPF_PI;
  return ppf_acos((trace - 1) / 2);
}

#endif /* PPF_CORE_H */
