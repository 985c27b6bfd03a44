/*
 * ppf_oracle.cpp — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product (yolo_ppf_pose_estimation_amd/, include/) never
 * imports, links or calls anything in oracle/.
 *
 * WHAT IT RESTATES
 *   The PPF path the reference calls at
 *     /root/reference/include/CloudProcessing.h:205,217,234  PPF3DDetector(relSampling, relDistance)
 *     /root/reference/include/CloudProcessing.h:236          detector.trainModel(Mat N x 6 f32)
 *     /root/reference/include/CloudProcessing.h:442          detector.match(scene, results, step, dist)
 *     /root/reference/include/CloudProcessing.h:495          detector.match_S2B(scene, edge, results, step, dist)
 *   The arithmetic behind those calls is NOT in /root/reference: it lives in the
 *   un-vendored third-party library OpenCV-contrib `surface_matching` (module
 *   opencv_contrib/modules/surface_matching; the reference only constrains
 *   "OpenCV >= 4.0", README.md:5, no pinned version, and needs a privately patched
 *   build for match_S2B/read/write).  This file restates that library's published
 *   algorithm (Drost et al. 2010 as implemented in ppf_match_3d.cpp, ppf_helpers.cpp,
 *   pose_3d.cpp, c_utils.hpp, t_hash_int.cpp, hash_murmur64.hpp) from knowledge of the
 *   public source.  Each function names the upstream routine it follows.
 *
 * PARITY STATUS: **parity unpinned.**  The reference ships no tests, no golden
 *   vectors and no recorded outputs for this path, and neither
 *   OpenCV nor PCL exists in the build container, so this restatement could not be
 *   checked against a reference binary.  What pins it instead: public MurmurHash3
 *   x64_128 vectors, invariance / identity known-answer tests, self-match recovery,
 *   and regression constants measured from the reference's bottle PLY
 *   (tests/test_oracle_*.py, tests/golden/).
 *
 * NUMERIC MODES
 *   mode 0 "det"  : elementary functions from include/ppf_detmath.h — the frozen spec the
 *                   GPU engine must match bit-for-bit.
 *   mode 1 "libm" : glibc acos/atan2/sin/cos — what an upstream build would call.  Tests
 *                   report how often the two modes disagree (last-ulp bin flips).
 *
 * Build: oracle/Makefile (g++ -O2 -ffp-contract=off -fopenmp).
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/ppf_detmath.h"

namespace {

const double EPS = 1.192092896e-07; /* upstream c_utils.hpp: EPS = FLT_EPSILON as double */

/* ---- elementary functions, mode-switched --------------------------------------- */
struct MathDet {
  static double acos_(double x) { return ppf_acos(x); }
  static double atan2_(double y, double x) { return ppf_atan2(y, x); }
  static double sin_(double x) { return ppf_sin(x); }
  static double cos_(double x) { return ppf_cos(x); }
};
struct MathLibm {
  static double acos_(double x) { return std::acos(x); }
  static double atan2_(double y, double x) { return std::atan2(y, x); }
  static double sin_(double x) { return std::sin(x); }
  static double cos_(double x) { return std::cos(x); }
};

struct V3 { double x, y, z; };
struct M33 { double m[3][3]; };

inline V3 v3(const float* p) { return V3{(double)p[0], (double)p[1], (double)p[2]}; }
inline double dot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 mulMV(const M33& R, const V3& p) {
  return V3{R.m[0][0] * p.x + R.m[0][1] * p.y + R.m[0][2] * p.z,
            R.m[1][0] * p.x + R.m[1][1] * p.y + R.m[1][2] * p.z,
            R.m[2][0] * p.x + R.m[2][1] * p.y + R.m[2][2] * p.z};
}

/* (int) of a double as the reference's x86-64 build evaluates it (cvttsd2si). */
inline int d2i(double x) { return ppf_d2i(x); }

/* ---- MurmurHash3 x64_128 (upstream hash_murmur64.hpp: hashMurmurx64) -------------- */
inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
inline uint64_t fmix64(uint64_t k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL;
  k ^= k >> 33;
  return k;
}
void murmur3_x64_128(const void* key, int len, uint32_t seed, uint64_t out[2]) {
  const uint8_t* data = (const uint8_t*)key;
  const int nblocks = len / 16;
  uint64_t h1 = seed, h2 = seed;
  const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
  for (int i = 0; i < nblocks; i++) {
    uint64_t k1, k2;
    memcpy(&k1, data + 16 * i, 8);
    memcpy(&k2, data + 16 * i + 8, 8);
    k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
    h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
    k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
    h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
  }
  const uint8_t* tail = data + nblocks * 16;
  uint64_t k1 = 0, k2 = 0;
  switch (len & 15) {
    case 15: k2 ^= ((uint64_t)tail[14]) << 48; /* fallthrough */
    case 14: k2 ^= ((uint64_t)tail[13]) << 40; /* fallthrough */
    case 13: k2 ^= ((uint64_t)tail[12]) << 32; /* fallthrough */
    case 12: k2 ^= ((uint64_t)tail[11]) << 24; /* fallthrough */
    case 11: k2 ^= ((uint64_t)tail[10]) << 16; /* fallthrough */
    case 10: k2 ^= ((uint64_t)tail[9]) << 8;   /* fallthrough */
    case 9:  k2 ^= ((uint64_t)tail[8]) << 0;
             k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2; /* fallthrough */
    case 8:  k1 ^= ((uint64_t)tail[7]) << 56; /* fallthrough */
    case 7:  k1 ^= ((uint64_t)tail[6]) << 48; /* fallthrough */
    case 6:  k1 ^= ((uint64_t)tail[5]) << 40; /* fallthrough */
    case 5:  k1 ^= ((uint64_t)tail[4]) << 32; /* fallthrough */
    case 4:  k1 ^= ((uint64_t)tail[3]) << 24; /* fallthrough */
    case 3:  k1 ^= ((uint64_t)tail[2]) << 16; /* fallthrough */
    case 2:  k1 ^= ((uint64_t)tail[1]) << 8;  /* fallthrough */
    case 1:  k1 ^= ((uint64_t)tail[0]) << 0;
             k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
  }
  h1 ^= (uint64_t)len; h2 ^= (uint64_t)len;
  h1 += h2; h2 += h1;
  h1 = fmix64(h1); h2 = fmix64(h2);
  h1 += h2; h2 += h1;
  out[0] = h1; out[1] = h2;
}

/* upstream hashPPF(): quantise the 4 features, murmur the 16-byte key with seed 42 and keep
 * KeyType (= unsigned int) hashKey[0], i.e. the low 32 bits of h1 on little-endian x86-64. */
uint32_t hashPPF(const double f[4], double angleStep, double distStep, int32_t keyOut[4] = nullptr) {
  int32_t key[4] = {d2i(f[0] / angleStep), d2i(f[1] / angleStep), d2i(f[2] / angleStep), d2i(f[3] / distStep)};
  if (keyOut) memcpy(keyOut, key, 16);
  uint64_t h[2];
  murmur3_x64_128(key, 16, 42, h);
  return (uint32_t)h[0];
}

/* upstream computePPFFeatures(): f left untouched (all-zero from the caller) when |d| <= EPS. */
template <class M>
void computePPFFeatures(const V3& p1, const V3& n1, const V3& p2, const V3& n2, double f[4]) {
  V3 d{p2.x - p1.x, p2.y - p1.y, p2.z - p1.z};
  f[3] = std::sqrt(d.x * d.x + d.y * d.y + d.z * d.z);
  if (f[3] <= EPS) return;
  double s = 1.0 / f[3];
  d.x *= s; d.y *= s; d.z *= s;
  f[0] = M::acos_(dot(n1, d)); /* TAngle3Normalized == acos(a.b) */
  f[1] = M::acos_(dot(n2, d));
  f[2] = M::acos_(dot(n1, n2));
}

/* PCL's pair feature (policy switch; pcl::computePairFeatures as PPFEstimation / PPFRegistration call it), in fp64 where
 * PCL computes in float: returns false for degenerate pairs, which PCL leaves out of table and vote alike. */
template <class M>
bool computePairFeaturesDarboux(const V3& p1, const V3& n1, const V3& p2, const V3& n2, double f[4]) {
  V3 d{p2.x - p1.x, p2.y - p1.y, p2.z - p1.z};
  const double f4 = std::sqrt(d.x * d.x + d.y * d.y + d.z * d.z);
  if (!(f4 > 0.0)) return false;
  const double a1 = dot(n1, d) / f4, a2 = dot(n2, d) / f4;
  V3 u = n1, n = n2;
  double f3 = a1;
  if (std::fabs(a1) < std::fabs(a2)) { /* PCL: acos(|a1|) > acos(|a2|): the other point becomes the source */
    u = n2; n = n1;
    d = V3{-d.x, -d.y, -d.z};
    f3 = -a2;
  }
  V3 v{d.y * u.z - d.z * u.y, d.z * u.x - d.x * u.z, d.x * u.y - d.y * u.x};
  const double vn = std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
  if (!(vn > 0.0)) return false;
  v = V3{v.x / vn, v.y / vn, v.z / vn};
  const V3 w{u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x};
  f[0] = M::atan2_(dot(w, n), dot(u, n));
  f[1] = dot(v, n);
  f[2] = f3;
  f[3] = f4;
  return true;
}
/* PPFHashMapSearch: keys are floor(f / step), the two cosines divided by the ANGLE step like the angle (PCL's own quirk) */
uint32_t hashDarboux(const double f[4], double angleStep, double distStep, int32_t keyOut[4]) {
  int32_t key[4] = {d2i(std::floor(f[0] / angleStep)), d2i(std::floor(f[1] / angleStep)), d2i(std::floor(f[2] / angleStep)),
                    d2i(std::floor(f[3] / distStep))};
  if (keyOut) memcpy(keyOut, key, 16);
  uint64_t h[2];
  murmur3_x64_128(key, 16, 42, h);
  return (uint32_t)h[0];
}

/* upstream aaToR(): Rodrigues rotation from axis/angle. */
template <class M>
void aaToR(const V3& axis, double angle, M33& R) {
  const double sinA = M::sin_(angle), cosA = M::cos_(angle), cos1A = 1.0 - cosA;
  const double ax[3] = {axis.x, axis.y, axis.z};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double v = (i == j) ? cosA : 0.0;
      if (i != j) v += (((i + 1) % 3 == j) ? -1.0 : 1.0) * sinA * ax[3 - i - j];
      v += cos1A * ax[i] * ax[j];
      R.m[i][j] = v;
    }
}

/* upstream computeTransformRT(): rotation taking n1 onto +x, translation taking p1 to the origin. */
template <class M>
void computeTransformRT(const V3& p1, const V3& n1, M33& R, V3& t) {
  double angle = M::acos_(n1.x);
  V3 axis{0.0, n1.z, -n1.y};
  if (n1.y == 0 && n1.z == 0) {
    axis.y = 1; axis.z = 0;
  } else {
    double norm = std::sqrt(axis.x * axis.x + axis.y * axis.y + axis.z * axis.z);
    if (norm > EPS) {
      double s = 1.0 / norm;
      axis.x *= s; axis.y *= s; axis.z *= s;
    }
  }
  aaToR<M>(axis, angle, R);
  V3 rp = mulMV(R, p1);
  t = V3{-rp.x, -rp.y, -rp.z};
}

/* The alpha sign dance shared by upstream computeAlpha() (training) and the inline block in
 * match().  Returns false where match() would `continue` (NaN); computeAlpha() returns 0 there. */
template <class M>
bool alphaFromTransformed(const V3& q, double& alpha) {
  alpha = M::atan2_(-q.z, q.y);
  if (alpha != alpha) return false;
  if (M::sin_(alpha) * q.z < 0.0) alpha = -alpha;
  alpha = -alpha;
  return true;
}
template <class M>
double computeAlpha(const V3& p1, const V3& n1, const V3& p2) {
  M33 R; V3 t;
  computeTransformRT<M>(p1, n1, R, t);
  V3 rp = mulMV(R, p2);
  V3 mpt{t.x + rp.x, t.y + rp.y, t.z + rp.z};
  double a;
  if (!alphaFromTransformed<M>(mpt, a)) return 0;
  return a;
}

/* ---- upstream ppf_helpers.cpp: computeBboxStd + samplePCByQuantization ---------------- */
void computeBbox(const float* pc, int n, int stride, float lo[3], float hi[3]) {
  for (int k = 0; k < 3; k++) { lo[k] = pc[k]; hi[k] = pc[k]; }
  for (int i = 0; i < n; i++)
    for (int k = 0; k < 3; k++) {
      float v = pc[(size_t)i * stride + k];
      if (v < lo[k]) lo[k] = v;
      if (v > hi[k]) hi[k] = v;
    }
}

/* Non-empty cells are emitted in ascending cell-index order; each output row is the mean
 * position and the re-normalised summed normal of the cell's points.  Cell index arithmetic
 * is float, exactly as upstream ((int)((float)n*(p-min)/range)).  A zero range (upstream
 * divides by zero there) is defined here as cell 0; a cell whose summed normal is <= EPS gets
 * a zero normal (upstream leaves the Mat row uninitialised). */
std::vector<float> samplePCByQuantization(const float* pc, int n, int stride, const float lo[3], const float hi[3],
                                          float sampleStep) {
  const int ns = (int)(1.0 / sampleStep);
  const float r[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
  const size_t cells = (size_t)(ns + 1) * (ns + 1) * (ns + 1);
  std::vector<std::vector<int>> map(cells);
  for (int i = 0; i < n; i++) {
    const float* p = pc + (size_t)i * stride;
    int c[3];
    for (int k = 0; k < 3; k++) c[k] = (r[k] > 0.0f) ? ppf_f2i((float)ns * (p[k] - lo[k]) / r[k]) : 0;
    const int index = c[0] * ns * ns + c[1] * ns + c[2];
    map[(size_t)index].push_back(i);
  }
  std::vector<float> out;
  for (size_t ci = 0; ci < cells; ci++) {
    const std::vector<int>& cell = map[ci];
    const int cn = (int)cell.size();
    if (!cn) continue;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int j = 0; j < cn; j++) {
      const float* p = pc + (size_t)cell[j] * stride;
      for (int k = 0; k < 6; k++) acc[k] += (double)p[k];
    }
    for (int k = 0; k < 6; k++) acc[k] /= (double)cn;
    float row[6] = {(float)acc[0], (float)acc[1], (float)acc[2], 0.f, 0.f, 0.f};
    double norm = std::sqrt(acc[3] * acc[3] + acc[4] * acc[4] + acc[5] * acc[5]);
    if (norm > EPS) {
      row[3] = (float)(acc[3] / norm); row[4] = (float)(acc[4] / norm); row[5] = (float)(acc[5] / norm);
    }
    out.insert(out.end(), row, row + 6);
  }
  return out;
}

/* ---- upstream t_hash_int.cpp: chained table, insert prepends, no key comparison ------- */
struct THash { uint32_t id; int i; int ppfInd; };
struct HashNode { uint32_t key; int data; /* index into hash_nodes */ int next; /* node index or -1 */ };

uint32_t nextPow2(uint32_t v) {
  v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; v++;
  return v;
}

struct Model {
  double sampling_step_relative, distance_step_relative, angle_step_relative, angle_step_radians;
  double angle_step, distance_step;
  double position_threshold, rotation_threshold;
  bool use_weighted_avg;
  int num_ref_points = 0;
  uint32_t slots = 0;
  std::vector<float> sampled_pc;  /* num_ref_points x 6 */
  std::vector<float> ppf;         /* N^2 x 5 */
  std::vector<THash> hash_nodes;  /* N^2, entry i*N+j (i==j unused) */
  std::vector<int> bucket_head;   /* slots, node index or -1 */
  std::vector<HashNode> list_nodes;
  bool trained = false;
  int mode = 0;
  /* PCL-semantics policy switches (closing paragraph; off = the OpenCV behaviour the reference uses):
   *   key_exact     PPFHashMapSearch::nearestNeighborSearch compares the quantised key, the OpenCV table walks a whole
   *                 hash bucket
   *   pair_radius   PPFRegistration pairs a reference point only with scene points within a radius (kd-tree search of
   *                 model_diameter / 2); <= 0: every point
   *   rot_relative  PPFRegistration::posesWithinErrorBounds tests the angle of the RELATIVE rotation of two poses,
   *                 OpenCV the difference of their rotation angles */
  bool key_exact = false;
  double pair_radius = 0.0;
  bool rot_relative = false;
  /*   alpha_2pi     PPFRegistration wraps alpha_m - alpha_s into [-pi, pi] and bins it over 2 pi (numAngles bins of
   *                 2 pi / numAngles); OpenCV bins the unwrapped difference over 4 pi */
  bool alpha_2pi = false;
  /*   darboux       PPFEstimation's pair feature (pcl::computePairFeatures) instead of the three acos angles; fixed at training */
  bool darboux = false;
  std::vector<int32_t> pair_key; /* N^2 x 4: quantised key of every model pair */
};

/* upstream PPF3DDetector ctor + setSearchParams() defaults. */
void initModel(Model& m, double relSampling, double relDistance, double numAngles) {
  m.sampling_step_relative = relSampling;
  m.distance_step_relative = relDistance;
  m.angle_step_relative = numAngles;
  m.angle_step_radians = (360.0 / numAngles) * M_PI / 180.0;
  m.angle_step = m.angle_step_radians;
  m.position_threshold = relSampling;                          /* upstream quirk: relative number used as metres */
  m.rotation_threshold = ((360 / m.angle_step) / 180.0 * M_PI); /* ~30 rad: no effective rotation gate */
  m.use_weighted_avg = false;
}

/* upstream PPF3DDetector::trainModel(). flags bit0: input already sampled (skip A2);
 * bit2: distance step = diameter * relDistance instead of upstream's diameter * relSampling. */
template <class M>
void trainModel(Model& m, const float* pc, int n, int stride, int flags) {
  float lo[3], hi[3];
  computeBbox(pc, n, stride, lo, hi);
  float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
  float diameter = std::sqrt(dx * dx + dy * dy + dz * dz);
  /* upstream: float distanceStep = (float)(diameter * sampling_step_relative);  (NOT the distance step) */
  float distanceStep = (float)(diameter * ((flags & 4) ? m.distance_step_relative : m.sampling_step_relative));
  m.darboux = (flags & 8) != 0; /* policy: PCL's pair feature, a property of the trained table */
  if (flags & 1) {
    m.sampled_pc.resize((size_t)n * 6);
    for (int i = 0; i < n; i++) memcpy(&m.sampled_pc[(size_t)i * 6], pc + (size_t)i * stride, 24);
  } else {
    m.sampled_pc = samplePCByQuantization(pc, n, stride, lo, hi, (float)m.sampling_step_relative);
  }
  const int N = (int)(m.sampled_pc.size() / 6);
  const size_t NN = (size_t)N * N;
  uint32_t sz = (uint32_t)NN;
  m.slots = nextPow2(sz < 16 ? 16 : sz);
  m.bucket_head.assign(m.slots, -1);
  m.ppf.assign(NN * 5, 0.f);
  m.hash_nodes.assign(NN, THash{0, 0, 0});
  m.pair_key.assign(NN * 4, 0);
  m.list_nodes.clear();
  m.list_nodes.reserve(NN);
  const float* S = m.sampled_pc.data();
  for (int i = 0; i < N; i++) {
    const V3 p1 = v3(S + (size_t)i * 6), n1 = v3(S + (size_t)i * 6 + 3);
    for (int j = 0; j < N; j++) {
      if (i == j) continue;
      const V3 p2 = v3(S + (size_t)j * 6), n2 = v3(S + (size_t)j * 6 + 3);
      double f[4] = {0, 0, 0, 0};
      int ppfInd = i * N + j;
      uint32_t hashValue;
      if (m.darboux) {
        if (!computePairFeaturesDarboux<M>(p1, n1, p2, n2, f)) continue; /* PCL leaves degenerate pairs out */
        hashValue = hashDarboux(f, m.angle_step_radians, distanceStep, &m.pair_key[(size_t)ppfInd * 4]);
      } else {
        computePPFFeatures<M>(p1, n1, p2, n2, f);
        hashValue = hashPPF(f, m.angle_step_radians, distanceStep, &m.pair_key[(size_t)ppfInd * 4]);
      }
      double alpha = computeAlpha<M>(p1, n1, p2);
      m.hash_nodes[ppfInd] = THash{hashValue, i, ppfInd};
      /* hashtableInsertHashed: prepend to bucket hash % size */
      uint32_t b = hashValue % m.slots;
      m.list_nodes.push_back(HashNode{hashValue, ppfInd, m.bucket_head[b]});
      m.bucket_head[b] = (int)m.list_nodes.size() - 1;
      float* row = &m.ppf[(size_t)ppfInd * 5];
      row[0] = (float)f[0]; row[1] = (float)f[1]; row[2] = (float)f[2]; row[3] = (float)f[3];
      row[4] = (float)alpha;
    }
  }
  m.distance_step = distanceStep;
  m.num_ref_points = N;
  m.trained = true;
}

/* ---- pose algebra (upstream pose_3d.cpp / c_utils.hpp) --------------------------------- */
struct Pose {
  double pose[16];
  double q[4];
  double t[3];
  double angle, alpha, residual;
  uint32_t modelIndex, numVotes;
};

void mat44mul(const double* A, const double* B, double* C) {
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      double s = 0;
      for (int k = 0; k < 4; k++) s += A[i * 4 + k] * B[k * 4 + j];
      C[i * 4 + j] = s;
    }
}
void rtToPose(const M33& R, const V3& t, double* P) {
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) P[i * 4 + j] = R.m[i][j];
  P[3] = t.x; P[7] = t.y; P[11] = t.z;
  P[12] = P[13] = P[14] = 0; P[15] = 1;
}
/* quaternion [w x y z]; the (conjugate-signed) convention of upstream dcmToQuat/quatToDCM pair. */
void dcmToQuat(const M33& Rm, double* q) {
  const double* R = &Rm.m[0][0];
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
  const double factor = 0.5 / std::sqrt(n4);
  for (int k = 0; k < 4; k++) q[k] *= factor;
}
void quatToDCM(const double* q, M33& Rm) {
  double* R = &Rm.m[0][0];
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
template <class M>
double angleFromTrace(double trace) {
  if (std::fabs(trace - 3) <= EPS) return 0;
  if (std::fabs(trace + 1) <= EPS) return M_PI;
  return M::acos_((trace - 1) / 2);
}
/* Pose3D::updatePose(Matx44d) */
template <class M>
void updatePose(Pose& p, const double* P) {
  memcpy(p.pose, P, sizeof(p.pose));
  M33 R;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R.m[i][j] = P[i * 4 + j];
  p.t[0] = P[3]; p.t[1] = P[7]; p.t[2] = P[11];
  p.angle = angleFromTrace<M>(R.m[0][0] + R.m[1][1] + R.m[2][2]);
  dcmToQuat(R, p.q);
}
/* Pose3D::updatePoseQuat(q, t) (t is also stored, so clustered poses report their translation). */
template <class M>
void updatePoseQuat(Pose& p, const double* q, const double* t) {
  M33 R;
  quatToDCM(q, R);
  memcpy(p.q, q, sizeof(p.q));
  memcpy(p.t, t, sizeof(p.t));
  rtToPose(R, V3{t[0], t[1], t[2]}, p.pose);
  p.angle = angleFromTrace<M>(R.m[0][0] + R.m[1][1] + R.m[2][2]);
}

struct VoteResult { uint32_t refIndMax, alphaIndMax, maxVotes; };

/* One scene reference point of upstream PPF3DDetector::match(): fills/clears `acc`
 * (numAngles * N u32), returns the argmax triple and the exact number of increments. */
template <class M>
VoteResult voteOneRef(const Model& m, const float* surf, int /*nSurf*/, const float* paired, int nPaired, bool sameCloud,
                      int i, int numAngles, uint32_t* acc, uint64_t* nVotes, uint64_t* nPairsHashed) {
  const uint32_t n = (uint32_t)m.num_ref_points;
  const size_t accSize = (size_t)numAngles * n;
  const float distanceStep = (float)m.distance_step;
  const V3 p1 = v3(surf + (size_t)i * 6), n1 = v3(surf + (size_t)i * 6 + 3);
  M33 Rsg; V3 tsg;
  computeTransformRT<M>(p1, n1, Rsg, tsg);
  uint64_t votes = 0, pairs = 0;
  for (int j = 0; j < nPaired; j++) {
    if (sameCloud && i == j) continue;
    /* match_S2B (build-defined): the reference point itself is not paired even when it
     * also appears in the edge cloud (bit-identical row), so edge == scene reduces exactly to match() */
    if (!sameCloud && memcmp(surf + (size_t)i * 6, paired + (size_t)j * 6, 24) == 0) continue;
    const V3 p2 = v3(paired + (size_t)j * 6), n2 = v3(paired + (size_t)j * 6 + 3);
    double f[4] = {0, 0, 0, 0};
    int32_t sceneKey[4];
    uint32_t hashValue;
    if (m.darboux) {
      if (!computePairFeaturesDarboux<M>(p1, n1, p2, n2, f)) continue;
      if (m.pair_radius > 0 && f[3] > m.pair_radius) continue;
      hashValue = hashDarboux(f, m.angle_step, distanceStep, sceneKey);
    } else {
      computePPFFeatures<M>(p1, n1, p2, n2, f);
      if (m.pair_radius > 0 && f[3] > m.pair_radius) continue; /* policy: neighbours within a radius only */
      hashValue = hashPPF(f, m.angle_step, distanceStep, sceneKey);
    }
    V3 rp = mulMV(Rsg, p2);
    V3 p2t{tsg.x + rp.x, tsg.y + rp.y, tsg.z + rp.z};
    double alpha_scene;
    if (!alphaFromTransformed<M>(p2t, alpha_scene)) continue;
    pairs++;
    int node = m.bucket_head[hashValue % m.slots]; /* hashtableGetBucketHashed: whole bucket, no key compare */
    while (node >= 0) {
      const HashNode& ln = m.list_nodes[node];
      const THash& tData = m.hash_nodes[ln.data];
      if (m.key_exact && memcmp(&m.pair_key[(size_t)tData.ppfInd * 4], sceneKey, 16) != 0) { node = ln.next; continue; }
      int corrI = tData.i;
      const float* ppfCorrScene = &m.ppf[(size_t)tData.ppfInd * 5];
      double alpha_model = (double)ppfCorrScene[4];
      double alpha = alpha_model - alpha_scene;
      int alpha_index;
      if (m.alpha_2pi) {
        if (alpha < -M_PI) alpha += 2 * M_PI; else if (alpha > M_PI) alpha -= 2 * M_PI;
        alpha_index = d2i(numAngles * (alpha + M_PI) / (2 * M_PI));
        if (alpha_index >= numAngles) alpha_index = numAngles - 1; /* alpha == +pi exactly */
      } else {
        alpha_index = d2i(numAngles * (alpha + 2 * M_PI) / (4 * M_PI));
      }
      size_t accIndex = (size_t)((int64_t)corrI * numAngles + alpha_index);
      /* alpha_index == numAngles happens (alpha_model is a float: (float)pi > pi) and upstream then
       * increments the next reference point's bin 0, or writes past the buffer for the last one.
       * The in-range spill is kept; the out-of-buffer write is dropped. */
      if (accIndex < accSize) { acc[accIndex]++; votes++; }
      node = ln.next;
    }
  }
  VoteResult r{0, 0, 0};
  for (uint32_t k = 0; k < n; k++)
    for (int j = 0; j < numAngles; j++) {
      const size_t accInd = (size_t)k * numAngles + j;
      const uint32_t accVal = acc[accInd];
      if (accVal > r.maxVotes) { r.maxVotes = accVal; r.refIndMax = k; r.alphaIndMax = (uint32_t)j; }
    }
  if (nVotes) *nVotes = votes;
  if (nPairsHashed) *nPairsHashed = pairs;
  return r;
}

/* rawPose = TsgInv * (Talpha * Tmg) of upstream match(). */
template <class M>
void assemblePose(const Model& m, const float* surf, int i, const VoteResult& v, int numAngles, Pose& out) {
  const V3 p1 = v3(surf + (size_t)i * 6), n1 = v3(surf + (size_t)i * 6 + 3);
  M33 Rsg; V3 tsg;
  computeTransformRT<M>(p1, n1, Rsg, tsg);
  M33 RInv;
  for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) RInv.m[a][b] = Rsg.m[b][a];
  V3 rt = mulMV(RInv, tsg);
  V3 tInv{-rt.x, -rt.y, -rt.z};
  double TsgInv[16], Tmg[16], Talpha[16], tmp[16], raw[16];
  rtToPose(RInv, tInv, TsgInv);
  const float* pm = &m.sampled_pc[(size_t)v.refIndMax * 6];
  M33 Rmg; V3 tmg;
  computeTransformRT<M>(v3(pm), v3(pm + 3), Rmg, tmg);
  rtToPose(Rmg, tmg, Tmg);
  int alpha_index = (int)v.alphaIndMax;
  double alpha = m.alpha_2pi ? (alpha_index * (2 * M_PI)) / numAngles - M_PI : (alpha_index * (4 * M_PI)) / numAngles - 2 * M_PI;
  const double sx = M::sin_(alpha), cx = M::cos_(alpha);
  M33 Rx{{{1, 0, 0}, {0, cx, -sx}, {0, sx, cx}}};
  rtToPose(Rx, V3{0, 0, 0}, Talpha);
  mat44mul(Talpha, Tmg, tmp);
  mat44mul(TsgInv, tmp, raw);
  memset(&out, 0, sizeof(out));
  out.alpha = alpha; out.modelIndex = v.refIndMax; out.numVotes = v.maxVotes; out.residual = 0;
  updatePose<M>(out, raw);
}

/* upstream clusterPoses()/matchPose()/PoseCluster3D.  std::sort ties are implementation-ordered
 * upstream; the frozen total order is (votes desc, input order asc) for poses and
 * (cluster votes desc, creation order asc) for clusters. */
template <class M>
void clusterPoses(const Model& m, std::vector<Pose>& poseList, int numPoses, std::vector<Pose>& finalPoses) {
  std::vector<int> order(poseList.size());
  for (size_t i = 0; i < order.size(); i++) order[i] = (int)i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return poseList[a].numVotes > poseList[b].numVotes; });
  struct Cluster { std::vector<int> members; uint64_t numVotes; };
  std::vector<Cluster> clusters;
  if (numPoses > (int)order.size()) numPoses = (int)order.size();
  for (int i = 0; i < numPoses; i++) {
    const Pose& pose = poseList[order[i]];
    bool assigned = false;
    for (size_t j = 0; j < clusters.size() && !assigned; j++) {
      const Pose& c = poseList[clusters[j].members[0]];
      double dvx = c.t[0] - pose.t[0], dvy = c.t[1] - pose.t[1], dvz = c.t[2] - pose.t[2];
      double dNorm = std::sqrt(dvx * dvx + dvy * dvy + dvz * dvz);
      bool rotOk;
      if (m.rot_relative) {
        /* angle of Ra^T Rb = 2 acos(|qa . qb|) < threshold  <=>  |qa . qb| > cos(threshold / 2)  (threshold in [0, 2 pi]) */
        const double d = std::fabs(c.q[0] * pose.q[0] + c.q[1] * pose.q[1] + c.q[2] * pose.q[2] + c.q[3] * pose.q[3]);
        rotOk = d > M::cos_(0.5 * m.rotation_threshold);
      } else {
        rotOk = std::fabs(pose.angle - c.angle) < m.rotation_threshold;
      }
      if (rotOk && dNorm < m.position_threshold) {
        clusters[j].members.push_back(order[i]);
        clusters[j].numVotes += pose.numVotes;
        assigned = true;
      }
    }
    if (!assigned) clusters.push_back(Cluster{{order[i]}, pose.numVotes});
  }
  std::vector<int> corder(clusters.size());
  for (size_t i = 0; i < corder.size(); i++) corder[i] = (int)i;
  std::stable_sort(corder.begin(), corder.end(), [&](int a, int b) { return clusters[a].numVotes > clusters[b].numVotes; });
  finalPoses.clear();
  finalPoses.reserve(clusters.size());
  for (size_t ci = 0; ci < corder.size(); ci++) {
    const Cluster& cl = clusters[corder[ci]];
    double qAvg[4] = {0, 0, 0, 0}, tAvg[3] = {0, 0, 0};
    const int curSize = (int)cl.members.size();
    if (m.use_weighted_avg) {
      double wSum = 0;
      for (int j = 0; j < curSize; j++) {
        const Pose& p = poseList[cl.members[j]];
        const double w = (double)p.numVotes;
        for (int k = 0; k < 4; k++) qAvg[k] += w * p.q[k];
        for (int k = 0; k < 3; k++) tAvg[k] += w * p.t[k];
        wSum += w;
      }
      for (int k = 0; k < 3; k++) tAvg[k] *= 1.0 / wSum;
      for (int k = 0; k < 4; k++) qAvg[k] *= 1.0 / wSum;
    } else {
      for (int j = 0; j < curSize; j++) {
        const Pose& p = poseList[cl.members[j]];
        for (int k = 0; k < 4; k++) qAvg[k] += p.q[k];
        for (int k = 0; k < 3; k++) tAvg[k] += p.t[k];
      }
      for (int k = 0; k < 3; k++) tAvg[k] *= 1.0 / curSize;
      for (int k = 0; k < 4; k++) qAvg[k] *= 1.0 / curSize;
    }
    Pose out = poseList[cl.members[0]];
    updatePoseQuat<M>(out, qAvg, tAvg);
    out.numVotes = (uint32_t)cl.numVotes;
    finalPoses.push_back(out);
  }
}

struct MatchOut {
  std::vector<float> sampledScene, sampledEdge;
  std::vector<int> refIdx;
  std::vector<VoteResult> votes;
  std::vector<uint64_t> nVotes, nPairs;
  std::vector<Pose> rawPoses, finalPoses;
};

/* upstream PPF3DDetector::match(); with `edge` != null the build-defined match_S2B
 * (reference points walk the sampled surface cloud with stride sceneStep,
 * paired points walk the sampled edge cloud; edge == scene reduces to match()).
 * flags bit0: clouds are already sampled (skip samplePCByQuantization).
 * refList != null: vote only for these sampled-scene indices (oracle-side sampling of big cases). */
template <class M>
void matchImpl(const Model& m, const float* scene, int ns, int sstride, const float* edge, int ne, int estride,
               double relSceneSampleStep, double relSceneDistance, int flags, const int* refList, int nRefList,
               int threads, bool doCluster, MatchOut& out) {
  const int sceneSamplingStep = (int)(1.0 / relSceneSampleStep);
  const int numAngles = (int)(std::floor(2 * M_PI / m.angle_step));
  auto sampleCloud = [&](const float* pc, int n, int stride) {
    std::vector<float> s;
    if (flags & 1) {
      s.resize((size_t)n * 6);
      for (int i = 0; i < n; i++) memcpy(&s[(size_t)i * 6], pc + (size_t)i * stride, 24);
    } else {
      float lo[3], hi[3];
      computeBbox(pc, n, stride, lo, hi);
      s = samplePCByQuantization(pc, n, stride, lo, hi, (float)relSceneDistance);
    }
    return s;
  };
  out.sampledScene = sampleCloud(scene, ns, sstride);
  const int rows = (int)(out.sampledScene.size() / 6);
  const float* paired = out.sampledScene.data();
  int nPaired = rows;
  bool same = true;
  if (edge) {
    out.sampledEdge = sampleCloud(edge, ne, estride);
    paired = out.sampledEdge.data();
    nPaired = (int)(out.sampledEdge.size() / 6);
    same = false;
  }
  out.refIdx.clear();
  if (refList) out.refIdx.assign(refList, refList + nRefList);
  else for (int i = 0; i < rows; i += sceneSamplingStep) out.refIdx.push_back(i);
  const int nRef = (int)out.refIdx.size();
  out.votes.assign(nRef, VoteResult{0, 0, 0});
  out.nVotes.assign(nRef, 0);
  out.nPairs.assign(nRef, 0);
  out.rawPoses.resize(nRef);
  const size_t accSize = (size_t)numAngles * m.num_ref_points;
#ifdef _OPENMP
  if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel num_threads(threads)
#endif
  {
    std::vector<uint32_t> acc(accSize, 0u);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
    for (int r = 0; r < nRef; r++) {
      std::fill(acc.begin(), acc.end(), 0u); /* upstream: calloc per reference point */
      out.votes[r] = voteOneRef<M>(m, out.sampledScene.data(), rows, paired, nPaired, same, out.refIdx[r], numAngles,
                                   acc.data(), &out.nVotes[r], &out.nPairs[r]);
      assemblePose<M>(m, out.sampledScene.data(), out.refIdx[r], out.votes[r], numAngles, out.rawPoses[r]);
    }
  }
  if (doCluster) {
    /* upstream: numPosesAdded = sampled.rows / sceneSamplingStep (integer division; may drop the last pose) */
    int numPosesAdded = refList ? nRef : rows / sceneSamplingStep;
    std::vector<Pose> poseList = out.rawPoses;
    clusterPoses<M>(m, poseList, numPosesAdded, out.finalPoses);
  }
}

}  // namespace

/* =============================== C entry points (ctypes) ================================ */
extern "C" {

struct oracle_pose {
  double pose[16];
  double q[4];
  double t[3];
  double angle, alpha, residual;
  uint32_t model_index, num_votes;
};

static void toC(const Pose& p, oracle_pose* o) {
  memcpy(o->pose, p.pose, sizeof(o->pose));
  memcpy(o->q, p.q, sizeof(o->q));
  memcpy(o->t, p.t, sizeof(o->t));
  o->angle = p.angle; o->alpha = p.alpha; o->residual = p.residual;
  o->model_index = p.modelIndex; o->num_votes = p.numVotes;
}

void* oracle_train(const float* pc, int n, int stride, double relSampling, double relDistance, double numAngles,
                   int flags, int mode) {
  if (!pc || n <= 0 || stride < 6) return nullptr;
  Model* m = new Model();
  initModel(*m, relSampling, relDistance, numAngles);
  m->mode = mode;
  if (mode == 1) trainModel<MathLibm>(*m, pc, n, stride, flags);
  else trainModel<MathDet>(*m, pc, n, stride, flags);
  return m;
}
void oracle_free(void* h) { delete (Model*)h; }

void oracle_set_search_params(void* h, double positionThreshold, double rotationThreshold, int useWeighted) {
  Model* m = (Model*)h;
  m->position_threshold = positionThreshold < 0 ? m->sampling_step_relative : positionThreshold;
  m->rotation_threshold = rotationThreshold < 0 ? ((360 / m->angle_step) / 180.0 * M_PI) : rotationThreshold;
  m->use_weighted_avg = useWeighted != 0;
}

void oracle_set_policy(void* h, int keyExact, double pairRadius, int rotRelative, int alpha2pi) {
  Model* m = (Model*)h;
  m->key_exact = keyExact != 0;
  m->pair_radius = pairRadius;
  m->rot_relative = rotRelative != 0;
  m->alpha_2pi = alpha2pi != 0;
}

void oracle_model_info(void* h, int* nRef, uint32_t* slots, double* angleStep, double* distanceStep, int* numAngles) {
  Model* m = (Model*)h;
  if (nRef) *nRef = m->num_ref_points;
  if (slots) *slots = m->slots;
  if (angleStep) *angleStep = m->angle_step;
  if (distanceStep) *distanceStep = m->distance_step;
  if (numAngles) *numAngles = (int)(std::floor(2 * M_PI / m->angle_step));
}
void oracle_model_sampled(void* h, float* out) {
  Model* m = (Model*)h;
  memcpy(out, m->sampled_pc.data(), m->sampled_pc.size() * sizeof(float));
}
/* per-pair training products, for parity of the device-side table build:
 * hash[i*N+j], alpha_m[i*N+j] (float), 0 on the diagonal */
void oracle_model_pairs(void* h, uint32_t* hash, float* alpha) {
  Model* m = (Model*)h;
  const size_t NN = (size_t)m->num_ref_points * m->num_ref_points;
  for (size_t k = 0; k < NN; k++) {
    if (hash) hash[k] = m->hash_nodes[k].id;
    if (alpha) alpha[k] = m->ppf[k * 5 + 4];
  }
}
/* bucket occupancy statistics: number of non-empty slots, the longest chain, sum of squares */
void oracle_model_bucket_stats(void* h, uint64_t* nonEmpty, uint64_t* maxLen, double* sumSq) {
  Model* m = (Model*)h;
  std::vector<uint32_t> len(m->slots, 0);
  for (const HashNode& n : m->list_nodes) len[n.key % m->slots]++;
  uint64_t ne = 0, mx = 0; double s2 = 0;
  for (uint32_t l : len) { ne += (l > 0); if (l > mx) mx = l; s2 += (double)l * l; }
  if (nonEmpty) *nonEmpty = ne;
  if (maxLen) *maxLen = mx;
  if (sumSq) *sumSq = s2;
}

/* Sampling alone (A2), for parity of the device sampler. Returns rows; writes up to cap rows. */
int oracle_sample(const float* pc, int n, int stride, double relStep, float* out, int cap) {
  float lo[3], hi[3];
  computeBbox(pc, n, stride, lo, hi);
  std::vector<float> s = samplePCByQuantization(pc, n, stride, lo, hi, (float)relStep);
  int rows = (int)(s.size() / 6);
  if (out) memcpy(out, s.data(), sizeof(float) * 6 * (size_t)std::min(rows, cap));
  return rows;
}

/*
 * Full match.  Outputs (any may be null):
 *   triples[3*r]      {refIndMax, alphaIndMax, maxVotes} per voted reference point
 *   votesPerRef[r]    exact number of accumulator increments
 *   pairsPerRef[r]    scene pairs hashed and looked up
 *   rawPoses[r]       per-reference pose before clustering
 *   finalPoses[c]     clustered poses (cap finalCap), *nFinal set
 *   sampledOut        sampled scene rows (cap sampledCap rows), *nSampled set
 * Returns the number of voted reference points, or -1 on error.
 */
int oracle_match(void* h, const float* scene, int ns, int sstride, const float* edge, int ne, int estride,
                 double relSceneSampleStep, double relSceneDistance, int flags, const int* refList, int nRefList,
                 int threads, uint32_t* triples, uint64_t* votesPerRef, uint64_t* pairsPerRef, oracle_pose* rawPoses,
                 int rawCap, oracle_pose* finalPoses, int finalCap, int* nFinal, float* sampledOut, int sampledCap,
                 int* nSampled) {
  Model* m = (Model*)h;
  if (!m || !m->trained || !scene || ns <= 0) return -1;
  if (!(relSceneSampleStep <= 1 && relSceneSampleStep > 0)) return -1;
  MatchOut out;
  const bool doCluster = finalPoses != nullptr || nFinal != nullptr;
  if (m->mode == 1)
    matchImpl<MathLibm>(*m, scene, ns, sstride, edge, ne, estride, relSceneSampleStep, relSceneDistance, flags, refList,
                        nRefList, threads, doCluster, out);
  else
    matchImpl<MathDet>(*m, scene, ns, sstride, edge, ne, estride, relSceneSampleStep, relSceneDistance, flags, refList,
                       nRefList, threads, doCluster, out);
  const int nRef = (int)out.refIdx.size();
  for (int r = 0; r < nRef; r++) {
    if (triples) { triples[3 * r] = out.votes[r].refIndMax; triples[3 * r + 1] = out.votes[r].alphaIndMax; triples[3 * r + 2] = out.votes[r].maxVotes; }
    if (votesPerRef) votesPerRef[r] = out.nVotes[r];
    if (pairsPerRef) pairsPerRef[r] = out.nPairs[r];
    if (rawPoses && r < rawCap) toC(out.rawPoses[r], &rawPoses[r]);
  }
  if (nFinal) *nFinal = (int)out.finalPoses.size();
  if (finalPoses) for (int c = 0; c < (int)out.finalPoses.size() && c < finalCap; c++) toC(out.finalPoses[c], &finalPoses[c]);
  const int rows = (int)(out.sampledScene.size() / 6);
  if (nSampled) *nSampled = rows;
  if (sampledOut) memcpy(sampledOut, out.sampledScene.data(), sizeof(float) * 6 * (size_t)std::min(rows, sampledCap));
  return nRef;
}

/* Full accumulator of one reference point (tiny known-answer cases). acc: numAngles*N u32. */
int oracle_accumulator(void* h, const float* sampledScene, int rows, const float* sampledPaired, int nPaired, int i,
                       uint32_t* acc) {
  Model* m = (Model*)h;
  const int numAngles = (int)(std::floor(2 * M_PI / m->angle_step));
  const size_t accSize = (size_t)numAngles * m->num_ref_points;
  memset(acc, 0, accSize * sizeof(uint32_t));
  const float* paired = sampledPaired ? sampledPaired : sampledScene;
  const int np = sampledPaired ? nPaired : rows;
  if (m->mode == 1) voteOneRef<MathLibm>(*m, sampledScene, rows, paired, np, !sampledPaired, i, numAngles, acc, nullptr, nullptr);
  else voteOneRef<MathDet>(*m, sampledScene, rows, paired, np, !sampledPaired, i, numAngles, acc, nullptr, nullptr);
  return (int)accSize;
}

/* Cluster a caller-supplied pose list (parity of the device clustering kernel). */
int oracle_cluster(void* h, const oracle_pose* in, int n, int numPoses, oracle_pose* out, int cap) {
  Model* m = (Model*)h;
  std::vector<Pose> list(n);
  for (int i = 0; i < n; i++) {
    memcpy(list[i].pose, in[i].pose, sizeof(list[i].pose));
    memcpy(list[i].q, in[i].q, sizeof(list[i].q));
    memcpy(list[i].t, in[i].t, sizeof(list[i].t));
    list[i].angle = in[i].angle; list[i].alpha = in[i].alpha; list[i].residual = in[i].residual;
    list[i].modelIndex = in[i].model_index; list[i].numVotes = in[i].num_votes;
  }
  std::vector<Pose> fin;
  if (m->mode == 1) clusterPoses<MathLibm>(*m, list, numPoses, fin);
  else clusterPoses<MathDet>(*m, list, numPoses, fin);
  for (int c = 0; c < (int)fin.size() && c < cap; c++) toC(fin[c], &out[c]);
  return (int)fin.size();
}

/* ---- unit-level probes ----------------------------------------------------------------- */
void oracle_murmur3_x64_128(const void* key, int len, uint32_t seed, uint64_t* out2) { murmur3_x64_128(key, len, seed, out2); }

/* f[4], key[4], hash for one pair (p1,n1,p2,n2 as 3 floats each) */
uint32_t oracle_pair_feature(const float* p1, const float* n1, const float* p2, const float* n2, double angleStep,
                             double distStep, int mode, double* f4, int32_t* key4) {
  double f[4] = {0, 0, 0, 0};
  if (mode == 1) computePPFFeatures<MathLibm>(v3(p1), v3(n1), v3(p2), v3(n2), f);
  else computePPFFeatures<MathDet>(v3(p1), v3(n1), v3(p2), v3(n2), f);
  if (f4) memcpy(f4, f, sizeof(f));
  return hashPPF(f, angleStep, distStep, key4);
}
/* the same for PCL's feature; returns 0 for a degenerate pair (f, key untouched), 1 otherwise with *hash set */
int oracle_pair_feature_darboux(const float* p1, const float* n1, const float* p2, const float* n2, double angleStep,
                                double distStep, int mode, double* f4, int32_t* key4, uint32_t* hash) {
  double f[4] = {0, 0, 0, 0};
  const bool ok = mode == 1 ? computePairFeaturesDarboux<MathLibm>(v3(p1), v3(n1), v3(p2), v3(n2), f)
                            : computePairFeaturesDarboux<MathDet>(v3(p1), v3(n1), v3(p2), v3(n2), f);
  if (!ok) return 0;
  if (f4) memcpy(f4, f, sizeof(f));
  const uint32_t h = hashDarboux(f, angleStep, distStep, key4);
  if (hash) *hash = h;
  return 1;
}
void oracle_transform_rt(const float* p, const float* n, int mode, double* R9, double* t3) {
  M33 R; V3 t;
  if (mode == 1) computeTransformRT<MathLibm>(v3(p), v3(n), R, t);
  else computeTransformRT<MathDet>(v3(p), v3(n), R, t);
  memcpy(R9, R.m, sizeof(R.m));
  t3[0] = t.x; t3[1] = t.y; t3[2] = t.z;
}
double oracle_alpha(const float* p1, const float* n1, const float* p2, int mode) {
  return mode == 1 ? computeAlpha<MathLibm>(v3(p1), v3(n1), v3(p2)) : computeAlpha<MathDet>(v3(p1), v3(n1), v3(p2));
}
/* elementwise detmath / libm evaluation: fn 0 acos, 1 sin, 2 cos, 3 atan2(x=y-arg, x2) */
void oracle_math_eval(int fn, int mode, const double* x, const double* x2, double* out, int n) {
  for (int i = 0; i < n; i++) {
    switch (fn) {
      case 0: out[i] = mode ? std::acos(x[i]) : ppf_acos(x[i]); break;
      case 1: out[i] = mode ? std::sin(x[i]) : ppf_sin(x[i]); break;
      case 2: out[i] = mode ? std::cos(x[i]) : ppf_cos(x[i]); break;
      default: out[i] = mode ? std::atan2(x[i], x2[i]) : ppf_atan2(x[i], x2[i]); break;
    }
  }
}
/* pose helpers for round-trip tests */
void oracle_dcm_to_quat(const double* R9, double* q4) { M33 R; memcpy(R.m, R9, sizeof(R.m)); dcmToQuat(R, q4); }
void oracle_quat_to_dcm(const double* q4, double* R9) { M33 R; quatToDCM(q4, R); memcpy(R9, R.m, sizeof(R.m)); }
int oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

}  // extern "C"
