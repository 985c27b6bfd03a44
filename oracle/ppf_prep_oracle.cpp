/*
 * ppf_prep_oracle.cpp — CPU ORACLE for the stages that PRODUCE the N x 6 cloud the matcher consumes (SURVEY.md §8f
 * row N4).  TEST INFRASTRUCTURE ONLY (same rules as ppf_oracle.cpp).
 *
 * WHAT IT RESTATES (all of it PCL called from /root/reference/include/CloudProcessing.h, driver order
 * /root/reference/src/YOLO_cropping_ppf_test.cpp:91-103):
 *   SceneCropping      :263-339   bbox +-30 px, mean corner depth, 4 back-projected corners pushed 0.15 m back + the
 *                                 camera centre -> ConvexHull + CropHull
 *   Subsampling        :361-380   pcl::VoxelGrid(leaf)
 *   OutlierProcessing  :341-360   pcl::StatisticalOutlierRemoval(meanK = 50, stddevMul)
 *   NormalEstimation   :381-405   pcl::NormalEstimationOMP(k = 30), viewpoint (0,0,0)
 *   EdgeExtraction     :406-427   curvature > 0.03
 *   PointCloudXYZNormalToMat :163-190   normals re-normalised into the N x 6 float Mat
 *
 * PARITY STATUS: **parity unpinned**.  PCL is neither vendored nor installed and the reference holds no outputs of
 * these stages.  Choices frozen here (the device kernels reproduce THIS file bit for bit):
 *   - crop: the hull of {4 corners, origin} is a pyramid; a point is kept when it is on the inner side of (or on) the
 *     four side planes through the origin and not behind the base plane (z <= z_base).  Plane normals a x b and the
 *     dot products in fp64.  (PCL casts three rays per point against the hull triangles; results can differ only for
 *     points within rounding of the boundary.)
 *   - voxel grid: PCL's index arithmetic (float floor(p * inv_leaf) - min_b, x fastest); cells emitted in ascending
 *     cell index; centroid = float sums IN ASCENDING POINT ORDER, divided by the float count.  (PCL's std::sort
 *     leaves the order inside a cell unspecified.)  Non-finite points are dropped.
 *   - k nearest neighbours: exact, float squared distance ((dx*dx + dy*dy) + dz*dz), ordered by (distance, index).
 *   - outlier removal: mean of sqrtf(d2) over neighbours 1..meanK accumulated in fp64 in neighbour order; population
 *     sums of the float distances in fp64 in chunks of 64 points; keep when !(d > mean + mul * stddev).
 *   - normals: fp64 two-pass covariance of the k neighbours (in neighbour order), cyclic Jacobi (fixed 12 sweeps,
 *     only + - * / sqrt), eigenvector of the smallest eigenvalue, re-normalised, flipped towards the origin;
 *     curvature = |lambda_min / trace(cov)|.  (PCL: float single-pass covariance + closed-form eigen33.)
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <utility>
#include <vector>

#include "../include/ppf_detmath.h"

namespace {

inline bool finite3(const float* p) { return std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]); }

/* Camera::back_projection_bbox (Camera.h:50-61): x = (float)(u - ppx) * z / fx, float z, double intrinsics */
void back_project(float z, int u, int v, double fx, double fy, double ppx, double ppy, float* out) {
  out[0] = (float)((double)((float)((double)u - ppx) * z) / fx);
  out[1] = (float)((double)((float)((double)v - ppy) * z) / fy);
  out[2] = z;
}

/* exact kNN by brute force: the k smallest (d2, index) pairs of every point, ascending */
void knn_all(const float* xyz, int n, int stride, int k, std::vector<int>& idx, std::vector<float>& d2) {
  idx.assign((size_t)n * k, -1);
  d2.assign((size_t)n * k, 0.f);
#pragma omp parallel num_threads(16) if ((long)n * n > 4000000L)
  {
    std::vector<std::pair<float, int>> cand((size_t)n);
#pragma omp for schedule(static)
    for (int i = 0; i < n; i++) {
      const float* p = xyz + (size_t)i * stride;
      for (int j = 0; j < n; j++) {
        const float* q = xyz + (size_t)j * stride;
        const float dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
        cand[(size_t)j] = {(dx * dx + dy * dy) + dz * dz, j};
      }
      std::partial_sort(cand.begin(), cand.begin() + k, cand.end());
      for (int m = 0; m < k; m++) { idx[(size_t)i * k + m] = cand[(size_t)m].second; d2[(size_t)i * k + m] = cand[(size_t)m].first; }
    }
  }
}

/* cyclic Jacobi on a symmetric 3x3 (a: 0 1 2 / 1 3 4 / 2 4 5 packed as full 3x3), V accumulates the rotations */
void jacobi3(double A[3][3], double V[3][3]) {
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) V[i][j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 12; sweep++) {
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        const double apq = A[p][q];
        if (apq == 0.0) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
        const double at = theta < 0 ? -theta : theta;
        double t = 1.0 / (at + ppf_sqrt(theta * theta + 1.0));
        if (theta < 0) t = -t;
        const double c = 1.0 / ppf_sqrt(t * t + 1.0), s = t * c;
        const double app = A[p][p], aqq = A[q][q];
        A[p][p] = app - t * apq;
        A[q][q] = aqq + t * apq;
        A[p][q] = 0.0; A[q][p] = 0.0;
        const int r = 3 - p - q;
        const double arp = A[r][p], arq = A[r][q];
        A[r][p] = c * arp - s * arq; A[p][r] = A[r][p];
        A[r][q] = s * arp + c * arq; A[q][r] = A[r][q];
        for (int k = 0; k < 3; k++) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
      }
  }
}

}  // namespace

extern "C" {

/* SceneCropping for one box.  box = {x, y, width, height} (cv::Rect), intr = {fx, fy, ppx, ppy}.
 * planes_out (optional): 4 inward side-plane normals (12 doubles) followed by z_base. */
int oracle_prep_crop(const float* xyz, int n, int stride, const int* box, const float* depth, int drows, int dcols,
                     const double* intr, int* keep_idx, int* n_out, double* planes_out) {
  double left = box[0] - 30; if (left < 0) left = 0;
  double top = box[1] - 30; if (top < 0) top = 0;
  double right = box[0] + box[2] + 30; if (right >= dcols) right = dcols - 1;
  double bottom = box[1] + box[3] + 30; if (bottom >= drows) bottom = drows - 1;
  const int il = (int)left, it = (int)top, ir = (int)right, ib = (int)bottom;
  const float d1 = depth[(size_t)it * dcols + il], d2 = depth[(size_t)it * dcols + ir], d3 = depth[(size_t)ib * dcols + il],
              d4 = depth[(size_t)ib * dcols + ir];
  const float davg = (d1 + d2 + d3 + d4) / 4;
  float c[4][3]; /* left_top, left_bot, right_top, right_bot */
  back_project(davg, il, it, intr[0], intr[1], intr[2], intr[3], c[0]);
  back_project(davg, il, ib, intr[0], intr[1], intr[2], intr[3], c[1]);
  back_project(davg, ir, it, intr[0], intr[1], intr[2], intr[3], c[2]);
  back_project(davg, ir, ib, intr[0], intr[1], intr[2], intr[3], c[3]);
  for (int k = 0; k < 4; k++) c[k][2] = (float)((double)c[k][2] + 0.15);
  const float zb = c[0][2];
  const double ctr[3] = {((double)c[0][0] + c[1][0] + c[2][0] + c[3][0]) / 4, ((double)c[0][1] + c[1][1] + c[2][1] + c[3][1]) / 4, (double)zb};
  const int face[4][2] = {{0, 1}, {1, 3}, {3, 2}, {2, 0}};
  double nrm[4][3];
  for (int f = 0; f < 4; f++) {
    const float* a = c[face[f][0]]; const float* b = c[face[f][1]];
    nrm[f][0] = (double)a[1] * b[2] - (double)a[2] * b[1];
    nrm[f][1] = (double)a[2] * b[0] - (double)a[0] * b[2];
    nrm[f][2] = (double)a[0] * b[1] - (double)a[1] * b[0];
    const double s = nrm[f][0] * ctr[0] + nrm[f][1] * ctr[1] + nrm[f][2] * ctr[2];
    if (s < 0) { nrm[f][0] = -nrm[f][0]; nrm[f][1] = -nrm[f][1]; nrm[f][2] = -nrm[f][2]; }
  }
  if (planes_out) { memcpy(planes_out, nrm, sizeof(nrm)); planes_out[12] = (double)zb; }
  int m = 0;
  for (int i = 0; i < n; i++) {
    const float* p = xyz + (size_t)i * stride;
    bool in = finite3(p) && p[2] <= zb;
    for (int f = 0; f < 4 && in; f++) in = (nrm[f][0] * (double)p[0] + nrm[f][1] * (double)p[1] + nrm[f][2] * (double)p[2]) >= 0.0;
    if (in) keep_idx[m++] = i;
  }
  *n_out = m;
  return 0;
}

/* pcl::VoxelGrid<PointXYZ>::applyFilter with leaf (x = y = z) */
int oracle_prep_voxel(const float* xyz, int n, int stride, float leaf, float* out_xyz, int* n_out) {
  float mn[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, mx[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
  for (int i = 0; i < n; i++) {
    const float* p = xyz + (size_t)i * stride;
    if (!finite3(p)) continue;
    for (int k = 0; k < 3; k++) { mn[k] = std::min(mn[k], p[k]); mx[k] = std::max(mx[k], p[k]); }
  }
  const float inv = 1.0f / leaf;
  int min_b[3], max_b[3], div_b[3];
  for (int k = 0; k < 3; k++) {
    min_b[k] = (int)std::floor(mn[k] * inv);
    max_b[k] = (int)std::floor(mx[k] * inv);
    div_b[k] = max_b[k] - min_b[k] + 1;
  }
  if ((int64_t)div_b[0] * div_b[1] * div_b[2] > 0x7fffffffLL) return 1; /* "Leaf size is too small" */
  std::vector<std::pair<int64_t, int>> key;
  for (int i = 0; i < n; i++) {
    const float* p = xyz + (size_t)i * stride;
    if (!finite3(p)) continue;
    const int i0 = (int)(std::floor(p[0] * inv) - (float)min_b[0]);
    const int i1 = (int)(std::floor(p[1] * inv) - (float)min_b[1]);
    const int i2 = (int)(std::floor(p[2] * inv) - (float)min_b[2]);
    key.push_back({(int64_t)i0 + (int64_t)i1 * div_b[0] + (int64_t)i2 * div_b[0] * div_b[1], i});
  }
  std::sort(key.begin(), key.end());
  int m = 0;
  size_t s = 0;
  while (s < key.size()) {
    size_t e = s;
    float acc[3] = {0.f, 0.f, 0.f};
    while (e < key.size() && key[e].first == key[s].first) {
      const float* p = xyz + (size_t)key[e].second * stride;
      acc[0] += p[0]; acc[1] += p[1]; acc[2] += p[2];
      e++;
    }
    const float cnt = (float)(e - s);
    out_xyz[(size_t)m * 3] = acc[0] / cnt; out_xyz[(size_t)m * 3 + 1] = acc[1] / cnt; out_xyz[(size_t)m * 3 + 2] = acc[2] / cnt;
    m++;
    s = e;
  }
  *n_out = m;
  return 0;
}

int oracle_prep_knn(const float* xyz, int n, int stride, int k, int* idx_out, float* d2_out) {
  std::vector<int> idx; std::vector<float> d2;
  const int ke = std::min(k, n);
  knn_all(xyz, n, stride, ke, idx, d2);
  for (int i = 0; i < n; i++)
    for (int m = 0; m < k; m++) {
      idx_out[(size_t)i * k + m] = m < ke ? idx[(size_t)i * ke + m] : -1;
      d2_out[(size_t)i * k + m] = m < ke ? d2[(size_t)i * ke + m] : 0.f;
    }
  return 0;
}

/* pcl::StatisticalOutlierRemoval: keep[i] = 1 when the point survives; distances (optional) = mean neighbour distance */
int oracle_prep_sor(const float* xyz, int n, int stride, int mean_k, double std_mul, unsigned char* keep, float* distances,
                    double* threshold_out) {
  std::vector<float> dist((size_t)n, 0.f);
  if (n > mean_k) {
    std::vector<int> idx; std::vector<float> d2;
    knn_all(xyz, n, stride, mean_k + 1, idx, d2);
    for (int i = 0; i < n; i++) {
      double s = 0;
      for (int m = 1; m <= mean_k; m++) s += (double)std::sqrt(d2[(size_t)i * (mean_k + 1) + m]); /* float sqrt */
      dist[(size_t)i] = (float)(s / mean_k);
    }
  }
  double sum = 0, sq = 0;
  for (int c0 = 0; c0 < n; c0 += 64) {
    double ps = 0, pq = 0;
    for (int i = c0; i < std::min(n, c0 + 64); i++) { ps += (double)dist[(size_t)i]; pq += (double)dist[(size_t)i] * (double)dist[(size_t)i]; }
    sum += ps; sq += pq;
  }
  const double mean = sum / (double)n;
  const double variance = (sq - sum * sum / (double)n) / ((double)n - 1);
  const double stddev = ppf_sqrt(variance);
  const double thr = mean + std_mul * stddev;
  for (int i = 0; i < n; i++) keep[i] = !((double)dist[(size_t)i] > thr);
  if (distances) memcpy(distances, dist.data(), (size_t)n * sizeof(float));
  if (threshold_out) *threshold_out = thr;
  return 0;
}

/* pcl::NormalEstimation(k), viewpoint (0,0,0): normals n x 3, curvature n */
int oracle_prep_normals(const float* xyz, int n, int stride, int k, float* normals, float* curvature) {
  const int ke = std::min(k, n);
  std::vector<int> idx; std::vector<float> d2;
  knn_all(xyz, n, stride, ke, idx, d2);
  const float qnan = std::nanf("");
  for (int i = 0; i < n; i++) {
    float* no = normals + (size_t)i * 3;
    if (ke < 3) { no[0] = no[1] = no[2] = qnan; curvature[i] = qnan; continue; }
    double c[3] = {0, 0, 0};
    for (int m = 0; m < ke; m++) { const float* q = xyz + (size_t)idx[(size_t)i * ke + m] * stride; c[0] += (double)q[0]; c[1] += (double)q[1]; c[2] += (double)q[2]; }
    for (int a = 0; a < 3; a++) c[a] /= (double)ke;
    double cov[6] = {0, 0, 0, 0, 0, 0}; /* xx xy xz yy yz zz */
    for (int m = 0; m < ke; m++) {
      const float* q = xyz + (size_t)idx[(size_t)i * ke + m] * stride;
      const double d[3] = {(double)q[0] - c[0], (double)q[1] - c[1], (double)q[2] - c[2]};
      cov[0] += d[0] * d[0]; cov[1] += d[0] * d[1]; cov[2] += d[0] * d[2];
      cov[3] += d[1] * d[1]; cov[4] += d[1] * d[2]; cov[5] += d[2] * d[2];
    }
    for (int a = 0; a < 6; a++) cov[a] /= (double)ke;
    double A[3][3] = {{cov[0], cov[1], cov[2]}, {cov[1], cov[3], cov[4]}, {cov[2], cov[4], cov[5]}}, V[3][3];
    const double trace = cov[0] + cov[3] + cov[5];
    jacobi3(A, V);
    int best = 0;
    for (int a = 1; a < 3; a++) if (A[a][a] < A[best][best]) best = a;
    double nv[3] = {V[0][best], V[1][best], V[2][best]};
    const double len = ppf_sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
    for (int a = 0; a < 3; a++) nv[a] /= len;
    const float* p = xyz + (size_t)i * stride;
    const double cos_theta = -((double)p[0] * nv[0] + (double)p[1] * nv[1] + (double)p[2] * nv[2]); /* (vp - p) . n, vp = 0 */
    if (cos_theta < 0) { nv[0] = -nv[0]; nv[1] = -nv[1]; nv[2] = -nv[2]; }
    no[0] = (float)nv[0]; no[1] = (float)nv[1]; no[2] = (float)nv[2];
    double lam = A[best][best];
    if (lam < 0) lam = -lam;
    const double at = trace < 0 ? -trace : trace;
    curvature[i] = trace != 0.0 ? (float)(lam / at) : 0.f;
  }
  return 0;
}

/* PointCloudXYZNormalToMat (:163-190): rows x y z n/|n| */
int oracle_prep_to_mat(const float* xyz, const float* normals, int n, float* rows6) {
  for (int i = 0; i < n; i++) {
    float* d = rows6 + (size_t)i * 6;
    d[0] = xyz[(size_t)i * 3]; d[1] = xyz[(size_t)i * 3 + 1]; d[2] = xyz[(size_t)i * 3 + 2];
    d[3] = normals[(size_t)i * 3]; d[4] = normals[(size_t)i * 3 + 1]; d[5] = normals[(size_t)i * 3 + 2];
    const float s = d[3] * d[3] + d[4] * d[4] + d[5] * d[5];
    const double A = (double)std::sqrt(s);
    if (A > 0.00001) { d[3] /= (float)A; d[4] /= (float)A; d[5] /= (float)A; }
  }
  return 0;
}
}
