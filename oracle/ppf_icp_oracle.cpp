/*
 * ppf_icp_oracle.cpp — CPU ORACLE for the ICP refinement step.  TEST INFRASTRUCTURE ONLY (same rules as
 * ppf_oracle.cpp: loaded by tests/, smoke() and bench.py's cpu_baseline leg, never by the product).
 *
 * WHAT IT RESTATES
 *   ICP icp(100, 0.005f, 2.5f, 8); icp.registerModelToScene(models[id], scene, resultsSub);
 *     /root/reference/include/CloudProcessing.h:465-470 and :518-523
 *   i.e. cv::ppf_match_3d::ICP of the un-vendored OpenCV-contrib surface_matching module (icp.cpp): a
 *   multi-resolution point-to-plane ICP ("Picky ICP" correspondences, robust rejection by median + MAD,
 *   linearised 6-DoF least squares per iteration), run for each of the top poses of the PPF match.
 *
 * PARITY STATUS: **parity unpinned** (no fixtures, no binary; see ppf_oracle.cpp).  Restated from knowledge of the
 * public source.  Choices frozen here where upstream depends on library internals:
 *   - nearest neighbour: upstream queries a FLANN kd-tree (exact search, float L2).  Here: exhaustive search,
 *     squared distance ((dx*dx + dy*dy) + dz*dz) in float, scene points in index order, first minimum wins.
 *   - least squares: upstream cv::solve(A, b, DECOMP_SVD) on the n x 6 system (minimum-norm solution when the
 *     correspondences leave directions unconstrained).  Here: normal equations (A^T A) x = A^T b in fp64, rows
 *     accumulated in chunks of 64 (sequential inside a chunk, chunk sums added sequentially) -- the order the device
 *     kernels reproduce -- with Tikhonov damping 1e-10 * trace, solved by Gaussian elimination (solve6).
 *   - median: element of rank (m-1)/2 of the sorted values (lower median).
 *   - a level stops when picky ICP leaves SIX OR FEWER correspondences (`sel.size() <= 6`).  Upstream only tests for zero
 *     (`if (selInd)` in ICP::registerModelToScene) and would hand a 1..6-row system to cv::solve(DECOMP_SVD), whose
 *     minimum-norm step moves the model along the few constrained directions; with fewer rows than unknowns that step
 *     carries no information about the pose, so this restatement ends the level instead (the pose of the level stays what it
 *     was).  It is the one place where an iteration COUNT can differ from an upstream build.
 *
 * WHY REFINED POSES CAN COME BACK WITH RESIDUAL 9999999999 AFTER ONE TO THREE ITERATIONS (profiles/r03_icp_levels.json,
 * tools/icp_levels.py: the per-pass trace of this oracle on the top-5 matched poses of the C2 crop): `residual` is upstream's
 * fval_min of the LAST (finest) level, which starts at 9999999999 and is only lowered by a completed iteration.  The reference
 * refines the five best clusters of the match (CloudProcessing.h:455-470); on a crop with one instance, clusters 2..5 are
 * wrong poses.  Picky ICP -- upstream's "if more than one model point is assigned to the same scene point keep the closest"
 * (the duplicateTable / hashtable pass of ICP::registerModelToScene) -- is hard on the coarse levels: level 7 pairs 155
 * model rows with 391 scene rows spread over the whole crop, and keeps 12-13 pairs EVEN FOR THE CORRECT POSE (which then
 * improves level by level: 26, 56, 99, ... 858 pairs, 19 iterations, residual 1.5e-3).  From a wrong pose those dozen pairs
 * do not describe one surface, the 6-DoF step they give throws the model off the data, after which every model point has
 * the same nearest scene point, the rejection threshold (distances all alike) accepts all of them and picky ICP keeps ONE
 * pair (trace: accepted 19,753, kept 1, at every remaining level).  The level ends (rule above), the finest level never
 * completes an iteration, and the sentinel comes back with the diverged pose.  An upstream build loses the same
 * correspondences by the same rule; what it would do differently is keep taking minimum-norm steps on the one remaining pair
 * and report that pair's residual.  Neither result is a usable pose: a residual of 9999999999 is the honest label.
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/ppf_detmath.h" /* deterministic sin/cos/sqrt: the spec the device side reproduces bit for bit */

namespace {

inline long cv_round(double v) { return std::lrint(v); } /* cvRound: round half to even */

struct Cloud { std::vector<float> d; int rows() const { return (int)(d.size() / 6); } float* row(int i) { return &d[(size_t)i * 6]; } const float* row(int i) const { return &d[(size_t)i * 6]; } };

/* transformPCPose (ppf_helpers.cpp): points through the 4x4 (with homogeneous divide), normals through R, re-normalised */
Cloud transformPCPose(const Cloud& pc, const double* T) {
  Cloud out; out.d.resize(pc.d.size());
  for (int i = 0; i < pc.rows(); i++) {
    const float* p = pc.row(i); float* o = out.row(i);
    double v[4];
    for (int r = 0; r < 4; r++) v[r] = T[r * 4] * p[0] + T[r * 4 + 1] * p[1] + T[r * 4 + 2] * p[2] + T[r * 4 + 3];
    if (std::fabs(v[3]) > 1.192092896e-07) { v[0] /= v[3]; v[1] /= v[3]; v[2] /= v[3]; }
    o[0] = (float)v[0]; o[1] = (float)v[1]; o[2] = (float)v[2];
    double nn[3];
    for (int r = 0; r < 3; r++) nn[r] = T[r * 4] * p[3] + T[r * 4 + 1] * p[4] + T[r * 4 + 2] * p[5];
    const double nrm = ppf_sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
    if (nrm > 1.192092896e-07) { nn[0] /= nrm; nn[1] /= nrm; nn[2] /= nrm; }
    o[3] = (float)nn[0]; o[4] = (float)nn[1]; o[5] = (float)nn[2];
  }
  return out;
}
Cloud sampleUniform(const Cloud& pc, int step) {
  Cloud out;
  for (int i = 0; i < pc.rows(); i += step) out.d.insert(out.d.end(), pc.row(i), pc.row(i) + 6);
  return out;
}
void mat44mul(const double* A, const double* B, double* C) {
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { double s = 0; for (int k = 0; k < 4; k++) s += A[i * 4 + k] * B[k * 4 + j]; C[i * 4 + j] = s; }
}
float lower_median(std::vector<float> v) {
  const size_t k = (v.size() - 1) / 2;
  std::nth_element(v.begin(), v.begin() + k, v.end());
  return v[k];
}
/* getRejectionThreshold: median + scale * 1.48257968 * MAD */
float rejection_threshold(const std::vector<float>& r, float scale) {
  const float med = lower_median(r);
  std::vector<float> t(r.size());
  for (size_t i = 0; i < r.size(); i++) t[i] = (float)std::fabs((double)r[i] - (double)med);
  const float s = 1.48257968f * lower_median(t);
  return scale * s + med;
}
/* eulerToDCM + getTransformMat (c_utils.hpp): R = Rz(yaw) * Ry(pitch) * Rx(roll), euler = (roll, pitch, yaw) */
void transform_from_euler(const double* e, const double* t, double* P) {
  const double cx = ppf_cos(e[0]), sx = ppf_sin(e[0]), cy = ppf_cos(e[1]), sy = ppf_sin(e[1]), cz = ppf_cos(e[2]), sz = ppf_sin(e[2]);
  const double Rx[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx}, Ry[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy}, Rz[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1};
  double T1[9], R[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += Ry[i * 3 + k] * Rx[k * 3 + j]; T1[i * 3 + j] = s; }
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += Rz[i * 3 + k] * T1[k * 3 + j]; R[i * 3 + j] = s; }
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) P[i * 4 + j] = R[i * 3 + j]; P[i * 4 + 3] = t[i]; }
  P[12] = P[13] = P[14] = 0; P[15] = 1;
}
/* Least-squares solution of the 6x6 normal equations M[:, :6] x = M[:, 6], robust to directions the correspondences
 * do not constrain (sliding on a plane, spinning about an axis of symmetry), where upstream's cv::solve(DECOMP_SVD)
 * returns the minimum-norm solution: Tikhonov damping (M + lambda I) x = b with lambda = 1e-10 * trace(M) -- identical
 * to the plain solution for constrained directions (relative change 1e-10 * condition number), ~0 along unconstrained
 * ones -- then Gaussian elimination with partial pivoting.  Result in M[i][6]; false when trace(M) is not positive. */
bool solve6(double M[6][7]) {
  double trace = 0;
  for (int i = 0; i < 6; i++) trace += M[i][i];
  if (!(trace > 0.0)) return false;
  const double lambda = 1e-10 * trace;
  for (int i = 0; i < 6; i++) M[i][i] += lambda;
  for (int c = 0; c < 6; c++) {
    int piv = c;
    for (int r = c + 1; r < 6; r++) if (std::fabs(M[r][c]) > std::fabs(M[piv][c])) piv = r;
    if (std::fabs(M[piv][c]) < 1e-300) return false;
    if (piv != c) for (int k = 0; k < 7; k++) std::swap(M[c][k], M[piv][k]);
    for (int r = c + 1; r < 6; r++) {
      const double f = M[r][c] / M[c][c];
      for (int k = c; k < 7; k++) M[r][k] -= f * M[c][k];
    }
  }
  for (int c = 5; c >= 0; c--) {
    double s = M[c][6];
    for (int k = c + 1; k < 6; k++) s -= M[c][k] * M[k][6];
    M[c][6] = s / M[c][c];
  }
  return true;
}

struct IcpParams { int iterations; float tolerance, rejection_scale; int num_levels; };

/* optional trace (oracle_icp_set_trace): one row {level, iteration, source rows, scene rows, accepted by the rejection
 * threshold, kept by picky ICP, exit code} per pass of the while loop -- exit 0: iterated, 1: six or fewer correspondences,
 * 2: the 6x6 solve failed, 3: NaN in the solution -- how tools/icp_levels.py explains a pose's iteration count and residual */
int* g_trace = nullptr;
int g_trace_cap = 0, g_trace_n = 0;
void trace_row(int level, int it, int ns, int nd, int acc, int sel, int code) {
  if (!g_trace || g_trace_n >= g_trace_cap) return;
  int* r = g_trace + (size_t)g_trace_n * 7;
  r[0] = level; r[1] = it; r[2] = ns; r[3] = nd; r[4] = acc; r[5] = sel; r[6] = code;
  g_trace_n++;
}

/* ICP::registerModelToScene(srcPC, dstPC, residual, pose) */
int icp_single(const Cloud& srcPC, const Cloud& dstPC, const IcpParams& prm, double* pose, double* residual, int* iters_total) {
  const int n = srcPC.rows();
  const bool robust = prm.rejection_scale > 0;
  Cloud srcTemp = srcPC, dstTemp = dstPC;
  /* sums over points: chunks of 64 rows summed sequentially, chunk sums added sequentially */
  auto chunk_sum3 = [](const Cloud& c, double* out) {
    out[0] = out[1] = out[2] = 0;
    for (int c0 = 0; c0 < c.rows(); c0 += 64) {
      double p[3] = {0, 0, 0};
      for (int i = c0; i < std::min(c.rows(), c0 + 64); i++) for (int k = 0; k < 3; k++) p[k] += (double)c.row(i)[k];
      for (int k = 0; k < 3; k++) out[k] += p[k];
    }
  };
  double meanSrc[3], meanDst[3];
  chunk_sum3(srcTemp, meanSrc);
  chunk_sum3(dstTemp, meanDst);
  double meanAvg[3];
  for (int k = 0; k < 3; k++) { meanSrc[k] /= n; meanDst[k] /= dstTemp.rows(); meanAvg[k] = 0.5 * (meanSrc[k] + meanDst[k]); }
  auto center = [&](Cloud& c) { for (int i = 0; i < c.rows(); i++) for (int k = 0; k < 3; k++) c.row(i)[k] = (float)((double)c.row(i)[k] - meanAvg[k]); };
  center(srcTemp); center(dstTemp);
  auto dist_sum = [](const Cloud& c) {
    double d = 0;
    for (int c0 = 0; c0 < c.rows(); c0 += 64) {
      double part = 0;
      for (int i = c0; i < std::min(c.rows(), c0 + 64); i++) {
        const float* p = c.row(i);
        part += ppf_sqrt((double)p[0] * p[0] + (double)p[1] * p[1] + (double)p[2] * p[2]);
      }
      d += part;
    }
    return d;
  };
  const double scale = (double)n / ((dist_sum(srcTemp) + dist_sum(dstTemp)) * 0.5);
  auto rescale = [&](Cloud& c) { for (int i = 0; i < c.rows(); i++) for (int k = 0; k < 3; k++) c.row(i)[k] = (float)((double)c.row(i)[k] * scale); };
  rescale(srcTemp); rescale(dstTemp);
  const Cloud& srcPC0 = srcTemp; const Cloud& dstPC0 = dstTemp;
  for (int k = 0; k < 16; k++) pose[k] = (k % 5 == 0) ? 1.0 : 0.0;
  double fval_min = 9999999999.0;
  int total = 0;
  for (int level = prm.num_levels - 1; level >= 0; level--) {
    const double div = std::pow(2.0, (double)level);
    const int numSamples = (int)cv_round((double)n / div);
    const double TolP = (double)prm.tolerance * (double)(level + 1) * (level + 1);
    const int maxIter = (int)cv_round((double)prm.iterations / (level + 1));
    Cloud srcPCT = transformPCPose(srcPC0, pose);
    const int sampleStep = std::max(1, (int)cv_round((double)n / (double)std::max(numSamples, 1)));
    srcPCT = sampleUniform(srcPCT, sampleStep);
    const Cloud dstPCS = sampleUniform(dstPC0, sampleStep);
    double fval_old = 9999999999.0, fval_perc = 0;
    fval_min = 9999999999.0;
    Cloud moved = srcPCT;
    const int ns = srcPCT.rows(), nd = dstPCS.rows();
    double PoseX[16];
    for (int k = 0; k < 16; k++) PoseX[k] = (k % 5 == 0) ? 1.0 : 0.0;
    std::vector<int> nn(ns);
    std::vector<float> dist(ns);
    int i = 0;
    while (!(fval_perc < (1 + TolP) && fval_perc > (1 - TolP)) && i < maxIter) {
#pragma omp parallel for schedule(static) num_threads(16) if ((long)ns * nd > 4000000L) /* rows are independent: threads change nothing in the result; small levels stay serial (a parallel region per iteration costs more than it saves, badly so on boxes whose CPU share is smaller than their core count) */
      for (int a = 0; a < ns; a++) { /* exhaustive NN, float squared distance, first minimum */
        const float* p = moved.row(a);
        float best = 3.402823466e+38f; int bi = 0;
        for (int b = 0; b < nd; b++) {
          const float* q = dstPCS.row(b);
          const float dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
          const float d2 = (dx * dx + dy * dy) + dz * dz;
          if (d2 < best) { best = d2; bi = b; }
        }
        nn[a] = bi; dist[a] = best;
      }
      std::vector<int> accI;
      if (robust) {
        const float thr = rejection_threshold(dist, prm.rejection_scale);
        for (int a = 0; a < ns; a++) if (dist[a] < thr) accI.push_back(a);
      } else {
        for (int a = 0; a < ns; a++) accI.push_back(a);
      }
      /* Picky ICP: a scene point keeps only its closest model point (ties: the smallest model index) */
      std::vector<int> owner(nd, -1);
      for (int a : accI) { const int b = nn[a]; if (owner[b] < 0 || dist[a] < dist[owner[b]]) owner[b] = a; }
      std::vector<std::pair<int, int>> sel; /* (model row, scene row), ordered by scene row */
      for (int b = 0; b < nd; b++) if (owner[b] >= 0) sel.push_back({owner[b], b});
      if ((int)sel.size() <= 6) { trace_row(level, i, ns, nd, (int)accI.size(), (int)sel.size(), 1); break; }
      /* minimizePointToPlaneMetric on the level's UNMOVED source rows; normal equations in chunk order */
      double M[6][7]; memset(M, 0, sizeof(M));
      double fsum = 0;
      for (size_t c0 = 0; c0 < sel.size(); c0 += 64) {
        double P[6][7]; memset(P, 0, sizeof(P));
        double fs = 0;
        for (size_t k = c0; k < std::min(sel.size(), c0 + 64); k++) {
          const float* s = srcPCT.row(sel[k].first); const float* d = dstPCS.row(sel[k].second);
          const double sp[3] = {s[0], s[1], s[2]}, dp[3] = {d[0], d[1], d[2]}, nr[3] = {d[3], d[4], d[5]};
          const double sub[3] = {dp[0] - sp[0], dp[1] - sp[1], dp[2] - sp[2]};
          const double ax[3] = {sp[1] * nr[2] - sp[2] * nr[1], sp[2] * nr[0] - sp[0] * nr[2], sp[0] * nr[1] - sp[1] * nr[0]};
          const double rowA[6] = {ax[0], ax[1], ax[2], nr[0], nr[1], nr[2]};
          const double bb = sub[0] * nr[0] + sub[1] * nr[1] + sub[2] * nr[2];
          for (int r = 0; r < 6; r++) { for (int cc = 0; cc < 6; cc++) P[r][cc] += rowA[r] * rowA[cc]; P[r][6] += rowA[r] * bb; }
          double e = 0; /* norm(Src_Match - Dst_Match)^2 over all 6 columns: per-row subtotal, then into the chunk sum */
          for (int cc = 0; cc < 6; cc++) { const double df = (double)s[cc] - (double)d[cc]; e += df * df; }
          fs += e;
        }
        for (int r = 0; r < 6; r++) for (int cc = 0; cc < 7; cc++) M[r][cc] += P[r][cc];
        fsum += fs;
      }
      if (!solve6(M)) { trace_row(level, i, ns, nd, (int)accI.size(), (int)sel.size(), 2); break; }
      const double rpy[3] = {M[0][6], M[1][6], M[2][6]}, t[3] = {M[3][6], M[4][6], M[5][6]};
      if (rpy[0] != rpy[0] || rpy[1] != rpy[1] || rpy[2] != rpy[2] || t[0] != t[0] || t[1] != t[1] || t[2] != t[2]) { trace_row(level, i, ns, nd, (int)accI.size(), (int)sel.size(), 3); break; }
      transform_from_euler(rpy, t, PoseX);
      moved = transformPCPose(srcPCT, PoseX);
      const double fval = ppf_sqrt(fsum) / (double)ns;
      fval_perc = fval / fval_old;
      fval_old = fval;
      if (fval < fval_min) fval_min = fval;
      trace_row(level, i, ns, nd, (int)accI.size(), (int)sel.size(), 0);
      i++;
    }
    total += i;
    double tmp[16];
    mat44mul(PoseX, pose, tmp);
    memcpy(pose, tmp, sizeof(tmp));
  }
  /* undo centring and scaling: t = t/scale + meanAvg - R*meanAvg */
  double Rm[3];
  for (int r = 0; r < 3; r++) Rm[r] = pose[r * 4] * meanAvg[0] + pose[r * 4 + 1] * meanAvg[1] + pose[r * 4 + 2] * meanAvg[2];
  for (int r = 0; r < 3; r++) pose[r * 4 + 3] = pose[r * 4 + 3] / scale + meanAvg[r] - Rm[r];
  *residual = fval_min;
  if (iters_total) *iters_total = total;
  return 0;
}

}  // namespace

extern "C" {
/* rows of 7 ints per pass of the ICP loop are written to buf from now on (NULL switches the trace off); returns the rows
 * written since the previous call */
int oracle_icp_set_trace(int* buf, int cap_rows) {
  const int n = g_trace_n;
  g_trace = buf; g_trace_cap = buf ? cap_rows : 0; g_trace_n = 0;
  return n;
}
/* ICP::registerModelToScene(model, scene, poses): for each initial pose, move the model, refine, append.
 * poses_io: n_poses x 16 doubles (row-major 4x4), updated in place to poseICP * pose; residuals: n_poses. */
int oracle_icp_refine(const float* model, int n_model, const float* scene, int n_scene, int iterations, float tolerance,
                      float rejection_scale, int num_levels, double* poses_io, double* residuals, int n_poses, int* iters) {
  Cloud M, S;
  M.d.assign(model, model + (size_t)n_model * 6);
  S.d.assign(scene, scene + (size_t)n_scene * 6);
  IcpParams prm{iterations, tolerance, rejection_scale, num_levels};
  for (int k = 0; k < n_poses; k++) {
    double* P = poses_io + (size_t)k * 16;
    Cloud moved = transformPCPose(M, P);
    double picp[16], res = 0; int it = 0;
    icp_single(moved, S, prm, picp, &res, &it);
    double out[16];
    mat44mul(picp, P, out);
    memcpy(P, out, sizeof(out));
    if (residuals) residuals[k] = res;
    if (iters) iters[k] = it;
  }
  return 0;
}
}
