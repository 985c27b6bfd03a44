/*
 * ppf_icp_kernels.h — ICP refinement of the matched poses on gfx950 (the step right after the
 * path; `ICP icp(100, 0.005f, 2.5f, 8); icp.registerModelToScene(models[id], scene, resultsSub);`
 * /root/reference/include/CloudProcessing.h:465-470 and :518-523).  Included by ppf_hip.hip.
 *
 * The arithmetic is the one oracle/ppf_icp_oracle.cpp freezes (multi-level point-to-plane ICP, picky
 * correspondences, median + MAD rejection), so poses, residuals and iteration counts are bit-identical to it:
 *   - nearest neighbour: exhaustive, float squared distance ((dx*dx + dy*dy) + dz*dz), first minimum in scene
 *     order.  One thread per model point, the scene slice is read through uniform (scalar) loads, slices of one
 *     point's search run in different workgroups and meet in a 64-bit atomicMin on (distance bits, scene index):
 *     distances are >= +0, so their bit patterns order like the floats and the minimum key IS "first minimum".
 *   - rejection threshold: lower median of the distances and of |d - median| by a 4-pass radix select on the float
 *     bits (no sort), one workgroup.
 *   - picky ownership: atomicMin on (distance bits, model index) per scene point, then an ordered compaction by
 *     scene index (one workgroup, thread-contiguous ranges + block scan).
 *   - normal equations: one wave per chunk of 64 correspondences; every lane builds its row, 28 lanes each add one
 *     entry of the symmetric 6x7 system (+ the residual) over the chunk IN ROW ORDER; the single-wave solve kernel
 *     adds the chunk sums IN CHUNK ORDER, solves the 6x6 (Tikhonov damping 1e-10*trace + Gaussian elimination:
 *     ~0 along directions the correspondences leave free, as upstream's SVD solve gives), builds
 *     PoseX = T(t) * Rz*Ry*Rx and updates the loop state.  fp64 throughout, no FMA contraction.
 *   - the loop state (PoseX, fval_old/perc/min, iteration counter, done flag) lives in HBM; every kernel starts with
 *     `if (st->done) return`, so the host enqueues iterations in batches and reads the flag once per batch.
 * All of it is latency/launch-bound except the NN search (ns*nd distance evaluations per iteration, VALU-bound:
 * 9 VALU ops per pair, scene points arrive in SGPRs).
 */
#ifndef PPF_ICP_KERNELS_H
#define PPF_ICP_KERNELS_H

struct IcpState {
  double T[16];     /* transform applied when a level starts (pose so far); also the initial pose */
  double PoseX[16]; /* the level's incremental pose */
  double mean_avg[3];
  double scale;
  double fval_old, fval_perc, fval_min, tol_p;
  float thr;
  int n_sel, iter, max_iter, done, robust;
};

constexpr int ICP_CHUNK = 64;
constexpr int ICP_ENTRIES = 28; /* 21 upper-triangle + 6 right-hand side + residual */
constexpr unsigned long long ICP_NONE = ~0ull;
constexpr uint32_t ICP_FLT_MAX_BITS = 0x7f7fffffu;

/* one row through a 4x4 (homogeneous divide) and its rotation block, normal re-normalised: transformPCPose */
__device__ __forceinline__ void icp_transform_row(const float* __restrict__ p, const float* __restrict__ pn, const double* __restrict__ T, float* __restrict__ o) {
  double v[4];
#pragma unroll
  for (int r = 0; r < 4; r++) v[r] = T[r * 4] * (double)p[0] + T[r * 4 + 1] * (double)p[1] + T[r * 4 + 2] * (double)p[2] + T[r * 4 + 3];
  if (ppf_fabs(v[3]) > PPF_EPS) { v[0] /= v[3]; v[1] /= v[3]; v[2] /= v[3]; }
  o[0] = (float)v[0]; o[1] = (float)v[1]; o[2] = (float)v[2];
  double nn[3];
#pragma unroll
  for (int r = 0; r < 3; r++) nn[r] = T[r * 4] * (double)pn[0] + T[r * 4 + 1] * (double)pn[1] + T[r * 4 + 2] * (double)pn[2];
  const double nrm = ppf_sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
  if (nrm > PPF_EPS) { nn[0] /= nrm; nn[1] /= nrm; nn[2] /= nrm; }
  o[3] = (float)nn[0]; o[4] = (float)nn[1]; o[5] = (float)nn[2];
}

/* out[i] = T * src[i*step] (transformPCPose followed by samplePCUniform), optional second copy, optional reset of
 * the NN keys of the rows written.  T is read from device memory; `st` (optional) gates on the done flag. */
__global__ __launch_bounds__(256) void k_icp_transform(const float* __restrict__ src, int stride, int noff, int step, int n_out,
                                                       const double* __restrict__ T, float* __restrict__ out,
                                                       float* __restrict__ out2, unsigned long long* __restrict__ best,
                                                       const IcpState* __restrict__ st) {
  if (st && st->done) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  double M[16];
#pragma unroll
  for (int k = 0; k < 16; k++) M[k] = T[k];
  float o[6];
  icp_transform_row(src + (size_t)i * step * stride, src + (size_t)i * step * stride + noff, M, o);
#pragma unroll
  for (int k = 0; k < 6; k++) {
    out[(size_t)i * 6 + k] = o[k];
    if (out2) out2[(size_t)i * 6 + k] = o[k];
  }
  if (best) best[i] = (unsigned long long)ICP_FLT_MAX_BITS << 32;
}

/* plain strided copy into packed rows (samplePCUniform without a transform) + float4 xyz pack for the NN search */
__global__ __launch_bounds__(256) void k_icp_sample(const float* __restrict__ src, int stride, int noff, int step, int n_out,
                                                    float* __restrict__ out, float4* __restrict__ q4) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  const float* p = src + (size_t)i * step * stride;
#pragma unroll
  for (int k = 0; k < 3; k++) { out[(size_t)i * 6 + k] = p[k]; out[(size_t)i * 6 + 3 + k] = p[noff + k]; }
  if (q4) q4[i] = make_float4(p[0], p[1], p[2], 0.f);
}

/* per-chunk sums (chunks of 64 rows, rows added sequentially): mode 0 -> xyz, mode 1 -> |xyz| */
__global__ __launch_bounds__(64) void k_icp_chunk_sums(const float* __restrict__ c, int n, int mode, double* __restrict__ parts) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int c0 = k * ICP_CHUNK;
  if (c0 >= n) return;
  const int c1 = min(n, c0 + ICP_CHUNK);
  double s[3] = {0, 0, 0};
  for (int i = c0; i < c1; i++) {
    const float* p = c + (size_t)i * 6;
    if (mode == 0) {
      s[0] += (double)p[0]; s[1] += (double)p[1]; s[2] += (double)p[2];
    } else {
      s[0] += ppf_sqrt((double)p[0] * p[0] + (double)p[1] * p[1] + (double)p[2] * p[2]);
    }
  }
  parts[(size_t)k * 3] = s[0]; parts[(size_t)k * 3 + 1] = s[1]; parts[(size_t)k * 3 + 2] = s[2];
}

/* sequential sum of chunk partials, loads issued 8 at a time */
__device__ __forceinline__ double icp_sum_parts(const double* __restrict__ parts, int n_chunks, int pitch) {
  double acc = 0;
  int c = 0;
  for (; c + 8 <= n_chunks; c += 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = parts[(size_t)(c + u) * pitch];
#pragma unroll
    for (int u = 0; u < 8; u++) acc += v[u];
  }
  for (; c < n_chunks; c++) acc += parts[(size_t)c * pitch];
  return acc;
}

/* mode 0: mean_avg = 0.5*(mean(src)+mean(dst)); mode 1: scale = n_src / (0.5*(sum|src| + sum|dst|)) */
__global__ __launch_bounds__(64) void k_icp_reduce(const double* __restrict__ parts_src, int n_src, const double* __restrict__ parts_dst,
                                                   int n_dst, int mode, IcpState* __restrict__ st) {
  __shared__ double tot[6];
  const int tid = threadIdx.x;
  if (tid < 6) {
    const bool is_dst = tid >= 3;
    const int n = is_dst ? n_dst : n_src;
    tot[tid] = icp_sum_parts((is_dst ? parts_dst : parts_src) + (tid % 3), (n + ICP_CHUNK - 1) / ICP_CHUNK, 3);
  }
  __syncthreads();
  if (tid == 0) {
    if (mode == 0) {
      for (int k = 0; k < 3; k++) {
        const double ms = tot[k] / (double)n_src, md = tot[3 + k] / (double)n_dst;
        st->mean_avg[k] = 0.5 * (ms + md);
      }
    } else {
      st->scale = (double)n_src / ((tot[0] + tot[3]) * 0.5);
    }
  }
}

/* mode 0: xyz = (float)(xyz - mean_avg); mode 1: xyz = (float)(xyz * scale) */
__global__ __launch_bounds__(256) void k_icp_center_scale(float* __restrict__ c, int n, int mode, const IcpState* __restrict__ st) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float* p = c + (size_t)i * 6;
#pragma unroll
  for (int k = 0; k < 3; k++) p[k] = mode == 0 ? (float)((double)p[k] - st->mean_avg[k]) : (float)((double)p[k] * st->scale);
}

struct IcpMat44 {
  double m[16];
};
/* state->T = T (the pose applied when a level starts / the initial pose): by kernel argument, so that several
 * registrations can be enqueued on different streams without pageable host copies serialising them */
__global__ void k_icp_set_pose(IcpState* __restrict__ st, IcpMat44 T) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  for (int k = 0; k < 16; k++) st->T[k] = T.m[k];
}

__global__ void k_icp_level_init(IcpState* __restrict__ st, double tol_p, int max_iter, int robust) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  for (int k = 0; k < 16; k++) st->PoseX[k] = (k % 5 == 0) ? 1.0 : 0.0;
  st->fval_old = 9999999999.0;
  st->fval_perc = 0;
  st->fval_min = 9999999999.0;
  st->tol_p = tol_p;
  st->iter = 0;
  st->max_iter = max_iter;
  st->n_sel = 0;
  st->robust = robust;
  st->thr = 0.f;
  const double fp = 0.0;
  st->done = (!(fp < (1 + tol_p) && fp > (1 - tol_p)) && 0 < max_iter) ? 0 : 1;
}

/* exhaustive nearest neighbour: thread = model point, blockIdx.y = slice of the scene */
__global__ __launch_bounds__(256) void k_icp_nn(const float* __restrict__ moved, int ns, const float4* __restrict__ q4, int nd,
                                                int slice, unsigned long long* __restrict__ best, const IcpState* __restrict__ st) {
  if (st->done) return;
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  const int b0 = blockIdx.y * slice, b1 = min(nd, b0 + slice);
  float px = 0.f, py = 0.f, pz = 0.f;
  if (a < ns) { px = moved[(size_t)a * 6]; py = moved[(size_t)a * 6 + 1]; pz = moved[(size_t)a * 6 + 2]; }
  float bd = 3.402823466e+38f;
  int bi = -1;
#pragma unroll 4
  for (int b = b0; b < b1; b++) {
    const float4 q = q4[b]; /* b is wave-uniform: scalar load */
    const float dx = px - q.x, dy = py - q.y, dz = pz - q.z;
    const float d2 = (dx * dx + dy * dy) + dz * dz;
    if (d2 < bd) { bd = d2; bi = b; }
  }
  if (a < ns && bi >= 0) atomicMin(&best[a], ((unsigned long long)__float_as_uint(bd) << 32) | (unsigned)bi);
}

/* k-th smallest (rank from 0) of n non-negative floats given by their bit patterns; all threads get the result */
template <class F>
__device__ uint32_t icp_block_select(F val, int n, uint32_t rank, uint32_t* hist, uint32_t* sh) {
  const int tid = threadIdx.x;
  uint32_t prefix = 0;
  for (int shift = 24; shift >= 0; shift -= 8) {
    for (int k = tid; k < 256; k += blockDim.x) hist[k] = 0;
    __syncthreads();
    const uint32_t mask = shift == 24 ? 0u : (0xFFFFFFFFu << (shift + 8));
    for (int i = tid; i < n; i += blockDim.x) {
      const uint32_t v = val(i);
      if ((v & mask) == prefix) atomicAdd(&hist[(v >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid < 64) { /* 4 bins per lane, wave scan, the lane whose range holds `rank` picks the bin */
      uint32_t c[4], s = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) { c[k] = hist[tid * 4 + k]; s += c[k]; }
      uint32_t incl = s;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
        if (tid >= o) incl += up;
      }
      uint32_t excl = incl - s;
      if (rank >= excl && rank < incl) {
        int b = 0;
        for (; b < 3; b++) { if (rank < excl + c[b]) break; excl += c[b]; }
        sh[0] = prefix | ((uint32_t)(tid * 4 + b) << shift);
        sh[1] = rank - excl;
      }
    }
    __syncthreads();
    prefix = sh[0];
    rank = sh[1];
    __syncthreads();
  }
  return prefix;
}

/* getRejectionThreshold: median + scale * 1.48257968 * MAD; also clears the ownership keys of the scene points.
 * The eight selection passes read the distances from LDS (dynamic, ns*4 bytes, staged once) when `staged`. */
__global__ __launch_bounds__(1024) void k_icp_threshold(const unsigned long long* __restrict__ best, int ns, float rej_scale,
                                                        unsigned long long* __restrict__ owner, int nd, int staged,
                                                        IcpState* __restrict__ st) {
  if (st->done) return;
  extern __shared__ uint32_t s_bits[];
  __shared__ uint32_t hist[256];
  __shared__ uint32_t sh[2];
  const int tid = threadIdx.x;
  for (int b = tid; b < nd; b += blockDim.x) owner[b] = ICP_NONE;
  if (!st->robust) return;
  if (staged) {
    for (int i = tid; i < ns; i += blockDim.x) s_bits[i] = (uint32_t)(best[i] >> 32);
    __syncthreads();
  }
  const uint32_t rank = (uint32_t)((ns - 1) / 2);
  auto dist_bits = [&](int i) { return staged ? s_bits[i] : (uint32_t)(best[i] >> 32); };
  const uint32_t med_bits = icp_block_select(dist_bits, ns, rank, hist, sh);
  const float med = __uint_as_float(med_bits);
  const uint32_t mad_bits = icp_block_select(
      [&](int i) { return __float_as_uint((float)ppf_fabs((double)__uint_as_float(dist_bits(i)) - (double)med)); }, ns, rank, hist, sh);
  if (tid == 0) {
    const float s = 1.48257968f * __uint_as_float(mad_bits);
    st->thr = rej_scale * s + med;
  }
}

/* picky ICP: every scene point keeps the closest of the model points that chose it (ties: smallest model index) */
__global__ __launch_bounds__(256) void k_icp_owner(const unsigned long long* __restrict__ best, int ns,
                                                   unsigned long long* __restrict__ owner, const IcpState* __restrict__ st) {
  if (st->done) return;
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= ns) return;
  const unsigned long long key = best[a];
  const uint32_t dbits = (uint32_t)(key >> 32), b = (uint32_t)key;
  if (st->robust && !(__uint_as_float(dbits) < st->thr)) return;
  atomicMin(&owner[b], ((unsigned long long)dbits << 32) | (unsigned)a);
}

/* ordered compaction of the owned scene points: sel[k] = (model row, scene row), ascending scene row */
__global__ __launch_bounds__(1024) void k_icp_compact(const unsigned long long* __restrict__ owner, int nd, int2* __restrict__ sel,
                                                      IcpState* __restrict__ st) {
  if (st->done) return;
  __shared__ uint32_t wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (nd + 1023) / 1024;
  const int b0 = min(nd, tid * per), b1 = min(nd, b0 + per);
  uint32_t cnt = 0;
  for (int b = b0; b < b1; b++) cnt += owner[b] != ICP_NONE;
  uint32_t incl = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
    if (lane >= o) incl += up;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  uint32_t base = 0, total = 0;
  for (int w = 0; w < 16; w++) {
    if (w < wave) base += wsum[w];
    total += wsum[w];
  }
  uint32_t pos = base + incl - cnt;
  for (int b = b0; b < b1; b++) {
    const unsigned long long o = owner[b];
    if (o != ICP_NONE) sel[pos++] = make_int2((int)(uint32_t)o, b);
  }
  if (tid == 0) {
    st->n_sel = (int)total;
    if (total <= 6) st->done = 1; /* `if (selInd <= 6) break;` */
  }
}

/* one wave per chunk of 64 correspondences: rows in parallel, the 28 sums in row order */
__global__ __launch_bounds__(64) void k_icp_chunks(const int2* __restrict__ sel, const float* __restrict__ src_pct,
                                                   const float* __restrict__ dst_pcs, double* __restrict__ parts,
                                                   const IcpState* __restrict__ st) {
  if (st->done) return;
  const int n_sel = st->n_sel;
  const int c0 = blockIdx.x * ICP_CHUNK;
  if (c0 >= n_sel) return;
  const int rows = min(ICP_CHUNK, n_sel - c0);
  __shared__ double val[ICP_CHUNK][9]; /* rowA[0..5], b, e, 1 */
  const int tid = threadIdx.x;
  if (tid < rows) {
    const int2 ab = sel[c0 + tid];
    const float* s = src_pct + (size_t)ab.x * 6;
    const float* d = dst_pcs + (size_t)ab.y * 6;
    const double sp[3] = {(double)s[0], (double)s[1], (double)s[2]}, dp[3] = {(double)d[0], (double)d[1], (double)d[2]},
                 nr[3] = {(double)d[3], (double)d[4], (double)d[5]};
    const double sub[3] = {dp[0] - sp[0], dp[1] - sp[1], dp[2] - sp[2]};
    val[tid][0] = sp[1] * nr[2] - sp[2] * nr[1];
    val[tid][1] = sp[2] * nr[0] - sp[0] * nr[2];
    val[tid][2] = sp[0] * nr[1] - sp[1] * nr[0];
    val[tid][3] = nr[0]; val[tid][4] = nr[1]; val[tid][5] = nr[2];
    val[tid][6] = sub[0] * nr[0] + sub[1] * nr[1] + sub[2] * nr[2];
    double e = 0;
#pragma unroll
    for (int cc = 0; cc < 6; cc++) { const double df = (double)s[cc] - (double)d[cc]; e += df * df; }
    val[tid][7] = e;
    val[tid][8] = 1.0;
  }
  __syncthreads();
  if (tid < ICP_ENTRIES) {
    int i = 0, j = 0;
    if (tid < 21) { /* upper triangle, row-major */
      int t = tid;
      while (t >= 6 - i) { t -= 6 - i; i++; }
      j = i + t;
    } else if (tid < 27) { i = tid - 21; j = 6; }
    else { i = 7; j = 8; }
    double acc = 0;
    for (int k = 0; k < rows; k++) acc += val[k][i] * val[k][j];
    parts[(size_t)blockIdx.x * ICP_ENTRIES + tid] = acc;
  }
}

/* eulerToDCM + getTransformMat: R = Rz(e2) * Ry(e1) * Rx(e0) */
__device__ void icp_transform_from_euler(const double* e, const double* t, double* P) {
  const double cx = ppf_cos(e[0]), sx = ppf_sin(e[0]), cy = ppf_cos(e[1]), sy = ppf_sin(e[1]), cz = ppf_cos(e[2]), sz = ppf_sin(e[2]);
  const double Rx[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx}, Ry[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy}, Rz[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1};
  double T1[9], R[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += Ry[i * 3 + k] * Rx[k * 3 + j]; T1[i * 3 + j] = s; }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += Rz[i * 3 + k] * T1[k * 3 + j]; R[i * 3 + j] = s; }
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) P[i * 4 + j] = R[i * 3 + j]; P[i * 4 + 3] = t[i]; }
  P[12] = P[13] = P[14] = 0; P[15] = 1;
}

__global__ __launch_bounds__(64) void k_icp_solve(const double* __restrict__ parts, int ns, IcpState* __restrict__ st) {
  if (st->done) return;
  __shared__ double tot[ICP_ENTRIES];
  __shared__ double M[6][7];
  const int tid = threadIdx.x;
  const int n_chunks = (st->n_sel + ICP_CHUNK - 1) / ICP_CHUNK;
  if (tid < ICP_ENTRIES) tot[tid] = icp_sum_parts(parts + tid, n_chunks, ICP_ENTRIES);
  __syncthreads();
  if (tid != 0) return;
  int e = 0;
  for (int i = 0; i < 6; i++)
    for (int j = i; j < 6; j++) { M[i][j] = tot[e]; M[j][i] = tot[e]; e++; }
  for (int i = 0; i < 6; i++) M[i][6] = tot[21 + i];
  const double fsum = tot[27];
  /* damped normal equations (M + 1e-10 trace I) x = b, Gaussian elimination with partial pivoting (oracle: solve6) */
  double trace = 0;
  for (int i = 0; i < 6; i++) trace += M[i][i];
  if (!(trace > 0.0)) { st->done = 1; return; }
  const double lambda = 1e-10 * trace;
  for (int i = 0; i < 6; i++) M[i][i] += lambda;
  bool ok = true;
  for (int c = 0; c < 6 && ok; c++) {
    int piv = c;
    for (int r = c + 1; r < 6; r++) if (ppf_fabs(M[r][c]) > ppf_fabs(M[piv][c])) piv = r;
    if (ppf_fabs(M[piv][c]) < 1e-300) { ok = false; break; }
    if (piv != c) for (int k = 0; k < 7; k++) { const double tmp = M[c][k]; M[c][k] = M[piv][k]; M[piv][k] = tmp; }
    for (int r = c + 1; r < 6; r++) {
      const double f = M[r][c] / M[c][c];
      for (int k = c; k < 7; k++) M[r][k] -= f * M[c][k];
    }
  }
  if (!ok) { st->done = 1; return; }
  for (int c = 5; c >= 0; c--) {
    double sacc = M[c][6];
    for (int k = c + 1; k < 6; k++) sacc -= M[c][k] * M[k][6];
    M[c][6] = sacc / M[c][c];
  }
  const double rpy[3] = {M[0][6], M[1][6], M[2][6]}, t[3] = {M[3][6], M[4][6], M[5][6]};
  if (rpy[0] != rpy[0] || rpy[1] != rpy[1] || rpy[2] != rpy[2] || t[0] != t[0] || t[1] != t[1] || t[2] != t[2]) { st->done = 1; return; }
  double P[16];
  icp_transform_from_euler(rpy, t, P);
  for (int k = 0; k < 16; k++) st->PoseX[k] = P[k];
  const double fval = ppf_sqrt(fsum) / (double)ns;
  const double perc = fval / st->fval_old;
  st->fval_perc = perc;
  st->fval_old = fval;
  if (fval < st->fval_min) st->fval_min = fval;
  const int it = st->iter + 1;
  st->iter = it;
  const double tp = st->tol_p;
  st->done = (!(perc < (1 + tp) && perc > (1 - tp)) && it < st->max_iter) ? 0 : 1;
}

/* ---- coarse levels: the whole level in ONE workgroup ---------------------------------------------------------
 * The pyramid's coarse levels hold a few hundred to ~2000 model rows; seven launches and a host round trip per
 * iteration cost far more than their arithmetic.  For ns <= ICP_SMALL_NS one 1024-thread workgroup runs the level's
 * complete loop: the same steps with the same arithmetic and orders as the kernels above (NN search in registers,
 * radix-select threshold over LDS, ownership by global atomicMin, ordered compaction, chunk sums by waves, solve by
 * thread 0), separated by workgroup barriers instead of kernel boundaries, until the loop condition fails.  One
 * launch per level, no intermediate read-back; results are bit-identical to the multi-kernel path. */
constexpr int ICP_SMALL_NS = 2048;
constexpr int ICP_SMALL_VAL = 4; /* chunk row buffers (waves working on chunk sums at a time) */

/* ownership keys are produced by global atomics of this workgroup's other waves: read them at the coherence point */
__device__ __forceinline__ unsigned long long icp_ld(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(1024) void k_icp_level_small(const float* __restrict__ src_pct, int ns, const float4* __restrict__ q4,
                                                          const float* __restrict__ dst_pcs, int nd, unsigned long long* __restrict__ owner,
                                                          float rej_scale, IcpState* __restrict__ st) {
  __shared__ float s_dist[ICP_SMALL_NS];
  __shared__ int s_nn[ICP_SMALL_NS];
  __shared__ int2 s_sel[ICP_SMALL_NS];
  __shared__ double s_val[ICP_SMALL_VAL][ICP_CHUNK][9];
  __shared__ double s_parts[ICP_SMALL_NS / ICP_CHUNK][ICP_ENTRIES];
  __shared__ double s_pose[16], s_tot[ICP_ENTRIES], s_M[6][7];
  __shared__ double s_fval_old, s_fval_perc, s_fval_min;
  __shared__ uint32_t hist[256], sh[2], wsum[16];
  __shared__ float s_thr;
  __shared__ int s_done, s_iter, s_nsel;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int robust = st->robust, max_iter = st->max_iter;
  const double tol_p = st->tol_p;
  if (tid == 0) {
    s_done = st->done; s_iter = st->iter; s_nsel = 0;
    s_fval_old = st->fval_old; s_fval_perc = st->fval_perc; s_fval_min = st->fval_min;
    for (int k = 0; k < 16; k++) s_pose[k] = st->PoseX[k];
  }
  __syncthreads();
  while (!s_done) {
    /* 1. nearest neighbours: up to two model points per thread, one pass over the scene (uniform index: scalar loads) */
    float px[2] = {0.f, 0.f}, py[2] = {0.f, 0.f}, pz[2] = {0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int a = tid + u * 1024;
      if (a < ns) {
        const float* p = src_pct + (size_t)a * 6;
        double v[4];
#pragma unroll
        for (int r = 0; r < 4; r++)
          v[r] = s_pose[r * 4] * (double)p[0] + s_pose[r * 4 + 1] * (double)p[1] + s_pose[r * 4 + 2] * (double)p[2] + s_pose[r * 4 + 3];
        if (ppf_fabs(v[3]) > PPF_EPS) { v[0] /= v[3]; v[1] /= v[3]; v[2] /= v[3]; }
        px[u] = (float)v[0]; py[u] = (float)v[1]; pz[u] = (float)v[2];
      }
    }
    float bd[2] = {3.402823466e+38f, 3.402823466e+38f};
    int bi[2] = {0, 0};
    for (int b = 0; b < nd; b++) {
      const float4 q = q4[b];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const float dx = px[u] - q.x, dy = py[u] - q.y, dz = pz[u] - q.z;
        const float d2 = (dx * dx + dy * dy) + dz * dz;
        if (d2 < bd[u]) { bd[u] = d2; bi[u] = b; }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int a = tid + u * 1024;
      if (a < ns) { s_dist[a] = bd[u]; s_nn[a] = bi[u]; }
    }
    for (int b = tid; b < nd; b += 1024) __hip_atomic_store(&owner[b], ICP_NONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    /* 2. rejection threshold */
    if (robust) {
      const uint32_t rank = (uint32_t)((ns - 1) / 2);
      const uint32_t med_bits = icp_block_select([&](int i) { return __float_as_uint(s_dist[i]); }, ns, rank, hist, sh);
      const float med = __uint_as_float(med_bits);
      const uint32_t mad_bits = icp_block_select(
          [&](int i) { return __float_as_uint((float)ppf_fabs((double)s_dist[i] - (double)med)); }, ns, rank, hist, sh);
      if (tid == 0) {
        const float sc = 1.48257968f * __uint_as_float(mad_bits);
        s_thr = rej_scale * sc + med;
      }
      __syncthreads();
    }
    /* 3. picky ownership */
    for (int a = tid; a < ns; a += 1024) {
      const float d = s_dist[a];
      if (!robust || d < s_thr) atomicMin(&owner[s_nn[a]], ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)a);
    }
    __syncthreads();
    /* 4. ordered compaction */
    {
      const int per = (nd + 1023) / 1024;
      const int b0 = min(nd, tid * per), b1 = min(nd, b0 + per);
      uint32_t cnt = 0;
      for (int b = b0; b < b1; b++) cnt += icp_ld(&owner[b]) != ICP_NONE;
      uint32_t incl = cnt;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
        if (lane >= o) incl += up;
      }
      if (lane == 63) wsum[wave] = incl;
      __syncthreads();
      uint32_t base = 0, total = 0;
      for (int w = 0; w < 16; w++) {
        if (w < wave) base += wsum[w];
        total += wsum[w];
      }
      uint32_t pos = base + incl - cnt;
      for (int b = b0; b < b1; b++) {
        const unsigned long long o = icp_ld(&owner[b]);
        if (o != ICP_NONE) s_sel[pos++] = make_int2((int)(uint32_t)o, b);
      }
      if (tid == 0) {
        s_nsel = (int)total;
        if (total <= 6) s_done = 1; /* `if (selInd <= 6) break;` */
      }
      __syncthreads();
    }
    if (s_done) break;
    /* 5. chunk sums: waves 0..ICP_SMALL_VAL-1 take the chunks round-robin */
    const int n_sel = s_nsel;
    const int n_chunks = (n_sel + ICP_CHUNK - 1) / ICP_CHUNK;
    if (wave < ICP_SMALL_VAL) {
      double (*val)[9] = s_val[wave];
      for (int c = wave; c < n_chunks; c += ICP_SMALL_VAL) {
        const int c0 = c * ICP_CHUNK, rows = min(ICP_CHUNK, n_sel - c0);
        if (lane < rows) {
          const int2 ab = s_sel[c0 + lane];
          const float* sp_ = src_pct + (size_t)ab.x * 6;
          const float* d = dst_pcs + (size_t)ab.y * 6;
          const double sp[3] = {(double)sp_[0], (double)sp_[1], (double)sp_[2]}, dp[3] = {(double)d[0], (double)d[1], (double)d[2]},
                       nr[3] = {(double)d[3], (double)d[4], (double)d[5]};
          const double sub[3] = {dp[0] - sp[0], dp[1] - sp[1], dp[2] - sp[2]};
          val[lane][0] = sp[1] * nr[2] - sp[2] * nr[1];
          val[lane][1] = sp[2] * nr[0] - sp[0] * nr[2];
          val[lane][2] = sp[0] * nr[1] - sp[1] * nr[0];
          val[lane][3] = nr[0]; val[lane][4] = nr[1]; val[lane][5] = nr[2];
          val[lane][6] = sub[0] * nr[0] + sub[1] * nr[1] + sub[2] * nr[2];
          double e = 0;
#pragma unroll
          for (int cc = 0; cc < 6; cc++) { const double df = (double)sp_[cc] - (double)d[cc]; e += df * df; }
          val[lane][7] = e;
          val[lane][8] = 1.0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane < ICP_ENTRIES) {
          int i = 0, j = 0;
          if (lane < 21) {
            int t = lane;
            while (t >= 6 - i) { t -= 6 - i; i++; }
            j = i + t;
          } else if (lane < 27) { i = lane - 21; j = 6; }
          else { i = 7; j = 8; }
          double acc = 0;
          for (int k = 0; k < rows; k++) acc += val[k][i] * val[k][j];
          s_parts[c][lane] = acc;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
    __syncthreads();
    /* 6. chunk sums in chunk order, solve, loop state */
    if (tid < ICP_ENTRIES) {
      double acc = 0;
      for (int c = 0; c < n_chunks; c++) acc += s_parts[c][tid];
      s_tot[tid] = acc;
    }
    __syncthreads();
    if (tid == 0) {
      int e = 0;
      for (int i = 0; i < 6; i++)
        for (int j = i; j < 6; j++) { s_M[i][j] = s_tot[e]; s_M[j][i] = s_tot[e]; e++; }
      for (int i = 0; i < 6; i++) s_M[i][6] = s_tot[21 + i];
      const double fsum = s_tot[27];
      double trace = 0;
      for (int i = 0; i < 6; i++) trace += s_M[i][i];
      bool ok = trace > 0.0;
      if (ok) {
        const double lambda = 1e-10 * trace;
        for (int i = 0; i < 6; i++) s_M[i][i] += lambda;
        for (int c = 0; c < 6 && ok; c++) {
          int piv = c;
          for (int r = c + 1; r < 6; r++) if (ppf_fabs(s_M[r][c]) > ppf_fabs(s_M[piv][c])) piv = r;
          if (ppf_fabs(s_M[piv][c]) < 1e-300) { ok = false; break; }
          if (piv != c) for (int k = 0; k < 7; k++) { const double tmp = s_M[c][k]; s_M[c][k] = s_M[piv][k]; s_M[piv][k] = tmp; }
          for (int r = c + 1; r < 6; r++) {
            const double f = s_M[r][c] / s_M[c][c];
            for (int k = c; k < 7; k++) s_M[r][k] -= f * s_M[c][k];
          }
        }
      }
      if (ok) {
        for (int c = 5; c >= 0; c--) {
          double sacc = s_M[c][6];
          for (int k = c + 1; k < 6; k++) sacc -= s_M[c][k] * s_M[k][6];
          s_M[c][6] = sacc / s_M[c][c];
        }
        const double rpy[3] = {s_M[0][6], s_M[1][6], s_M[2][6]}, t[3] = {s_M[3][6], s_M[4][6], s_M[5][6]};
        if (rpy[0] != rpy[0] || rpy[1] != rpy[1] || rpy[2] != rpy[2] || t[0] != t[0] || t[1] != t[1] || t[2] != t[2]) ok = false;
        if (ok) {
          double P[16];
          icp_transform_from_euler(rpy, t, P);
          for (int k = 0; k < 16; k++) s_pose[k] = P[k];
          const double fval = ppf_sqrt(fsum) / (double)ns;
          const double perc = fval / s_fval_old;
          s_fval_perc = perc;
          s_fval_old = fval;
          if (fval < s_fval_min) s_fval_min = fval;
          const int it = s_iter + 1;
          s_iter = it;
          s_done = (!(perc < (1 + tol_p) && perc > (1 - tol_p)) && it < max_iter) ? 0 : 1;
        }
      }
      if (!ok) s_done = 1;
    }
    __syncthreads();
  }
  if (tid == 0) {
    for (int k = 0; k < 16; k++) st->PoseX[k] = s_pose[k];
    st->fval_old = s_fval_old; st->fval_perc = s_fval_perc; st->fval_min = s_fval_min;
    st->iter = s_iter; st->n_sel = s_nsel; st->done = 1;
  }
}

#endif /* PPF_ICP_KERNELS_H */
