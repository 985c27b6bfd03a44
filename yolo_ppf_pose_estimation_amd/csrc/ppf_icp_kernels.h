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


/* ============================================================================================================
 * Batched path (the default): every pose of a call is one "job"; all jobs advance in lock-step through the SAME
 * launches (blockIdx.y or blockIdx.x = job), so a call costs the launches of one registration whatever the number of
 * poses, and an iteration is TWO launches instead of seven:
 *   k_icp2_nn    exact nearest neighbours, one WAVE per model point.  Fine levels search a two-level 4x4x4 grid over the
 *                job's scene (64 nodes, 4,096 leaves, boxes = the real extent of the points inside): lane l tests node l,
 *                passing nodes are opened in turn (lane l tests child l), passing leaves are scanned 64 points at a
 *                time.  A box is skipped only when its lower bound, shrunk by 1e-4, still exceeds the best distance found,
 *                so the result is the exhaustive search's (smallest float d2, then smallest scene index) -- the key
 *                (distance bits, index) is min-reduced whatever the visiting order.  Coarse levels (few scene rows) scan
 *                them all, 64 per step.  A wave takes eight rows in turn and then sends their picky-ownership keys:
 *                atomicMin of (distance bits, model row) on the scene row each one chose, rows of the wave that chose the same
 *                scene row combined first (a model thrown off the data sends ALL its rows to one scene row).  The rejection
 *                threshold is not known yet and need not be: the smallest key of a scene row passes the threshold if
 *                any key of that row does.
 *   k_icp2_tail  one workgroup per job: rejection threshold (radix select over LDS), ordered compaction of the scene rows
 *                whose owner passes it (clearing the keys as it goes), chunk sums, 6x6 solve (one column per lane), loop
 *                state; when the level ends it folds PoseX into the job's pose.  Same arithmetic and orders as the
 *                kernels above.
 * The host reads one flag per job (pinned memory, written by k_icp2_tail) after every batch of iterations.
 * ============================================================================================================ */
constexpr int ICP_MAX_JOBS = 8;     /* poses refined per batch of launches */
constexpr int ICP_LEAVES = 4096;    /* 16 x 16 x 16 cells, grouped 4 x 4 x 4 under 64 nodes */
constexpr int ICP_BRUTE_ND = 1024;  /* levels with at most this many scene rows scan them all (16 steps of a wave; C1 level 2, 2,599 rows: 58 us scanning them all, 25 us through the grid) */
constexpr float ICP_LB_SHRINK = 0.9999f;

struct IcpState2 {
  double pose[16];  /* product of the finished levels' PoseX */
  double PoseX[16]; /* the running level's incremental pose */
  double mean_avg[3];
  double scale;
  double fval_old, fval_perc, fval_min, tol_p;
  float thr;
  float org[3], inv_h; /* grid: cell = (int)((p - org) * inv_h) clamped to 0..15 */
  int n_sel, iter, max_iter, done, robust, total, pad0;
  float raw_lo[3], raw_hi[3];    /* extent of the scene rows as given (NaN left out) */
#ifdef PPF_ICP_CLOCKS
  unsigned long long ph[8];      /* diagnostic build: ticks of the 100 MHz clock k_icp2_tail's thread 0 spent per phase */
#endif
};
#ifdef PPF_ICP_CLOCKS
/* phase clocks of the tail (100 MHz): kept in registers and added to the state once, at the end -- a read-modify-write of the
 * state per mark put a memory round trip of thread 0 (and a late arrival at the next barrier) into every phase */
#define ICP_PH_DECL unsigned long long ph_t_ = __builtin_amdgcn_s_memrealtime(), ph_a_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define ICP_PH(st, k) do { if (threadIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); ph_a_[k] += t_ - ph_t_; ph_t_ = t_; } } while (0)
#define ICP_PH_FLUSH(st) do { if (threadIdx.x == 0) { for (int k_ = 0; k_ < 8; k_++) (st)->ph[k_] += ph_a_[k_]; } } while (0)
#else
#define ICP_PH_DECL do { } while (0)
#define ICP_PH(st, k) do { } while (0)
#define ICP_PH_FLUSH(st) do { } while (0)
#endif

/* floats as unsigned integers of the same order (atomicMin / atomicMax on floats of either sign) */
__device__ __forceinline__ uint32_t icp_f2o(float f) { const uint32_t b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float icp_o2f(uint32_t o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o); }

struct IcpBatch {
  const float* src; /* the model as given (rows of sstride floats, normal at snoff) */
  const float* dst;
  int n, sstride, snoff, nd_all, dstride, dnoff;
  /* per-job arrays: job j starts at base + j * pitch */
  float *src0, *dst0, *src_pct;       /* n*6, nd_all*6, n*6 */
  unsigned long long *best, *owner;   /* n, nd_all */
  int2* sel;                          /* min(n, nd_all) */
  double *parts, *sum_src, *sum_dst;  /* p_parts, p_sums, p_sumd */
  float* bb_parts;                    /* 6 per chunk of scene rows: its extent (pitch p_sumd * 2) */
  uint32_t* own_a;                    /* nd_all: model row of a scene row's owner, between two phases of k_icp2_tail */
  float4* g_pts;                      /* nd_all: x y z + original row index (bits) in leaf order */
  uint32_t* g_start;                  /* ICP_LEAVES + 1 (pitch ICP_LEAVES + 64) */
  uint32_t* g_cur;                    /* ICP_LEAVES: rows per leaf, then the scatter cursors */
  float4* g_box2;                     /* 2 per leaf: lo, hi */
  uint32_t* g_box1u;                  /* 8 per node: lo xyz -, hi xyz - as ordered-uint coded floats */
  IcpState2* state;
  int* h_done;                        /* pinned host memory: done flag per job */
  unsigned long long* h_ticks;        /* pinned: k_icp2_tail workgroups that have finished since the call began (the host waits on it) */
  IcpState2* h_state;                 /* pinned: a job's loop state, copied out by k_icp2_tail when a level ends */
  size_t p_sel, p_parts, p_sums, p_sumd;
  int has_init;
  double T0[ICP_MAX_JOBS][16];        /* initial poses */
};

static_assert(sizeof(IcpState2) % 8 == 0, "the state is copied to the host in 8-byte words");
/* the job's loop state to pinned host memory (one thread) */
__device__ __forceinline__ void icp_publish_state(const IcpBatch& B, int job) {
  const unsigned long long* src = reinterpret_cast<const unsigned long long*>(B.state + job);
  unsigned long long* dst = reinterpret_cast<unsigned long long*>(B.h_state + job);
  for (size_t k = 0; k < sizeof(IcpState2) / 8; k++) dst[k] = src[k];
}

/* one more finished k_icp2_tail workgroup.  `wrote`: the calling thread has stored something the host will read once it sees
 * the count (done flag, state): the count then goes out with release semantics at system scope, which orders it behind those
 * stores (waiting for the stores' own completion is not enough: the count is an atomic and travels another way).  A release
 * also writes the L2 back, so a workgroup that has nothing to show sends a plain one. */
__device__ __forceinline__ void icp_tick(const IcpBatch& B, bool wrote) {
  if (wrote) __hip_atomic_fetch_add(B.h_ticks, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  else __hip_atomic_fetch_add(B.h_ticks, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

/* Smallest value over the wave, in every lane.  Through DPP (data-parallel primitives of the VALU: a lane reads its neighbour's
 * register inside the instruction), not through __shfl: the shuffles of this compiler are ds_bpermute_b32, LDS instructions with
 * a hundred cycles of latency each, and the neighbour search makes a dozen reductions per query (88 ds_bpermute in its ISA).
 * Prefix-min inside the rows of 16 (row_shr 1, 2, 4, 8: a lane without a source keeps the identity), row 0's and row 2's last
 * lane into rows 1 and 3 (row_bcast15), row 1's into rows 2 and 3 (row_bcast31): lane 63 has it. */
__device__ __forceinline__ uint32_t icp_wave_min_u32(uint32_t v) {
#define ICP_DPP_MIN(ctrl, rows) v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, ctrl, rows, 0xf, false))
  ICP_DPP_MIN(0x111, 0xf);
  ICP_DPP_MIN(0x112, 0xf);
  ICP_DPP_MIN(0x114, 0xf);
  ICP_DPP_MIN(0x118, 0xf);
  ICP_DPP_MIN(0x142, 0xa);
  ICP_DPP_MIN(0x143, 0xc);
#undef ICP_DPP_MIN
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
/* ... of floats that are >= +0 or +inf (their bit patterns order like they do) */
__device__ __forceinline__ float icp_wave_minf(float v) { return __uint_as_float(icp_wave_min_u32(__float_as_uint(v))); }
/* ... of 64-bit keys (distance bits, index): the smallest high word, then the smallest low word among the lanes that have it */
__device__ __forceinline__ unsigned long long icp_wave_min64(unsigned long long k) {
  const uint32_t hi = (uint32_t)(k >> 32);
  const uint32_t hmin = icp_wave_min_u32(hi);
  const uint32_t lmin = icp_wave_min_u32(hi == hmin ? (uint32_t)k : 0xFFFFFFFFu);
  return ((unsigned long long)hmin << 32) | lmin;
}

/* prologue 1: src0 = T0 * src (or a copy), dst0 = a copy, and the per-chunk coordinate sums (rows of a chunk in order) */
__global__ __launch_bounds__(64) void k_icp2_pack_sums(IcpBatch B) {
  __shared__ float xyz[64][3];
  const int job = blockIdx.y, lane = threadIdx.x;
  const int chunks_src = (B.n + ICP_CHUNK - 1) / ICP_CHUNK;
  const bool is_dst = (int)blockIdx.x >= chunks_src;
  const int c = is_dst ? (int)blockIdx.x - chunks_src : (int)blockIdx.x;
  const int rows_all = is_dst ? B.nd_all : B.n;
  const int i = c * ICP_CHUNK + lane;
  float o[6] = {0, 0, 0, 0, 0, 0};
  if (i < rows_all) {
    const float* p = is_dst ? B.dst + (size_t)i * B.dstride : B.src + (size_t)i * B.sstride;
    const int noff = is_dst ? B.dnoff : B.snoff;
    if (!is_dst && B.has_init) {
      double M[16];
#pragma unroll
      for (int k = 0; k < 16; k++) M[k] = B.T0[job][k];
      icp_transform_row(p, p + noff, M, o);
    } else {
#pragma unroll
      for (int k = 0; k < 3; k++) { o[k] = p[k]; o[3 + k] = p[noff + k]; }
    }
    float* out = is_dst ? B.dst0 + ((size_t)job * B.nd_all + i) * 6 : B.src0 + ((size_t)job * B.n + i) * 6;
#pragma unroll
    for (int k = 0; k < 6; k++) out[k] = o[k];
  }
  xyz[lane][0] = o[0]; xyz[lane][1] = o[1]; xyz[lane][2] = o[2];
  __syncthreads();
  if (lane == 0) {
    const int rows = min(ICP_CHUNK, rows_all - c * ICP_CHUNK);
    double s[3] = {0, 0, 0};
    for (int k = 0; k < rows; k++) { s[0] += (double)xyz[k][0]; s[1] += (double)xyz[k][1]; s[2] += (double)xyz[k][2]; }
    double* parts = is_dst ? B.sum_dst + (size_t)job * B.p_sumd + (size_t)c * 3 : B.sum_src + (size_t)job * B.p_sums + (size_t)c * 3;
    parts[0] = s[0]; parts[1] = s[1]; parts[2] = s[2];
  }
  if (is_dst) { /* extent of the scene rows (the centring and scaling that follow are monotone: the grid's extent follows from it) */
    const bool v = i < rows_all;
    float lo[3], hi[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { lo[k] = v ? o[k] : __builtin_inff(); hi[k] = v ? o[k] : -__builtin_inff(); }
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) {
#pragma unroll
      for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], __shfl_xor(lo[k], sh)); hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], sh)); }
    }
    if (lane == 0) { /* per chunk; k_icp2_mean joins them (a thousand co-resident blocks all found the accumulator "unset" and queued their atomics on six addresses) */
      float* bb = B.bb_parts + ((size_t)job * B.p_sumd * 2 + (size_t)c * 6);
#pragma unroll
      for (int k = 0; k < 3; k++) { bb[k] = lo[k]; bb[3 + k] = hi[k]; }
    }
  }
}

/* before the prologue: the extent accumulators and the leaf counters of every job */
__global__ __launch_bounds__(256) void k_icp2_reset(IcpBatch B) {
  const int job = blockIdx.x, tid = threadIdx.x;
  uint32_t* cur = B.g_cur + (size_t)job * ICP_LEAVES;
  for (int k = tid; k < ICP_LEAVES; k += 256) cur[k] = 0u;
  for (int k = tid; k < 64 * 8; k += 256) B.g_box1u[(size_t)job * 64 * 8 + k] = (k & 4) ? icp_f2o(-__builtin_inff()) : icp_f2o(__builtin_inff());
  /* the ownership keys: all "no owner" from here on (k_icp2_tail clears the ones an iteration set) */
  unsigned long long* owner = B.owner + (size_t)job * B.nd_all;
  for (int b = blockIdx.y * 256 + tid; b < B.nd_all; b += gridDim.y * 256) owner[b] = ICP_NONE;
}

/* sums of chunk partials (3 per chunk) in chunk order: thread c < 3 of the block returns the sum of component c.  The partials
 * go through LDS a tile at a time, loaded by the whole block (one thread reading them from memory one after the other is a
 * chain of a thousand load latencies).  Called by every thread of the block. */
constexpr int ICP_SUM_TILE = 1024; /* chunks per tile */
__device__ __forceinline__ double icp_sum_staged(const double* __restrict__ parts, int chunks, double* lds, int tid, int nthreads) {
  double acc = 0;
  for (int c0 = 0; c0 < chunks; c0 += ICP_SUM_TILE) {
    const int cn = min(ICP_SUM_TILE, chunks - c0);
    __syncthreads();
    for (int k = tid; k < cn * 3; k += nthreads) lds[k] = parts[(size_t)c0 * 3 + k];
    __syncthreads();
    if (tid < 3)
      for (int c = 0; c < cn; c++) acc += lds[c * 3 + tid];
  }
  return acc;
}

/* prologue 2: mean_avg = 0.5 * (mean(src0) + mean(dst0)) */
__global__ __launch_bounds__(256) void k_icp2_mean(IcpBatch B) {
  __shared__ double m_parts[ICP_SUM_TILE * 3];
  __shared__ double tot[6];
  const int job = blockIdx.x, tid = threadIdx.x;
  const double a_src = icp_sum_staged(B.sum_src + (size_t)job * B.p_sums, (B.n + ICP_CHUNK - 1) / ICP_CHUNK, m_parts, tid, 256);
  const double a_dst = icp_sum_staged(B.sum_dst + (size_t)job * B.p_sumd, (B.nd_all + ICP_CHUNK - 1) / ICP_CHUNK, m_parts, tid, 256);
  if (tid < 3) { tot[tid] = a_src; tot[3 + tid] = a_dst; }
  /* the scene's extent from its chunks' */
  float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
  const int chunks_dst = (B.nd_all + ICP_CHUNK - 1) / ICP_CHUNK;
  for (int c = tid; c < chunks_dst; c += 256) {
    const float* bb = B.bb_parts + ((size_t)job * B.p_sumd * 2 + (size_t)c * 6);
#pragma unroll
    for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], bb[k]); hi[k] = fmaxf(hi[k], bb[3 + k]); }
  }
#pragma unroll
  for (int sh = 32; sh > 0; sh >>= 1) {
#pragma unroll
    for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], __shfl_xor(lo[k], sh)); hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], sh)); }
  }
  __shared__ float ext[4][6];
  if ((tid & 63) == 0) { for (int k = 0; k < 3; k++) { ext[tid >> 6][k] = lo[k]; ext[tid >> 6][3 + k] = hi[k]; } }
  __syncthreads();
  if (tid == 0) {
    IcpState2* st = B.state + job;
    for (int k = 0; k < 3; k++) {
      const double ms = tot[k] / (double)B.n, md = tot[3 + k] / (double)B.nd_all;
      st->mean_avg[k] = 0.5 * (ms + md);
      st->raw_lo[k] = fminf(fminf(ext[0][k], ext[1][k]), fminf(ext[2][k], ext[3][k]));
      st->raw_hi[k] = fmaxf(fmaxf(ext[0][3 + k], ext[1][3 + k]), fmaxf(ext[2][3 + k], ext[3][3 + k]));
    }
  }
}

/* the centred row as the sequential version stores it: (float)((double)p - mean) */
__device__ __forceinline__ void icp_centred(const float* __restrict__ p, const double* __restrict__ mean, float* c) {
#pragma unroll
  for (int k = 0; k < 3; k++) c[k] = (float)((double)p[k] - mean[k]);
}

/* prologue 3: per-chunk sums of the centred rows' distances from the origin (rows of a chunk in order) */
__global__ __launch_bounds__(64) void k_icp2_dist_sums(IcpBatch B) {
  __shared__ double dist[64];
  const int job = blockIdx.y, lane = threadIdx.x;
  const int chunks_src = (B.n + ICP_CHUNK - 1) / ICP_CHUNK;
  const bool is_dst = (int)blockIdx.x >= chunks_src;
  const int c = is_dst ? (int)blockIdx.x - chunks_src : (int)blockIdx.x;
  const int rows_all = is_dst ? B.nd_all : B.n;
  const int i = c * ICP_CHUNK + lane;
  const IcpState2* st = B.state + job;
  double d = 0;
  if (i < rows_all) {
    const float* p = is_dst ? B.dst0 + ((size_t)job * B.nd_all + i) * 6 : B.src0 + ((size_t)job * B.n + i) * 6;
    float cf[3];
    icp_centred(p, st->mean_avg, cf);
    d = ppf_sqrt((double)cf[0] * (double)cf[0] + (double)cf[1] * (double)cf[1] + (double)cf[2] * (double)cf[2]);
  }
  dist[lane] = d;
  __syncthreads();
  if (lane == 0) {
    const int rows = min(ICP_CHUNK, rows_all - c * ICP_CHUNK);
    double s = 0;
    for (int k = 0; k < rows; k++) s += dist[k];
    double* parts = is_dst ? B.sum_dst + (size_t)job * B.p_sumd + (size_t)c * 3 : B.sum_src + (size_t)job * B.p_sums + (size_t)c * 3;
    parts[0] = s;
  }
}

/* hierarchical leaf id of a cell (cx, cy, cz in 0..15): node (cx/4, cy/4, cz/4) * 64 + child (cx%4, cy%4, cz%4) */
__device__ __forceinline__ int icp_leaf_id(int cx, int cy, int cz) {
  return ((((cx >> 2) * 4 + (cy >> 2)) * 4 + (cz >> 2)) << 6) | (((cx & 3) * 4 + (cy & 3)) * 4 + (cz & 3));
}
__device__ __forceinline__ int icp_cell_of(float v, float org, float inv_h) {
  const int c = (int)((v - org) * inv_h); /* NaN -> 0 */
  return min(max(c, 0), 15);
}

/* prologue 4, one block per job: scale = n / (0.5 * (sum |src0| + sum |dst0|)), the job's loop state, the grid's geometry
 * (from the extent of the rows as given, taken through the centring and scaling: both are monotone) */
__global__ __launch_bounds__(256) void k_icp2_scale(IcpBatch B) {
  __shared__ double m_parts[ICP_SUM_TILE * 3];
  __shared__ double tot[2];
  const int job = blockIdx.x, tid = threadIdx.x;
  const double a_src = icp_sum_staged(B.sum_src + (size_t)job * B.p_sums, (B.n + ICP_CHUNK - 1) / ICP_CHUNK, m_parts, tid, 256);
  const double a_dst = icp_sum_staged(B.sum_dst + (size_t)job * B.p_sumd, (B.nd_all + ICP_CHUNK - 1) / ICP_CHUNK, m_parts, tid, 256);
  if (tid == 0) { tot[0] = a_src; tot[1] = a_dst; } /* component 0: the distance sums */
  __syncthreads();
  if (tid != 0) return;
  IcpState2* st = B.state + job;
  const double scale = (double)B.n / ((tot[0] + tot[1]) * 0.5);
  st->scale = scale;
  float lo[3], hi[3];
  for (int k = 0; k < 3; k++) {
    lo[k] = (float)((double)(float)((double)st->raw_lo[k] - st->mean_avg[k]) * scale);
    hi[k] = (float)((double)(float)((double)st->raw_hi[k] - st->mean_avg[k]) * scale);
  }
  const float ext = fmaxf(fmaxf(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
  const bool ok = ext > 0.f && ext < 1e30f; /* otherwise every row falls into cell 0 of each axis: still exact, just slow */
  for (int k = 0; k < 3; k++) st->org[k] = ok ? lo[k] : 0.f;
  st->inv_h = ok ? 16.0f / (ext * 1.0001f) : 0.f;
  for (int k = 0; k < 16; k++) { st->pose[k] = (k % 5 == 0) ? 1.0 : 0.0; st->PoseX[k] = (k % 5 == 0) ? 1.0 : 0.0; }
  st->fval_old = 9999999999.0; st->fval_perc = 0; st->fval_min = 9999999999.0; st->tol_p = 0;
  st->thr = 0.f; st->n_sel = 0; st->iter = 0; st->max_iter = 0; st->done = 1; st->robust = 0; st->total = 0;
#ifdef PPF_ICP_CLOCKS
  for (int k = 0; k < 8; k++) st->ph[k] = 0;
#endif
  B.h_done[job] = 1;
  icp_publish_state(B, job);
}

/* prologue 5: the final rows (centred, scaled) of both clouds; the scene rows counted per leaf.  A block takes 8,192 rows: the
 * leaf counters are LDS counters first (a dense surface puts a thousand rows into one leaf: that many atomics on one address
 * in memory take tens of microseconds), a block then adds its non-zero ones to the job's.  Blocks past the scene's take the
 * model's rows. */
constexpr int ICP_ROWS_BLOCK = 8192;
__global__ __launch_bounds__(1024) void k_icp2_rows_count(IcpBatch B) {
  __shared__ uint32_t hist[ICP_LEAVES];
  const int job = blockIdx.y, tid = threadIdx.x;
  const IcpState2* st = B.state + job;
  const double scale = st->scale;
  const int nb_dst = (B.nd_all + ICP_ROWS_BLOCK - 1) / ICP_ROWS_BLOCK;
  if ((int)blockIdx.x >= nb_dst) {
    const int r0 = ((int)blockIdx.x - nb_dst) * ICP_ROWS_BLOCK;
    for (int u = 0; u < ICP_ROWS_BLOCK / 1024; u++) {
      const int i = r0 + u * 1024 + tid;
      if (i < B.n) {
        float* p = B.src0 + ((size_t)job * B.n + i) * 6;
        float cf[3];
        icp_centred(p, st->mean_avg, cf);
#pragma unroll
        for (int k = 0; k < 3; k++) p[k] = (float)((double)cf[k] * scale);
      }
    }
    return;
  }
  for (int k = tid; k < ICP_LEAVES; k += 1024) hist[k] = 0;
  __syncthreads();
  const int r0 = (int)blockIdx.x * ICP_ROWS_BLOCK;
  const float ox = st->org[0], oy = st->org[1], oz = st->org[2], inv_h = st->inv_h;
  for (int u = 0; u < ICP_ROWS_BLOCK / 1024; u++) {
    const int i = r0 + u * 1024 + tid;
    if (i < B.nd_all) {
      float* p = B.dst0 + ((size_t)job * B.nd_all + i) * 6;
      float cf[3], v[3];
      icp_centred(p, st->mean_avg, cf);
#pragma unroll
      for (int k = 0; k < 3; k++) { v[k] = (float)((double)cf[k] * scale); p[k] = v[k]; }
      atomicAdd(&hist[icp_leaf_id(icp_cell_of(v[0], ox, inv_h), icp_cell_of(v[1], oy, inv_h), icp_cell_of(v[2], oz, inv_h))], 1u);
    }
  }
  __syncthreads();
  uint32_t* cur = B.g_cur + (size_t)job * ICP_LEAVES;
  for (int k = tid; k < ICP_LEAVES; k += 1024) { const uint32_t c = hist[k]; if (c) atomicAdd(&cur[k], c); }
}

/* prologue 6, one workgroup per job: where every leaf starts in the leaf-ordered row list */
__global__ __launch_bounds__(1024) void k_icp2_grid_scan(IcpBatch B) {
  __shared__ uint32_t wsum[16];
  const int job = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t* cur = B.g_cur + (size_t)job * ICP_LEAVES;
  uint32_t* g_start = B.g_start + (size_t)job * (ICP_LEAVES + 64);
  uint32_t c[4], s = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) { c[k] = cur[tid * 4 + k]; s += c[k]; }
  uint32_t incl = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
    if (lane >= o) incl += up;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  uint32_t base = 0;
  for (int w = 0; w < wave; w++) base += wsum[w];
  uint32_t run = base + incl - s;
#pragma unroll
  for (int k = 0; k < 4; k++) { g_start[tid * 4 + k] = run; cur[tid * 4 + k] = run; run += c[k]; }
  if (tid == 1023) g_start[ICP_LEAVES] = run;
}

/* prologue 7: the scene rows in leaf order (any order inside a leaf), each with its row index.  Same blocks as the count: a row's
 * rank among its block's rows of the same leaf comes from the LDS counter, the block's share of the leaf from one atomic on the
 * leaf's cursor. */
__global__ __launch_bounds__(1024) void k_icp2_grid_scatter(IcpBatch B) {
  __shared__ uint32_t hist[ICP_LEAVES];
  const int job = blockIdx.y, tid = threadIdx.x;
  const IcpState2* st = B.state + job;
  for (int k = tid; k < ICP_LEAVES; k += 1024) hist[k] = 0;
  __syncthreads();
  const int r0 = (int)blockIdx.x * ICP_ROWS_BLOCK;
  const float ox = st->org[0], oy = st->org[1], oz = st->org[2], inv_h = st->inv_h;
  constexpr int U = ICP_ROWS_BLOCK / 1024;
  float x[U], y[U], z[U];
  int leaf[U];
  uint32_t rank[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int i = r0 + u * 1024 + tid;
    leaf[u] = -1;
    if (i < B.nd_all) {
      const float* p = B.dst0 + ((size_t)job * B.nd_all + i) * 6;
      x[u] = p[0]; y[u] = p[1]; z[u] = p[2];
      leaf[u] = icp_leaf_id(icp_cell_of(x[u], ox, inv_h), icp_cell_of(y[u], oy, inv_h), icp_cell_of(z[u], oz, inv_h));
      rank[u] = atomicAdd(&hist[leaf[u]], 1u);
    }
  }
  __syncthreads();
  uint32_t* cur = B.g_cur + (size_t)job * ICP_LEAVES;
  for (int k = tid; k < ICP_LEAVES; k += 1024) { const uint32_t c = hist[k]; hist[k] = c ? atomicAdd(&cur[k], c) : 0u; }
  __syncthreads();
  float4* g_pts = B.g_pts + (size_t)job * B.nd_all;
#pragma unroll
  for (int u = 0; u < U; u++)
    if (leaf[u] >= 0) g_pts[hist[leaf[u]] + rank[u]] = make_float4(x[u], y[u], z[u], __int_as_float(r0 + u * 1024 + tid));
}

/* prologue 8: the real extent of the rows of every leaf (one WAVE per leaf: a leaf on a dense surface holds a thousand rows)
 * and of every node (ordered-uint atomics on the node's box, reset by k_icp2_reset); an empty one is (+inf, -inf): its lower
 * bound is +inf */
__global__ __launch_bounds__(256) void k_icp2_grid_boxes(IcpBatch B) {
  const int job = blockIdx.y, lane = threadIdx.x & 63;
  const int leaf = blockIdx.x * 4 + (threadIdx.x >> 6); /* grid.x = ICP_LEAVES / 4 */
  const uint32_t* g_start = B.g_start + (size_t)job * (ICP_LEAVES + 64);
  const float4* g_pts = B.g_pts + (size_t)job * B.nd_all;
  float l[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, h[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
  const uint32_t s = g_start[leaf], e = g_start[leaf + 1];
  for (uint32_t q = s + (uint32_t)lane; q < e; q += 64) {
    const float4 v = g_pts[q];
    l[0] = fminf(l[0], v.x); l[1] = fminf(l[1], v.y); l[2] = fminf(l[2], v.z);
    h[0] = fmaxf(h[0], v.x); h[1] = fmaxf(h[1], v.y); h[2] = fmaxf(h[2], v.z);
  }
#pragma unroll
  for (int sh = 32; sh > 0; sh >>= 1) {
#pragma unroll
    for (int k = 0; k < 3; k++) { l[k] = fminf(l[k], __shfl_xor(l[k], sh)); h[k] = fmaxf(h[k], __shfl_xor(h[k], sh)); }
  }
  if (lane != 0) return;
  float4* box2 = B.g_box2 + (size_t)job * ICP_LEAVES * 2;
  box2[leaf * 2] = make_float4(l[0], l[1], l[2], __uint_as_float(s));         /* .w: where the leaf's rows start ... */
  box2[leaf * 2 + 1] = make_float4(h[0], h[1], h[2], __uint_as_float(e - s)); /* ... and how many: the search reads no second table */
  if (e > s) {
    uint32_t* nb = B.g_box1u + ((size_t)job * 64 + (leaf >> 6)) * 8;
#pragma unroll
    for (int k = 0; k < 3; k++) { atomicMin(&nb[k], icp_f2o(l[k])); atomicMax(&nb[4 + k], icp_f2o(h[k])); }
  }
}

/* a level starts: its source rows = pose * src0[a * step]; block 0 resets the level's loop state */
__global__ __launch_bounds__(256) void k_icp2_level_begin(IcpBatch B, int step, int ns, double tol_p, int max_iter, int robust, int last_level) {
  const int job = blockIdx.y;
  IcpState2* st = B.state + job;
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a < ns) {
    double M[16];
#pragma unroll
    for (int k = 0; k < 16; k++) M[k] = st->pose[k];
    const float* p = B.src0 + ((size_t)job * B.n + (size_t)a * step) * 6;
    float o[6];
    icp_transform_row(p, p + 3, M, o);
    float* out = B.src_pct + ((size_t)job * B.n + a) * 6;
#pragma unroll
    for (int k = 0; k < 6; k++) out[k] = o[k];
  }
  if (a == 0) {
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
    const int done = (!(fp < (1 + tol_p) && fp > (1 - tol_p)) && 0 < max_iter) ? 0 : 1;
    st->done = done;
    B.h_done[job] = done;
    if (done && last_level) icp_publish_state(B, job); /* no pass of this level will run */
  }
}

/* squared distance from q to a box (0 inside); an empty box (lo = +inf, hi = -inf) gives +inf */
__device__ __forceinline__ float icp_box_lb2(const float4 lo, const float4 hi, float qx, float qy, float qz) {
  const float dx = fmaxf(fmaxf(lo.x - qx, qx - hi.x), 0.f), dy = fmaxf(fmaxf(lo.y - qy, qy - hi.y), 0.f), dz = fmaxf(fmaxf(lo.z - qz, qz - hi.z), 0.f);
  return (dx * dx + dy * dy) + dz * dz;
}

/* exact nearest neighbour of every source row of the level and the picky-ownership keys (see the header of this section) */
constexpr int ICP_NN_ROWS = 8; /* rows a wave takes in turn, at most */
/* the jobs a pass is launched for: those that had not finished their level when the host last looked (a finished job's
 * workgroups return at once, but 20,000 of them per finished job and pass are 5 us) */
struct IcpLive { int job[ICP_MAX_JOBS]; };
__global__ __launch_bounds__(256) void k_icp2_nn(IcpBatch B, IcpLive live, int ns, int nd, int step, int step_shift, int rows, int brute_nd) {
  const int job = live.job[blockIdx.y];
  const IcpState2* st = B.state + job;
  if (st->done) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int a0 = (blockIdx.x * 4 + wave) * rows;
  if (a0 >= ns) return;
  const float* dst0 = B.dst0 + (size_t)job * B.nd_all * 6;
  const float4* __restrict__ pts = B.g_pts + (size_t)job * B.nd_all;
  const float4* __restrict__ box2 = B.g_box2 + (size_t)job * ICP_LEAVES * 2;
  const uint32_t* __restrict__ box1u = B.g_box1u + ((size_t)job * 64 + lane) * 8;
  const float4 n_lo = make_float4(icp_o2f(box1u[0]), icp_o2f(box1u[1]), icp_o2f(box1u[2]), 0.f);
  const float4 n_hi = make_float4(icp_o2f(box1u[4]), icp_o2f(box1u[5]), icp_o2f(box1u[6]), 0.f);
  const bool first_pass = st->iter == 0;
  double X[12];
#pragma unroll
  for (int k = 0; k < 12; k++) X[k] = st->PoseX[k];
  const double X3[4] = {st->PoseX[12], st->PoseX[13], st->PoseX[14], st->PoseX[15]};
  const float ox = st->org[0], oy = st->org[1], oz = st->org[2], inv_h = st->inv_h;
  unsigned long long mine = ICP_NONE; /* lane q keeps the key of the wave's q-th row */
#pragma unroll 1
  for (int q = 0; q < rows; q++) {
    const int a = a0 + q;
    if (a >= ns) break;
    float qx, qy, qz;
    {
      const float* p = B.src_pct + ((size_t)job * B.n + a) * 6;
      if (first_pass) { /* the level's first pass searches from its rows as they are (`moved = srcPCT`) */
        qx = p[0]; qy = p[1]; qz = p[2];
      } else {          /* later passes from PoseX * row: the xyz of transformPCPose */
        double v[4];
#pragma unroll
        for (int r = 0; r < 3; r++) v[r] = X[r * 4] * (double)p[0] + X[r * 4 + 1] * (double)p[1] + X[r * 4 + 2] * (double)p[2] + X[r * 4 + 3];
        v[3] = X3[0] * (double)p[0] + X3[1] * (double)p[1] + X3[2] * (double)p[2] + X3[3];
        /* the last row of a composition of rigid motions is exactly 0 0 0 1 and a division by exactly 1 changes nothing: the three
         * fp64 divisions (a seventh of this kernel's instructions) are only issued for a pose that is not one */
        if (v[3] != 1.0 && ppf_fabs(v[3]) > PPF_EPS) { v[0] /= v[3]; v[1] /= v[3]; v[2] /= v[3]; }
        qx = (float)v[0]; qy = (float)v[1]; qz = (float)v[2];
      }
    }
    unsigned long long key = (unsigned long long)ICP_FLT_MAX_BITS << 32; /* nothing closer than FLT_MAX: index 0, as the sequential loop leaves it */
    if (nd <= brute_nd) {
      for (int b = lane; b < nd; b += 64) {
        const float* pq = dst0 + (size_t)b * step * 6;
        const float dx = qx - pq[0], dy = qy - pq[1], dz = qz - pq[2];
        const float d2 = (dx * dx + dy * dy) + dz * dz;
        const unsigned long long k = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)b;
        key = k < key ? k : key;
      }
      key = icp_wave_min64(key);
    } else {
      float ubest = 3.402823466e+38f; /* wave-uniform: smallest distance found so far */
      /* one row of the leaf-ordered list against the query */
      auto take = [&](const float4 pq, const bool valid) {
        const int idx = __float_as_int(pq.w);
        int b;
        bool in;
        if (step_shift >= 0) { b = idx >> step_shift; in = (idx & (step - 1)) == 0; }
        else { b = idx / step; in = b * step == idx; }
        const float dx = qx - pq.x, dy = qy - pq.y, dz = qz - pq.z;
        const float d2 = (dx * dx + dy * dy) + dz * dz;
        const unsigned long long k = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)b;
        if (valid && in) key = k < key ? k : key;
      };
      /* the rows [s0, s0 + n0) and [s1, s1 + n1) against the query: TWO leaves per memory round trip, and four steps' loads of
       * each at once (a leaf on a dense surface holds hundreds of rows) */
      auto scan_two = [&](const uint32_t s0, const uint32_t n0, const uint32_t s1, const uint32_t n1) {
        const uint32_t nmax = n0 > n1 ? n0 : n1;
        for (uint32_t i0 = 0; i0 < nmax; i0 += 256) {
          /* up to four steps of 64 rows, their loads issued together; the steps past the longer leaf's end are not issued at all
           * (wave-uniform): a sparse scene's leaves hold a few dozen rows, and three empty steps were three quarters of a scan */
          const uint32_t rem = nmax - i0;
          float4 pa[4], pb[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            if (u == 0 || (uint32_t)(u * 64) < rem) {
              const uint32_t i = i0 + (uint32_t)(u * 64 + lane);
              pa[u] = pts[s0 + (i < n0 ? i : 0u)]; /* a leaf with rows starts inside the list; an empty one reads row s0 (clamped into the list by the caller) */
              pb[u] = pts[s1 + (i < n1 ? i : 0u)];
            }
          }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            if (u == 0 || (uint32_t)(u * 64) < rem) {
              const uint32_t i = i0 + (uint32_t)(u * 64 + lane);
              take(pa[u], i < n0);
              take(pb[u], i < n1);
            }
          }
        }
        key = icp_wave_min64(key);
        ubest = __uint_as_float((uint32_t)(key >> 32));
      };
      const uint32_t last_row = (uint32_t)B.nd_all - 1u;
      /* one node: lane l tests child l (its box arrives with the child's row range: no second table to read); `first` (>= 0) is
       * scanned whatever its bound -- the query's own leaf, which gives the search a distance to prune with --, then the
       * children nearest first, two at a time: the first one whose bound exceeds the best distance ends the list (the second of
       * a pair is scanned on the bound as it stood before the first: scanning a leaf too many costs nothing but its rows) */
      auto visit_node = [&](const int k1, const int first) {
        const int c0 = k1 * 64 + lane;
        const float4 c_lo = box2[c0 * 2], c_hi = box2[c0 * 2 + 1];
        float lb2 = icp_box_lb2(c_lo, c_hi, qx, qy, qz);
        lb2 = lb2 != lb2 ? 0.f : lb2; /* a NaN query: every box is opened */
        const int c_start = (int)min((uint32_t)__float_as_int(c_lo.w), last_row), c_count = __float_as_int(c_hi.w);
        int ka = first;
        if (first >= 0 && lane == first) lb2 = __builtin_inff();
        while (true) {
          if (ka < 0) {
            const float m2 = icp_wave_minf(lb2);
            if (m2 * ICP_LB_SHRINK > ubest) break;
            ka = __builtin_ctzll(__ballot(lb2 == m2));
            if (lane == ka) lb2 = __builtin_inff();
          }
          int kb = -1;
          {
            const float m2 = icp_wave_minf(lb2);
            if (!(m2 * ICP_LB_SHRINK > ubest)) {
              kb = __builtin_ctzll(__ballot(lb2 == m2));
              if (lane == kb) lb2 = __builtin_inff();
            }
          }
          scan_two((uint32_t)__builtin_amdgcn_readlane(c_start, ka), (uint32_t)__builtin_amdgcn_readlane(c_count, ka),
                   kb >= 0 ? (uint32_t)__builtin_amdgcn_readlane(c_start, kb) : 0u, kb >= 0 ? (uint32_t)__builtin_amdgcn_readlane(c_count, kb) : 0u);
          ka = -1;
          if (kb < 0) break;
        }
      };
      const int own = icp_leaf_id(icp_cell_of(qx, ox, inv_h), icp_cell_of(qy, oy, inv_h), icp_cell_of(qz, oz, inv_h));
      float lb1 = icp_box_lb2(n_lo, n_hi, qx, qy, qz);
      lb1 = lb1 != lb1 ? 0.f : lb1;
      visit_node(own >> 6, own & 63);
      if (lane == (own >> 6)) lb1 = __builtin_inff();
      while (true) { /* the other nodes, nearest first */
        const float m = icp_wave_minf(lb1);
        if (m * ICP_LB_SHRINK > ubest) break;
        const int k1 = __builtin_ctzll(__ballot(lb1 == m));
        if (lane == k1) lb1 = __builtin_inff();
        visit_node(k1, -1);
      }
    }
    if (lane == 0) B.best[(size_t)job * B.n + a] = key;
    if (lane == q) mine = key;
  }
  /* picky ownership: the scene row each of the wave's rows chose gets atomicMin(distance bits, model row); rows that chose the
   * same scene row send one atomic with their smallest key */
  {
    const uint32_t b_mine = (uint32_t)mine;
    const bool have = lane < rows && a0 + lane < ns;
    unsigned long long k = have ? ((mine & 0xFFFFFFFF00000000ull) | (unsigned)(a0 + lane)) : ICP_NONE;
    bool leader = have;
#pragma unroll
    for (int j = 0; j < ICP_NN_ROWS; j++) {
      const uint32_t bj = (uint32_t)__shfl((int)b_mine, j);
      const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)k, j), hi = (uint32_t)__shfl((int)(uint32_t)(k >> 32), j);
      const unsigned long long kj = ((unsigned long long)hi << 32) | lo; /* lane j's own key (it is never changed before step j) */
      const bool valid_j = j < rows && a0 + j < ns;
      if (have && valid_j && j != lane && bj == b_mine) {
        if (j < lane) leader = false;      /* an earlier lane speaks for this scene row */
        else k = kj < k ? kj : k;          /* this lane does: it takes the later lanes' keys in */
      }
    }
    /* only a key that can lower the scene row's current one is sent (a stale read only lets a redundant atomic through): when a
     * model has been thrown off the data, twenty thousand rows name the same scene row */
    if (leader) {
      unsigned long long* o = &B.owner[(size_t)job * B.nd_all + b_mine];
      if (k < __hip_atomic_load(o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(o, k);
    }
  }
}

/* the candidate with `sub` smaller ones before it among list[0 .. cand), cand <= 256: thread t < cand takes candidate t and counts
 * the others below / not above it, eight candidates per step from two 16-byte reads (one 4-byte read per step waits an LDS round
 * trip per candidate: 256 of them were 6 us).  The list is 16-byte aligned and padded with 0xFFFFFFFF to a multiple of eight
 * (a pad never counts as smaller, and only for the largest value as not larger, where the count is at its maximum anyway). */
__device__ __forceinline__ void icp_rank_candidates(const uint32_t* list, const uint32_t cand, const uint32_t sub, uint32_t* out) {
  if (threadIdx.x < cand) {
    const uint32_t v = list[threadIdx.x];
    uint32_t less = 0, leq = 0;
    for (uint32_t j = 0; j < cand; j += 8u) {
      const uint4 a = *reinterpret_cast<const uint4*>(&list[j]), b = *reinterpret_cast<const uint4*>(&list[j + 4u]);
      less += ((a.x < v ? 1u : 0u) + (a.y < v ? 1u : 0u)) + ((a.z < v ? 1u : 0u) + (a.w < v ? 1u : 0u));
      less += ((b.x < v ? 1u : 0u) + (b.y < v ? 1u : 0u)) + ((b.z < v ? 1u : 0u) + (b.w < v ? 1u : 0u));
      leq += ((a.x <= v ? 1u : 0u) + (a.y <= v ? 1u : 0u)) + ((a.z <= v ? 1u : 0u) + (a.w <= v ? 1u : 0u));
      leq += ((b.x <= v ? 1u : 0u) + (b.y <= v ? 1u : 0u)) + ((b.z <= v ? 1u : 0u) + (b.w <= v ? 1u : 0u));
    }
    if (less <= sub && sub < leq) *out = v;
  }
}
/* pads list[cand ..] up to the next multiple of eight (the list holds 264 words) */
__device__ __forceinline__ void icp_pad_candidates(uint32_t* list, const uint32_t cand) {
  if (threadIdx.x < 8u && ((cand + threadIdx.x) >> 3) == (cand >> 3) && (cand & 7u)) list[cand + threadIdx.x] = 0xFFFFFFFFu;
}

/* icp_block_select with the histogram updates of a wave combined for its most frequent digit: the distances of one level
 * share their leading byte (and, for a model thrown off the data, all their bits), and 64 lanes adding to one LDS counter
 * are 64 serial updates */
template <class F>
__device__ uint32_t icp_block_select2(F val, int n, uint32_t rank, uint32_t* hist /* 264 words, 16-byte aligned */, uint32_t* sh /* 4 words */) {
  const int tid = threadIdx.x, lane = tid & 63;
  uint32_t prefix = 0;
  uint32_t cand = (uint32_t)n; /* values that still share the prefix */
  for (int shift = 24; shift >= 0; shift -= 8) {
    const uint32_t mask = shift == 24 ? 0u : (0xFFFFFFFFu << (shift + 8));
    if (cand <= 256u) {
      /* few candidates left (a coarse level: from the start): list them, and the one with `rank` smaller ones before it is the
       * answer -- no further digit passes, each of which costs four barriers whatever it counts */
      if (tid == 0) sh[3] = 0u;
      __syncthreads();
      for (int i = tid; i < n; i += blockDim.x) {
        const uint32_t v = val(i);
        if ((v & mask) == prefix) hist[atomicAdd(&sh[3], 1u)] = v;
      }
      icp_pad_candidates(hist, cand); /* cand is exact: the pads do not meet the entries */
      __syncthreads();
      icp_rank_candidates(hist, cand, rank, &sh[0]);
      __syncthreads();
      const uint32_t r = sh[0];
      __syncthreads();
      return r;
    }
    for (int k = tid; k < 256; k += blockDim.x) hist[k] = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += blockDim.x) {
      const int i = i0 + tid;
      const uint32_t v = i < n ? val(i) : 0u;
      const bool act = i < n && (v & mask) == prefix;
      const uint32_t digit = (v >> shift) & 255u;
      const unsigned long long m = __ballot(act);
      if (m) {
        const int first = __builtin_ctzll(m);
        const uint32_t d0 = (uint32_t)__shfl((int)digit, first);
        const unsigned long long same = __ballot(act && digit == d0);
        if (lane == first) atomicAdd(&hist[d0], (uint32_t)__popcll(same));
        if (act && digit != d0) atomicAdd(&hist[digit], 1u);
      }
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
        sh[2] = c[b];
      }
    }
    __syncthreads();
    prefix = sh[0];
    rank = sh[1];
    cand = sh[2];
    __syncthreads();
  }
  return prefix;
}

/* The selection the tail uses on values that sit in LDS: ONE histogram pass over a window around the wanted rank instead of a
 * pass per byte.  A digit pass of icp_block_select2 costs the sixteen waves four barriers whatever it counts, a level's two
 * selections need three or four passes each, and on the levels of a few thousand rows that was over half of the tail's time.
 * Here every wave ranks the same 64 samples (every n/64-th value) for itself and takes the samples twelve sample ranks below
 * and above the wanted one as a window [lo, hi]: three standard deviations of where the wanted value can sit among them, so it
 * is inside but for one selection in a few hundred, and the window holds about 3/8 of the values whatever their distribution
 * (outliers that stretch the range do not thin the bins out).  One pass counts the values below the window and spreads those
 * inside over 256 counters, (v - lo) >> shift; the counter that holds the rank keeps a few dozen candidates, which are ranked
 * directly.  Every wave scans the counters for itself; three barriers in all.  The wanted value outside the window or more
 * than 256 candidates (many equal values): the byte passes take over.  u32 order, like the byte passes.  Needs
 * hist[0..255] == 0 and sh[1] == sh[3] == 0 on entry (set before the caller's last barrier) and leaves them so. */
template <class F>
__device__ uint32_t icp_block_select3(F val, const int n, const uint32_t rank, uint32_t* hist /* 264 words */, uint32_t* list /* 264 words, 16-byte aligned */, uint32_t* sh /* 8 words */) {
  const int tid = threadIdx.x, lane = tid & 63;
  if (n <= 256) { /* a coarse level: every value is a candidate */
    if (tid < n) list[tid] = val(tid);
    icp_pad_candidates(list, (uint32_t)n);
    __syncthreads();
    icp_rank_candidates(list, (uint32_t)n, rank, &sh[0]);
    __syncthreads();
    return sh[0]; /* the next write of sh[0] comes after the caller's or the next selection's barrier */
  }
  uint32_t lo = 0u, hi = 0xFFFFFFFFu;
  {
    /* the window: wave w ranks samples 4 w .. 4 w + 3 among the 64 (two ballots each) and files the two that sit at the window's
     * sample ranks; equal samples file the same value */
    const uint32_t mine = val((int)(((long long)lane * n) >> 6));
    const int rs = (int)(((unsigned long long)rank * 64ull) / (unsigned long long)n);
    const int w4 = (tid >> 6) * 4;
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint32_t pj = (uint32_t)__builtin_amdgcn_readlane((int)mine, w4 + u);
      const int less = __popcll(__ballot(mine < pj)), leq = __popcll(__ballot(mine <= pj));
      if (lane == 0) {
        if (less <= rs - 12 && rs - 12 < leq) sh[4] = pj;
        if (less <= rs + 12 && rs + 12 < leq) sh[5] = pj;
      }
    }
    __syncthreads();
    if (rs - 12 >= 0) lo = sh[4];
    if (rs + 12 <= 63) hi = sh[5];
  }
  const uint32_t range = hi - lo;
  const int shift = range > 255u ? 24 - __builtin_clz(range) : 0; /* (range >> shift) <= 255 */
  {
    uint32_t n_below = 0, n_equal = 0; /* n_equal: a window of one value (a model thrown off the data) is counted, not binned */
    for (int i0 = 0; i0 < n; i0 += blockDim.x) {
      const int i = i0 + tid;
      if (i < n) {
        const uint32_t v = val(i);
        n_below += v < lo ? 1u : 0u;
        if (range == 0u) n_equal += v == lo ? 1u : 0u;
        else if (v >= lo && v <= hi) atomicAdd(&hist[(v - lo) >> shift], 1u);
      }
    }
    /* the wave's counts, one atomic each per wave */
#define ICP_DPP_ADD(x, ctrl, rows) x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, ctrl, rows, 0xf, false)
#define ICP_WAVE_ADD(x) ICP_DPP_ADD(x, 0x111, 0xf); ICP_DPP_ADD(x, 0x112, 0xf); ICP_DPP_ADD(x, 0x114, 0xf); ICP_DPP_ADD(x, 0x118, 0xf); ICP_DPP_ADD(x, 0x142, 0xa); ICP_DPP_ADD(x, 0x143, 0xc)
    ICP_WAVE_ADD(n_below);
    ICP_WAVE_ADD(n_equal);
#undef ICP_WAVE_ADD
#undef ICP_DPP_ADD
    if (lane == 63 && n_below) atomicAdd(&sh[1], n_below);
    if (lane == 63 && n_equal) atomicAdd(&hist[0], n_equal);
  }
  __syncthreads();
  uint32_t bin, sub, cand;
  bool inside;
  {
    const uint32_t n_below = sh[1];
    const uint4 c = *reinterpret_cast<const uint4*>(&hist[lane * 4]);
    const uint32_t sum = (c.x + c.y) + (c.z + c.w);
    uint32_t incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
      if (lane >= o) incl += up;
    }
    const uint32_t in_window = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    inside = rank >= n_below && rank - n_below < in_window;
    const uint32_t r = rank - n_below;
    uint32_t excl = incl - sum;
    const bool has = inside && r >= excl && r < incl; /* one lane */
    uint32_t b = 0, cb = c.x;
    if (!(r < excl + c.x)) { excl += c.x; b = 1u; cb = c.y;
      if (!(r < excl + c.y)) { excl += c.y; b = 2u; cb = c.z;
        if (!(r < excl + c.z)) { excl += c.z; b = 3u; cb = c.w; } } }
    const unsigned long long hm = __ballot(has);
    const int src = hm ? __builtin_ctzll(hm) : 0;
    bin = (uint32_t)__builtin_amdgcn_readlane((int)((uint32_t)lane * 4u + b), src);
    sub = (uint32_t)__builtin_amdgcn_readlane((int)(r - excl), src);
    cand = (uint32_t)__builtin_amdgcn_readlane((int)cb, src);
  }
  if (inside && range == 0u) { /* every value of the window is `lo` */
    __syncthreads(); /* every wave has read the counters */
    if (tid < 256) hist[tid] = 0u;
    if (tid == 256) sh[1] = 0u;
    return lo; /* the caller's next barrier comes before the next use */
  }
  if (!inside || cand > 256u) { /* the same in every wave */
    __syncthreads(); /* every wave has read the counters */
    const uint32_t r = icp_block_select2(val, n, rank, hist, sh);
    if (tid < 256) hist[tid] = 0u;
    if (tid == 256) sh[1] = 0u;
    if (tid == 257) sh[3] = 0u;
    return r; /* the caller's next barrier comes before the next use */
  }
  for (int i0 = 0; i0 < n; i0 += blockDim.x) {
    const int i = i0 + tid;
    if (i < n) {
      const uint32_t v = val(i);
      if (v >= lo && v <= hi && (v - lo) >> shift == bin) list[atomicAdd(&sh[3], 1u)] = v;
    }
  }
  icp_pad_candidates(list, cand);
  __syncthreads();
  icp_rank_candidates(list, cand, sub, &sh[0]);
  if (tid >= 256 && tid < 512) hist[tid - 256] = 0u;
  if (tid == 512) sh[1] = 0u;
  if (tid == 513) sh[3] = 0u;
  __syncthreads();
  return sh[0];
}

/* everything of an iteration after the neighbour search, one workgroup per job (see the header of this section).
 * Dynamic LDS: max(ns * 4 when staged, 16 waves x 64 x 9 doubles of chunk rows). */
constexpr int ICP_TAIL_VAL_BYTES = 16 * ICP_CHUNK * 9 * 8;
__global__ __launch_bounds__(1024) void k_icp2_tail(IcpBatch B, IcpLive live, int ns, int nd, int step, float rej_scale, int staged, int last_level) {
  extern __shared__ __align__(16) unsigned char t_dyn[];
  __shared__ __align__(16) uint32_t hist[264], s_list[264];
  __shared__ uint32_t sh[8], wsum[16];
  __shared__ double s_tot[ICP_ENTRIES];
  __shared__ float s_thr;
  __shared__ int s_done, s_nsel;
  const int job = live.job[blockIdx.x];
  IcpState2* st = B.state + job;
  if (st->done) { /* this job's level is over: only the host's count of finished workgroups moves */
    if (threadIdx.x == 0) icp_tick(B, false);
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int robust = st->robust;
  const unsigned long long* best = B.best + (size_t)job * B.n;
  unsigned long long* owner = B.owner + (size_t)job * B.nd_all;
  int2* sel = B.sel + (size_t)job * B.p_sel;
  uint32_t* own_a = B.own_a + (size_t)job * B.nd_all;
  double* parts = B.parts + (size_t)job * B.p_parts;
  const float* src_pct = B.src_pct + (size_t)job * B.n * 6;
  const float* dst0 = B.dst0 + (size_t)job * B.nd_all * 6;
  uint32_t* s_bits = reinterpret_cast<uint32_t*>(t_dyn);
  ICP_PH_DECL;
  /* 1. rejection threshold */
  if (robust) {
    if (staged) {
      if (tid < 256) hist[tid] = 0u; /* icp_block_select3's entry state */
      if (tid == 256) sh[1] = 0u;
      if (tid == 257) sh[3] = 0u;
      for (int i0 = 0; i0 < ns; i0 += 16 * 1024) { /* sixteen loads in flight per thread */
        uint32_t v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) v[u] = reinterpret_cast<const uint2*>(best)[min(i0 + u * 1024 + tid, ns - 1)].y;
#pragma unroll
        for (int u = 0; u < 16; u++)
          if (i0 + u * 1024 + tid < ns) s_bits[i0 + u * 1024 + tid] = v[u];
      }
      __syncthreads();
    }
    const uint32_t rank = (uint32_t)((ns - 1) / 2);
    if (staged) { /* one histogram pass over a window around the rank (icp_block_select3) */
      ICP_PH(st, 6);
      const uint32_t med_bits = icp_block_select3([&](int i) { return s_bits[i]; }, ns, rank, hist, s_list, sh);
      ICP_PH(st, 7);
      const float med = __uint_as_float(med_bits);
      const uint32_t mad_bits = icp_block_select3(
          [&](int i) { return __float_as_uint((float)ppf_fabs((double)__uint_as_float(s_bits[i]) - (double)med)); }, ns, rank, hist, s_list, sh);
      if (tid == 0) {
        const float sc = 1.48257968f * __uint_as_float(mad_bits);
        s_thr = rej_scale * sc + med;
      }
    } else {
      auto dist_bits = [&](int i) { return (uint32_t)(best[i] >> 32); };
      const uint32_t med_bits = icp_block_select2(dist_bits, ns, rank, hist, sh);
      const float med = __uint_as_float(med_bits);
      const uint32_t mad_bits = icp_block_select2(
          [&](int i) { return __float_as_uint((float)ppf_fabs((double)__uint_as_float(dist_bits(i)) - (double)med)); }, ns, rank, hist, sh);
      if (tid == 0) {
        const float sc = 1.48257968f * __uint_as_float(mad_bits);
        s_thr = rej_scale * sc + med;
      }
    }
    __syncthreads();
  }
  ICP_PH(st, 0);
  /* 2. (the ownership keys were set by k_icp2_nn) */
  const float thr = robust ? s_thr : 0.f;
  __syncthreads();
  ICP_PH(st, 1);
  /* 3. ordered compaction by scene row.  The keys were produced by atomics: they are read at the coherence point, coalesced
   * (thread t reads rows t, t + 1024, ...), 16 loads in flight per thread; a byte per row in LDS says whether it has an owner,
   * and each thread then walks its own contiguous range of those bytes: only owned rows (a few hundred) are read again. */
  {
    unsigned char* s_flag = t_dyn; /* the distance bits staged above are no longer needed */
    const int cap = (int)(max(staged ? (size_t)ns * 4 : (size_t)0, (size_t)ICP_TAIL_VAL_BYTES));
    uint32_t done_rows = 0; /* rows of earlier rounds, their owners already written: sel[0 .. done_rows) */
    for (int r0 = 0; r0 < nd; r0 += cap) {
      const int rn = min(cap, nd - r0);
      __syncthreads();
      for (int b0 = 0; b0 < rn; b0 += 16 * 1024) {
        unsigned long long o[16];
#pragma unroll
        for (int u = 0; u < 16; u++) o[u] = icp_ld(&owner[r0 + min(b0 + u * 1024 + tid, rn - 1)]);
#pragma unroll
        for (int u = 0; u < 16; u++) { /* an owned row: is its owner within the threshold? which model row?  Its key is cleared for the next pass. */
          const int bb = b0 + u * 1024 + tid;
          if (bb < rn) {
            const bool owned = o[u] != ICP_NONE;
            s_flag[bb] = owned && (!robust || __uint_as_float((uint32_t)(o[u] >> 32)) < thr) ? 1 : 0;
            if (owned) {
              own_a[r0 + bb] = (uint32_t)o[u];
              __hip_atomic_store(&owner[r0 + bb], ICP_NONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
        }
      }
      __syncthreads();
      const int per = (rn + 1023) / 1024;
      const int b0 = min(rn, tid * per), b1 = min(rn, b0 + per);
      uint32_t cnt = 0;
      for (int bb = b0; bb < b1; bb++) cnt += s_flag[bb];
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
      uint32_t pos = done_rows + base + incl - cnt;
      for (int bb = b0; bb < b1; bb++)
        if (s_flag[bb]) sel[pos++] = make_int2((int)own_a[r0 + bb], r0 + bb);
      done_rows += total;
    }
    if (tid == 0) {
      s_nsel = (int)done_rows;
      s_done = done_rows <= 6 ? 1 : 0; /* `if (selInd <= 6) break;` */
    }
    __syncthreads();
  }
  const int n_sel = s_nsel;
  ICP_PH(st, 2);
  if (!s_done) {
    /* 4. chunk sums: wave w takes chunks w, w + 16, ...; rows in parallel, the 28 sums in row order */
    const int n_chunks = (n_sel + ICP_CHUNK - 1) / ICP_CHUNK;
    double (*val)[9] = reinterpret_cast<double (*)[9]>(t_dyn + (size_t)wave * ICP_CHUNK * 9 * 8);
    for (int c = wave; c < n_chunks; c += 16) {
      const int c0 = c * ICP_CHUNK, rows = min(ICP_CHUNK, n_sel - c0);
      if (lane < rows) {
        const int2 ab = sel[c0 + lane];
        const float* sp_ = src_pct + (size_t)ab.x * 6;
        const float* d = dst0 + (size_t)ab.y * step * 6;
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
        parts[(size_t)c * ICP_ENTRIES + lane] = acc;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    ICP_PH(st, 3);
    /* 5. chunk sums in chunk order, solve, loop state */
    if (tid < ICP_ENTRIES) s_tot[tid] = icp_sum_parts(parts + tid, n_chunks, ICP_ENTRIES);
    __syncthreads();
  }
  ICP_PH(st, 4);
  if (wave != 0) return;
  int done = s_done;
  if (!done) {
    /* the damped 6x6 system, one COLUMN per lane (lanes 0..5: the matrix, lane 6: the right-hand side): the row operations
     * of the elimination are the sequential solve's, element for element; the back substitution runs on lane 0 */
    double col[6];
    {
      int e = 0;
#pragma unroll
      for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = i; j < 6; j++) { if (lane == j) col[i] = s_tot[e]; if (lane == i) col[j] = s_tot[e]; e++; }
#pragma unroll
      for (int i = 0; i < 6; i++) if (lane == 6) col[i] = s_tot[21 + i];
      if (lane > 6) { for (int i = 0; i < 6; i++) col[i] = 0; }
    }
    const double fsum = s_tot[27];
    auto bcast = [&](double v, int src) {
      const int lo = __shfl(__double2loint(v), src), hi = __shfl(__double2hiint(v), src);
      return __hiloint2double(hi, lo);
    };
    double trace = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) trace += bcast(col[i], i); /* M[i][i] in order */
    bool ok = trace > 0.0;
    if (ok) {
      const double lambda = 1e-10 * trace;
#pragma unroll
      for (int i = 0; i < 6; i++) if (lane == i) col[i] += lambda;
#pragma unroll
      for (int c = 0; c < 6; c++) {
        /* pivot: the first largest |M[r][c]|, r >= c (lane c holds column c) */
        int piv = c;
        double pv = ppf_fabs(col[c]);
#pragma unroll
        for (int r = c + 1; r < 6; r++) { const double av = ppf_fabs(col[r]); if (av > pv) { pv = av; piv = r; } }
        piv = __shfl(piv, c);
        pv = bcast(pv, c);
        if (pv < 1e-300) { ok = false; break; }
#pragma unroll
        for (int r = c + 1; r < 6; r++) if (r == piv) { const double tmp = col[c]; col[c] = col[r]; col[r] = tmp; }
        const double dcc = bcast(col[c], c);
#pragma unroll
        for (int r = c + 1; r < 6; r++) {
          const double f = bcast(col[r], c) / dcc;
          if (lane >= c && lane < 7) col[r] -= f * col[c];
        }
      }
    }
    double M[6][7];
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
      for (int j = 0; j < 7; j++) M[i][j] = bcast(col[i], j);
    if (lane == 0) {
      if (ok) {
#pragma unroll
        for (int c = 5; c >= 0; c--) {
          double sacc = M[c][6];
#pragma unroll
          for (int k = c + 1; k < 6; k++) sacc -= M[c][k] * M[k][6];
          M[c][6] = sacc / M[c][c];
        }
        const double rpy[3] = {M[0][6], M[1][6], M[2][6]}, t[3] = {M[3][6], M[4][6], M[5][6]};
        if (rpy[0] != rpy[0] || rpy[1] != rpy[1] || rpy[2] != rpy[2] || t[0] != t[0] || t[1] != t[1] || t[2] != t[2]) ok = false;
        if (ok) {
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
          done = (!(perc < (1 + tp) && perc > (1 - tp)) && it < st->max_iter) ? 0 : 1;
        }
      }
      if (!ok) done = 1;
    }
  }
  if (lane != 0) return;
  st->n_sel = n_sel;
  if (done) { /* the level is over: pose = PoseX * pose (what the host did between levels) */
    double tmp[16];
    ppf_mat44_mul(st->PoseX, st->pose, tmp);
    for (int k = 0; k < 16; k++) st->pose[k] = tmp[k];
    st->total += st->iter;
    st->done = 1;
    if (last_level) icp_publish_state(B, job); /* what the host needs in the end: pose, residual, iterations, centring, scale */
    __hip_atomic_store(&B.h_done[job], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  ICP_PH(st, 5);
  ICP_PH_FLUSH(st);
  icp_tick(B, done != 0); /* after everything this thread wrote for the host */
}

#endif /* PPF_ICP_KERNELS_H */
