/*
 * ppf_prep_kernels.h — the stages that produce the matcher's N x 6 input, on gfx950 (SURVEY.md §8f row N4;
 * /root/reference/include/CloudProcessing.h: SceneCropping :263-339, Subsampling :361-380, OutlierProcessing :341-360,
 * NormalEstimation :381-405, EdgeExtraction :406-427, PointCloudXYZNormalToMat :163-190).  Included by ppf_hip.hip.
 *
 * The arithmetic and every order-dependent choice is the one oracle/ppf_prep_oracle.cpp freezes; results are
 * bit-identical to it.  Clouds are device rows `x y z nx ny nz` (pitch 6) plus a curvature array.
 *   crop / outlier / edge : per-point predicate -> flags -> exclusive scan -> ordered gather (HBM streaming, 28 B/pt)
 *   voxel grid            : finite min/max -> PCL's cell index -> stable LSD radix sort (shared with the sampler) ->
 *                           one thread per cell, float sums in point order
 *   k nearest neighbours  : uniform grid (cells sorted by the same radix sort), ONE WAVE per query, cube of cells
 *                           grown until the k-th distance is provably final; the k <= 64 best (distance bits, index)
 *                           keys live one per lane and 64 candidates at a time are merged in by a register bitonic
 *                           network.  Exact: equals the oracle's exhaustive search bit for bit.
 *   normals               : one thread per point, fp64 two-pass covariance over its neighbour list, cyclic Jacobi in
 *                           registers (12 sweeps, + - * / sqrt only), smallest eigenvector, flip towards the origin.
 */
#ifndef PPF_PREP_KERNELS_H
#define PPF_PREP_KERNELS_H

constexpr int KNN_WAVES = 4;  /* queries (waves) per workgroup */
constexpr int KNN_MAX_K = 64; /* the k best keys live one per lane */

struct CropPlanes {
  double n[4][3]; /* inward normals of the four side planes through the origin */
  float z_base;
};

__device__ __forceinline__ bool prep_finite3(const float* p) { return isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]); }

/* flags[i] = point inside the pyramid {origin, 4 corners} */
__global__ __launch_bounds__(256) void k_prep_crop_flags(const float* __restrict__ rows, int n, CropPlanes pl, uint32_t* __restrict__ flags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = rows + (size_t)i * 6;
  bool in = prep_finite3(p) && p[2] <= pl.z_base;
#pragma unroll
  for (int f = 0; f < 4; f++) in = in && (pl.n[f][0] * (double)p[0] + pl.n[f][1] * (double)p[1] + pl.n[f][2] * (double)p[2]) >= 0.0;
  flags[i] = in ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_prep_finite_flags(const float* __restrict__ rows, int n, uint32_t* __restrict__ flags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flags[i] = prep_finite3(rows + (size_t)i * 6) ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_prep_curv_flags(const float* __restrict__ curv, int n, float thr, uint32_t* __restrict__ flags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flags[i] = curv[i] > thr ? 1u : 0u;
}
/* ordered gather of the flagged rows (pos = exclusive scan of flags) */
__global__ __launch_bounds__(256) void k_prep_gather(const float* __restrict__ rows, const float* __restrict__ curv, int n,
                                                     const uint32_t* __restrict__ flags, const uint32_t* __restrict__ pos,
                                                     float* __restrict__ out_rows, float* __restrict__ out_curv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !flags[i]) return;
  const uint32_t o = pos[i];
#pragma unroll
  for (int k = 0; k < 6; k++) out_rows[(size_t)o * 6 + k] = rows[(size_t)i * 6 + k];
  out_curv[o] = curv[i];
}

/* rows of `cols` floats at `stride` -> packed rows of 6 (+ zero curvature) */
__global__ __launch_bounds__(256) void k_prep_pack(const float* __restrict__ src, int n, int stride, int noff, int cols, float* __restrict__ rows,
                                                   float* __restrict__ curv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    rows[(size_t)i * 6 + k] = src[(size_t)i * stride + k];
    rows[(size_t)i * 6 + 3 + k] = cols == 6 ? src[(size_t)i * stride + noff + k] : 0.f;
  }
  curv[i] = 0.f;
}

/* ---- voxel grid ------------------------------------------------------------------------------------------- */
/* mm[0..2] = min, mm[3..5] = max as order-preserving uints; input must be finite */
__global__ __launch_bounds__(256) void k_prep_minmax(const float* __restrict__ rows, int n, uint32_t* __restrict__ mm) {
  uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0, 0, 0};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const uint32_t o = float_to_ordered(rows[(size_t)i * 6 + k]);
      lo[k] = min(lo[k], o);
      hi[k] = max(hi[k], o);
    }
  }
#pragma unroll
  for (int k = 0; k < 3; k++) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      lo[k] = min(lo[k], (uint32_t)__shfl_down(lo[k], o));
      hi[k] = max(hi[k], (uint32_t)__shfl_down(hi[k], o));
    }
  }
  /* one set of atomics per workgroup, and few workgroups (the host caps the grid): six addresses serve them one at a time, and
   * 560 waves' worth were 35 of this kernel's 41 us */
  __shared__ uint32_t s_mm[4][6];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < 3; k++) { s_mm[wave][k] = lo[k]; s_mm[wave][3 + k] = hi[k]; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int k = threadIdx.x;
    uint32_t v = s_mm[0][k];
    for (int w = 1; w < 4; w++) v = k < 3 ? min(v, s_mm[w][k]) : max(v, s_mm[w][k]);
    if (k < 3) atomicMin(&mm[k], v); else atomicMax(&mm[k], v);
  }
}
struct VoxelGridDims {
  float inv_leaf;
  int min_b[3], div_b[3];
};
/* PCL: ijk = (int)(floor(p * inv_leaf) - (float)min_b); idx = i + j*div0 + k*div0*div1 */
__global__ __launch_bounds__(256) void k_prep_voxel_keys(const float* __restrict__ rows, int n, VoxelGridDims g, uint32_t* __restrict__ keys,
                                                         uint32_t* __restrict__ vals) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = rows + (size_t)i * 6;
  const int i0 = ppf_f2i(floorf(p[0] * g.inv_leaf) - (float)g.min_b[0]);
  const int i1 = ppf_f2i(floorf(p[1] * g.inv_leaf) - (float)g.min_b[1]);
  const int i2 = ppf_f2i(floorf(p[2] * g.inv_leaf) - (float)g.min_b[2]);
  keys[i] = (uint32_t)(i0 + i1 * g.div_b[0] + i2 * g.div_b[0] * g.div_b[1]);
  vals[i] = (uint32_t)i;
}
/* one thread per occupied cell: float sums in ascending point order, divided by the float count */
__global__ __launch_bounds__(64) void k_prep_voxel_sum(const float* __restrict__ rows, const uint32_t* __restrict__ vals,
                                                       const uint32_t* __restrict__ starts, int n_cells, int n, float* __restrict__ out_rows,
                                                       float* __restrict__ out_curv) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_cells) return;
  const uint32_t s = starts[r], e = (r + 1 < n_cells) ? starts[r + 1] : (uint32_t)n;
  float acc[3] = {0.f, 0.f, 0.f};
  for (uint32_t k = s; k < e; k++) {
    const float* p = rows + (size_t)vals[k] * 6;
    acc[0] += p[0]; acc[1] += p[1]; acc[2] += p[2];
  }
  const float cnt = (float)(e - s);
  out_rows[(size_t)r * 6] = acc[0] / cnt; out_rows[(size_t)r * 6 + 1] = acc[1] / cnt; out_rows[(size_t)r * 6 + 2] = acc[2] / cnt;
  out_rows[(size_t)r * 6 + 3] = 0.f; out_rows[(size_t)r * 6 + 4] = 0.f; out_rows[(size_t)r * 6 + 5] = 0.f;
  out_curv[r] = 0.f;
}

/* ---- exact k nearest neighbours ----------------------------------------------------------------------------- */
/* Uniform grid over the cloud's bounding box; points sorted by cell (x fastest).  A query scans the cube of cells of
 * radius r around its own cell and keeps its k best keys (float d2 bits << 32 | original index: d2 >= +0, so the
 * u64 order IS (distance, index) order, independent of the scan order).  Every point
 * outside that cube is farther than r*h, so the list is final once its k-th distance is within (0.9999 r h)^2 (the
 * margin covers the float rounding of the cell assignment); otherwise the cube grows.  When the cube covers the whole
 * grid the search has been exhaustive.  Results equal the oracle's brute force bit for bit. */
struct KnnGrid {
  float lo[3];
  float inv_h, h;
  int dim[3];
};
__device__ __forceinline__ void knn_cell(const KnnGrid& g, float x, float y, float z, int* c) {
  c[0] = min(max(ppf_f2i(floorf((x - g.lo[0]) * g.inv_h)), 0), g.dim[0] - 1);
  c[1] = min(max(ppf_f2i(floorf((y - g.lo[1]) * g.inv_h)), 0), g.dim[1] - 1);
  c[2] = min(max(ppf_f2i(floorf((z - g.lo[2]) * g.inv_h)), 0), g.dim[2] - 1);
}
__global__ __launch_bounds__(256) void k_prep_knn_keys(const float* __restrict__ rows, int n, KnnGrid g, uint32_t* __restrict__ keys,
                                                       uint32_t* __restrict__ vals, uint32_t* __restrict__ cell_count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int c[3];
  knn_cell(g, rows[(size_t)i * 6], rows[(size_t)i * 6 + 1], rows[(size_t)i * 6 + 2], c);
  const uint32_t key = (uint32_t)((c[2] * g.dim[1] + c[1]) * g.dim[0] + c[0]);
  keys[i] = key;
  vals[i] = (uint32_t)i;
  atomicAdd(&cell_count[key], 1u);
}
/* pts[s] = xyz of the s-th point in cell order, w = its original index */
__global__ __launch_bounds__(256) void k_prep_knn_pack(const float* __restrict__ rows, const uint32_t* __restrict__ order, int n,
                                                       float4* __restrict__ pts) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const uint32_t i = order[s];
  pts[s] = make_float4(rows[(size_t)i * 6], rows[(size_t)i * 6 + 1], rows[(size_t)i * 6 + 2], __uint_as_float(i));
}
/* bitonic compare-exchange across lanes */
__device__ __forceinline__ unsigned long long knn_cex(unsigned long long key, int stride, bool take_min) {
  const unsigned long long other = __shfl_xor(key, stride);
  return take_min ? (key < other ? key : other) : (key < other ? other : key);
}
/* ONE WAVE PER QUERY.  idx/d2: [n][k] at the ORIGINAL row of each point, ascending (d2, index); k <= 64.
 * The wave reads 64 candidates per step (coalesced within a row of cells); its current k best keys live one per
 * lane, ascending by lane (lanes >= k hold the sentinel).  A step whose candidates all fail the k-th key costs one
 * ballot; otherwise the 64 new keys are bitonic-sorted across the lanes (21 exchanges) and merged with the list
 * (reverse + min = the 64 smallest of both as a bitonic sequence, 6 more exchanges).  No LDS. */
__global__ __launch_bounds__(256) void k_prep_knn(const float4* __restrict__ pts, const uint32_t* __restrict__ cell_begin, KnnGrid g,
                                                  int n, int k, int* __restrict__ idx_out, float* __restrict__ d2_out) {
  const int lane = threadIdx.x & 63;
  const int s = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); /* wave-uniform */
  if (s >= n) return;
  const float4 p = pts[s];
  int c[3];
  knn_cell(g, p.x, p.y, p.z, c);
  const unsigned long long none = ~0ull;
  unsigned long long best;
  for (int r = 1;; r++) {
    best = none;
    unsigned long long worst = none; /* the k-th key, wave-uniform */
    const int x0 = max(c[0] - r, 0), x1 = min(c[0] + r, g.dim[0] - 1);
    const int y0 = max(c[1] - r, 0), y1 = min(c[1] + r, g.dim[1] - 1);
    const int z0 = max(c[2] - r, 0), z1 = min(c[2] + r, g.dim[2] - 1);
    for (int cz = z0; cz <= z1; cz++)
      for (int cy = y0; cy <= y1; cy++) {
        const int base = (cz * g.dim[1] + cy) * g.dim[0];
        const uint32_t jb = cell_begin[base + x0], je = cell_begin[base + x1 + 1];
        for (uint32_t j0 = jb; j0 < je; j0 += 64) {
          const uint32_t j = j0 + (uint32_t)lane;
          unsigned long long key = none;
          if (j < je) {
            const float4 q = pts[j];
            const float dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
            const float d = (dx * dx + dy * dy) + dz * dz;
            if (d >= 0.f) key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)__float_as_uint(q.w); /* drops NaN */
          }
          if (!__any(key < worst)) continue;
          /* sort the 64 new keys ascending by lane */
#pragma unroll
          for (int size = 2; size <= 64; size <<= 1) {
            const bool up = (lane & size) == 0 || size == 64;
#pragma unroll
            for (int stride = size >> 1; stride > 0; stride >>= 1) key = knn_cex(key, stride, ((lane & stride) == 0) == up);
          }
          /* the 64 smallest of (best, key): best ascending, key reversed -> elementwise min is bitonic */
          const unsigned long long rev = __shfl(key, 63 - lane);
          best = best < rev ? best : rev;
#pragma unroll
          for (int stride = 32; stride > 0; stride >>= 1) best = knn_cex(best, stride, (lane & stride) == 0);
          if (lane >= k) best = none;
          worst = __shfl(best, k - 1);
        }
      }
    const bool whole = x0 == 0 && y0 == 0 && z0 == 0 && x1 == g.dim[0] - 1 && y1 == g.dim[1] - 1 && z1 == g.dim[2] - 1;
    const float lim = (float)r * g.h * 0.9999f;
    if (whole || (worst != none && __uint_as_float((uint32_t)(worst >> 32)) <= lim * lim)) break;
  }
  if (lane < k) {
    const size_t row = (size_t)__float_as_uint(p.w) * k;
    idx_out[row + lane] = best == none ? -1 : (int)(uint32_t)best;
    d2_out[row + lane] = best == none ? 0.f : __uint_as_float((uint32_t)(best >> 32));
  }
}

/* ---- statistical outlier removal ---------------------------------------------------------------------------- */
/* dist[i] = (float)(sum_{m=1..mean_k} sqrtf(d2[i][m]) / mean_k), fp64 sum in neighbour order; 0 when n <= mean_k */
__global__ __launch_bounds__(256) void k_prep_sor_dist(const float* __restrict__ d2, int n, int mean_k, int valid, float* __restrict__ dist) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (!valid) { dist[i] = 0.f; return; }
  double s = 0;
  for (int m = 1; m <= mean_k; m++) s += (double)sqrtf(d2[(size_t)i * (mean_k + 1) + m]);
  dist[i] = (float)(s / (double)mean_k);
}
/* per-chunk (64 points) sums of d and d*d in fp64 */
__global__ __launch_bounds__(64) void k_prep_sor_chunks(const float* __restrict__ dist, int n, double* __restrict__ parts) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int c0 = c * 64;
  if (c0 >= n) return;
  double ps = 0, pq = 0;
  for (int i = c0; i < min(n, c0 + 64); i++) { const double v = (double)dist[i]; ps += v; pq += v * v; }
  parts[(size_t)c * 2] = ps; parts[(size_t)c * 2 + 1] = pq;
}
/* out[0] = mean + mul * stddev */
__global__ __launch_bounds__(64) void k_prep_sor_threshold(const double* __restrict__ parts, int n, double std_mul, double* __restrict__ out) {
  __shared__ double tot[2];
  const int n_chunks = (n + 63) / 64;
  if (threadIdx.x < 2) tot[threadIdx.x] = icp_sum_parts(parts + threadIdx.x, n_chunks, 2);
  __syncthreads();
  if (threadIdx.x == 0) {
    const double sum = tot[0], sq = tot[1];
    const double mean = sum / (double)n;
    const double variance = (sq - sum * sum / (double)n) / ((double)n - 1);
    out[0] = mean + std_mul * ppf_sqrt(variance);
  }
}
__global__ __launch_bounds__(256) void k_prep_sor_flags(const float* __restrict__ dist, int n, const double* __restrict__ thr, uint32_t* __restrict__ flags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flags[i] = !((double)dist[i] > thr[0]) ? 1u : 0u;
}

/* ---- normals + curvature ------------------------------------------------------------------------------------ */
__device__ __forceinline__ void prep_jacobi_rotate(double (&A)[3][3], double (&V)[3][3], const int p, const int q) {
  const double apq = A[p][q];
  if (apq == 0.0) return;
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
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const double vkp = V[k][p], vkq = V[k][q];
    V[k][p] = c * vkp - s * vkq;
    V[k][q] = s * vkp + c * vkq;
  }
}

/* in place: rows[i][3..5] = normal, curv[i] = curvature; idx = [n][k] neighbour lists (k_eff valid entries) */
__global__ __launch_bounds__(64) void k_prep_normals(float* __restrict__ rows, float* __restrict__ curv, int n, const int* __restrict__ idx,
                                                     int k, const float4* __restrict__ q4) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float* o = rows + (size_t)i * 6;
  if (k < 3) {
    const float qn = __builtin_nanf("");
    o[3] = o[4] = o[5] = qn; curv[i] = qn;
    return;
  }
  const int* nb = idx + (size_t)i * k;
  double c[3] = {0, 0, 0};
  for (int m = 0; m < k; m++) { const float4 q = q4[nb[m]]; c[0] += (double)q.x; c[1] += (double)q.y; c[2] += (double)q.z; }
  c[0] /= (double)k; c[1] /= (double)k; c[2] /= (double)k;
  double cov[6] = {0, 0, 0, 0, 0, 0};
  for (int m = 0; m < k; m++) {
    const float4 q = q4[nb[m]];
    const double d0 = (double)q.x - c[0], d1 = (double)q.y - c[1], d2 = (double)q.z - c[2];
    cov[0] += d0 * d0; cov[1] += d0 * d1; cov[2] += d0 * d2;
    cov[3] += d1 * d1; cov[4] += d1 * d2; cov[5] += d2 * d2;
  }
#pragma unroll
  for (int a = 0; a < 6; a++) cov[a] /= (double)k;
  double A[3][3] = {{cov[0], cov[1], cov[2]}, {cov[1], cov[3], cov[4]}, {cov[2], cov[4], cov[5]}};
  double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  const double trace = cov[0] + cov[3] + cov[5];
  for (int sweep = 0; sweep < 12; sweep++) {
    prep_jacobi_rotate(A, V, 0, 1);
    prep_jacobi_rotate(A, V, 0, 2);
    prep_jacobi_rotate(A, V, 1, 2);
  }
  double lam = A[0][0], nv[3] = {V[0][0], V[1][0], V[2][0]};
  if (A[1][1] < lam) { lam = A[1][1]; nv[0] = V[0][1]; nv[1] = V[1][1]; nv[2] = V[2][1]; }
  if (A[2][2] < lam) { lam = A[2][2]; nv[0] = V[0][2]; nv[1] = V[1][2]; nv[2] = V[2][2]; }
  const double len = ppf_sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
  nv[0] /= len; nv[1] /= len; nv[2] /= len;
  const double cos_theta = -((double)o[0] * nv[0] + (double)o[1] * nv[1] + (double)o[2] * nv[2]);
  if (cos_theta < 0) { nv[0] = -nv[0]; nv[1] = -nv[1]; nv[2] = -nv[2]; }
  o[3] = (float)nv[0]; o[4] = (float)nv[1]; o[5] = (float)nv[2];
  if (lam < 0) lam = -lam;
  const double at = trace < 0 ? -trace : trace;
  curv[i] = trace != 0.0 ? (float)(lam / at) : 0.f;
}

/* PointCloudXYZNormalToMat: n /= (float)sqrtf(n.n) when that length exceeds 1e-5 */
__global__ __launch_bounds__(256) void k_prep_to_mat(const float* __restrict__ rows, int n, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float d[6];
#pragma unroll
  for (int k = 0; k < 6; k++) d[k] = rows[(size_t)i * 6 + k];
  const float s = d[3] * d[3] + d[4] * d[4] + d[5] * d[5];
  const double A = (double)sqrtf(s);
  if (A > 0.00001) { d[3] /= (float)A; d[4] /= (float)A; d[5] /= (float)A; }
#pragma unroll
  for (int k = 0; k < 6; k++) out[(size_t)i * 6 + k] = d[k];
}

#endif /* PPF_PREP_KERNELS_H */
