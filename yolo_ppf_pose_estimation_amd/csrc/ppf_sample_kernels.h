/*
 * ppf_sample_kernels.h — cloud sampling (computeBboxStd + samplePCByQuantization, the first
 * step inside trainModel() and match(); /root/reference/include/CloudProcessing.h:236,442,495) on gfx950.
 * Included by ppf_hip.hip.
 *
 * The reference bins every point into an (n+1)^3 grid with float arithmetic, then emits one row per non-empty
 * cell IN ASCENDING CELL ORDER: mean position and re-normalised summed normal, accumulated in fp64 IN POINT
 * ORDER.  Both orders decide the bits of the output, so the device version is
 *   bbox (ordered-uint atomics) -> cell key per point -> STABLE LSD radix sort of (key, point index) ->
 *   segment starts -> one thread per cell sums its points sequentially (ascending point index).
 * Everything is HBM-bound streaming over 24 B points except the per-cell serial sums.
 */
#ifndef PPF_SAMPLE_KERNELS_H
#define PPF_SAMPLE_KERNELS_H

constexpr int RS_BLOCK = 1024; /* elements per radix-sort tile == threads per block */

__device__ __forceinline__ uint32_t float_to_ordered(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ __forceinline__ float ordered_to_float(uint32_t o) {
  const uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  float f;
  __builtin_memcpy(&f, &u, 4);
  return f;
}

/* bbox[0..2] = min xyz, bbox[3..5] = max xyz, as order-preserving uints (init: 0xFFFFFFFF / 0) */
__global__ __launch_bounds__(256) void k_bbox(const float* __restrict__ src, int n, int stride, uint32_t* __restrict__ bbox) {
  uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0, 0, 0};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float* p = src + (size_t)i * stride;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const uint32_t o = float_to_ordered(p[k]);
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
  /* one set of atomics per workgroup, and few workgroups (the host caps the grid): six addresses serve them one at a time */
  __shared__ uint32_t s_bb[4][6];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < 3; k++) { s_bb[wave][k] = lo[k]; s_bb[wave][3 + k] = hi[k]; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int k = threadIdx.x;
    uint32_t v = s_bb[0][k];
    for (int w = 1; w < 4; w++) v = k < 3 ? min(v, s_bb[w][k]) : max(v, s_bb[w][k]);
    if (k < 3) atomicMin(&bbox[k], v); else atomicMax(&bbox[k], v);
  }
}

/* cell index of every point, float arithmetic exactly as the reference: (int)((float)n*(p-min)/range) */
__global__ __launch_bounds__(256) void k_cell_keys(const float* __restrict__ src, int n, int stride,
                                                   const uint32_t* __restrict__ bbox, int ns, uint32_t* __restrict__ keys,
                                                   uint32_t* __restrict__ vals) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = src + (size_t)i * stride;
  int c[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float lo = ordered_to_float(bbox[k]), hi = ordered_to_float(bbox[3 + k]);
    const float rg = hi - lo;
    c[k] = rg > 0.0f ? ppf_f2i((float)ns * (p[k] - lo) / rg) : 0;
  }
  keys[i] = (uint32_t)(c[0] * ns * ns + c[1] * ns + c[2]);
  vals[i] = (uint32_t)i;
}

/* ---- stable LSD radix sort, 8-bit digits, tiles of 1024 ----------------------------------------------------- */
__global__ __launch_bounds__(RS_BLOCK) void k_rs_hist(const uint32_t* __restrict__ keys, int n, int shift, int nblk,
                                                      uint32_t* __restrict__ hist /* [256][nblk] */) {
  __shared__ uint32_t h[256];
  const int tid = threadIdx.x;
  if (tid < 256) h[tid] = 0;
  __syncthreads();
  const int i = blockIdx.x * RS_BLOCK + tid;
  if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);
  __syncthreads();
  if (tid < 256) hist[(size_t)tid * nblk + blockIdx.x] = h[tid];
}

__global__ __launch_bounds__(RS_BLOCK) void k_rs_scatter(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                         int n, int shift, int nblk, const uint32_t* __restrict__ offs,
                                                         uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out) {
  __shared__ uint32_t wcnt[RS_BLOCK / 64][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int k = tid; k < (RS_BLOCK / 64) * 256; k += RS_BLOCK) (&wcnt[0][0])[k] = 0;
  __syncthreads();
  const int i = blockIdx.x * RS_BLOCK + tid;
  const bool valid = i < n;
  uint32_t key = 0, val = 0, d = 0;
  if (valid) { key = keys[i]; val = vals[i]; d = (key >> shift) & 255u; }
  unsigned long long same = __ballot(valid);
#pragma unroll
  for (int bit = 0; bit < 8; bit++) {
    const unsigned long long bb = __ballot((d >> bit) & 1u);
    same &= ((d >> bit) & 1u) ? bb : ~bb;
  }
  const uint32_t rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
  if (valid && rank == 0) wcnt[wave][d] = (uint32_t)__popcll(same);
  __syncthreads();
  if (valid) {
    uint32_t pos = offs[(size_t)d * nblk + blockIdx.x] + rank;
    for (int w = 0; w < wave; w++) pos += wcnt[w][d];
    keys_out[pos] = key;
    vals_out[pos] = val;
  }
}

__global__ __launch_bounds__(256) void k_seg_flags(const uint32_t* __restrict__ keys, int n, uint32_t* __restrict__ flags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_seg_starts(const uint32_t* __restrict__ flags, const uint32_t* __restrict__ segid, int n,
                                                    uint32_t* __restrict__ starts) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flags[i]) starts[segid[i]] = (uint32_t)i;
}

/* one thread per non-empty cell: fp64 sums in ascending point order, mean, re-normalised normal */
__global__ __launch_bounds__(64) void k_seg_sum(const float* __restrict__ src, int stride, int noff, const uint32_t* __restrict__ vals,
                                                const uint32_t* __restrict__ starts, int n_rows, int n,
                                                float* __restrict__ soa, int pitch, float* __restrict__ aos) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  const uint32_t s = starts[r], e = (r + 1 < n_rows) ? starts[r + 1] : (uint32_t)n;
  double acc[6] = {0, 0, 0, 0, 0, 0};
  for (uint32_t k = s; k < e; k++) {
    const float* p = src + (size_t)vals[k] * stride;
#pragma unroll
    for (int c = 0; c < 3; c++) { acc[c] += (double)p[c]; acc[3 + c] += (double)p[noff + c]; }
  }
  const double cn = (double)(e - s);
#pragma unroll
  for (int c = 0; c < 6; c++) acc[c] /= cn;
  float row[6] = {(float)acc[0], (float)acc[1], (float)acc[2], 0.f, 0.f, 0.f};
  const double norm = ppf_sqrt(acc[3] * acc[3] + acc[4] * acc[4] + acc[5] * acc[5]);
  if (norm > PPF_EPS) {
    row[3] = (float)(acc[3] / norm); row[4] = (float)(acc[4] / norm); row[5] = (float)(acc[5] / norm);
  }
#pragma unroll
  for (int c = 0; c < 6; c++) {
    if (soa) soa[(size_t)c * pitch + r] = row[c];
    if (aos) aos[(size_t)r * 6 + c] = row[c];
  }
}

#endif /* PPF_SAMPLE_KERNELS_H */
