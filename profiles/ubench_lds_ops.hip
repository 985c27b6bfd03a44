// LDS instruction cost on gfx950: what one wave-instruction of each kind costs the CU's LDS pipe, with all four SIMDs issuing.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_lds_ops profiles/ubench_lds_ops.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s\n", hipGetErrorString(e)); return 1;}}while(0)
constexpr int WORDS = 32768; // 128 KB
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int OP> __global__ __launch_bounds__(1024) void k(uint32_t* out, int iters) {
  extern __shared__ __align__(16) uint32_t acc[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < WORDS; i += 1024) acc[i] = 0;
  __syncthreads();
  uint32_t s = 0;
  // per-lane base addresses, drawn once; every iteration moves all of them by the same amount, which keeps the bank
  // pattern of the wave (conflicts depend on address differences) and costs one VALU add per operation
  uint32_t base[8], adr[8];
  const uint32_t h = mix((uint32_t)tid * 2654435761u + 12345u);
#pragma unroll
  for (int u = 0; u < 8; u++) {
    const uint32_t hu = mix(h + (uint32_t)u * 0x9E3779B9u);
    uint32_t w;
    if (OP == 0 || OP == 1 || OP == 2 || OP == 4 || OP == 10 || OP == 11) w = ((uint32_t)tid + 64u * (uint32_t)u) & 16383u;   // conflict-free
    else if (OP == 3 || OP == 5) w = (2u * (uint32_t)tid + 128u * (uint32_t)u) & 16382u;                                        // 8-byte slots, conflict-free
    else if (OP == 6) w = hu & 16383u;                                                                                         // random words
    else if (OP == 7) w = ((hu % 500u) * 31u + ((hu >> 20) % 30u));                                                            // vote-like, pitch 31
    else if (OP == 8) w = ((hu % 500u) * 32u + ((hu >> 20) % 30u));                                                            // pitch 32
    else if (OP == 9) w = ((hu % 1000u) * 16u + ((hu >> 20) % 15u));                                                           // packed halves: pitch 16 words
    else if (OP == 12) w = ((hu % 900u) * 17u + ((hu >> 20) % 15u));                                                           // packed, pitch 17
    else if (OP == 13) w = (((hu % 500u) * 32u + ((hu >> 20) % 30u)) & ~1u);                                                   // u64 on vote-like rows
    else if (OP == 14) w = (hu & 16382u);                                                                                      // u64 random
    else if (OP == 15) w = (hu & 16383u);                                                                                      // read random
    else if (OP == 16) w = ((mix((uint32_t)(lane & 31) * 7919u + (uint32_t)u) * 0u + (((uint32_t)lane * 13u + (uint32_t)u * 5u) & 31u)) + 32u * (hu % 500u)); // distinct mod 32 inside each half-wave, random otherwise
    else if (OP == 17) w = ((((uint32_t)lane * 13u + (uint32_t)u * 5u) & 63u) + 64u * (hu % 250u));                             // distinct mod 64 over the wave, random otherwise
    else if (OP == 18) w = ((((uint32_t)lane * 5u + (uint32_t)u * 3u) & 15u) + 16u * (hu % 1000u));                             // distinct mod 16 inside each group of 16 lanes
    else if (OP == 19) w = (((uint32_t)lane & 15u) + 64u * (((uint32_t)lane >> 4) + 4u * (uint32_t)u + 32u * (hu % 7u)));   // distinct banks inside each 16 lanes, the same 16 banks in all four groups
    else if (OP == 20) w = (((uint32_t)lane & 31u) + 64u * (((uint32_t)lane >> 5) + 2u * (uint32_t)u + 16u * (hu % 13u)));  // distinct inside each 32 lanes, the same 32 banks in both halves
    else if (OP == 21) w = ((((uint32_t)lane & 15u) * 4u + ((uint32_t)lane >> 4)) + 64u * ((uint32_t)u + 8u * (hu % 29u)));   // all 64 lanes distinct banks, lanes l and l+16 in neighbouring banks
    else w = 0;
    base[u] = w;
  }
  for (int it = 0; it < iters; it++) {
    const uint32_t mv = ((uint32_t)it * 34u) & 16382u;
#pragma unroll
    for (int u = 0; u < 8; u++) adr[u] = base[u] + mv;
#pragma unroll
    for (int u = 0; u < 8; u++) {
      uint32_t* p = acc + adr[u];
      if (OP == 0 || OP == 6 || OP == 7 || OP == 8 || OP == 16 || OP == 17 || OP == 18 || OP == 19 || OP == 20 || OP == 21) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (OP == 9 || OP == 12) __hip_atomic_fetch_add(p, 1u << (16u * ((h >> (u + 3)) & 1u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (OP == 1) { if (lane < 16) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
      if (OP == 2) { if ((lane & 3) == 0) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
      if (OP == 3 || OP == 13 || OP == 14) __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(p), 0x100000001ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (OP == 4 || OP == 15) s += *(volatile lds_u32*)(lds_u32*)p;
      if (OP == 5) { const unsigned long long v = *(volatile lds_u64*)(lds_u64*)p; s += (uint32_t)v + (uint32_t)(v >> 32); }
      if (OP == 10) s += __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (OP == 11) *(volatile lds_u32*)(lds_u32*)p = h;
    }
  }
  __syncthreads();
  out[blockIdx.x * 1024 + tid] = s + acc[tid];
}
template <int OP> float run(uint32_t* d, int iters) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, WORDS * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<256, 1024, WORDS * 4>>>(d, 10); hipDeviceSynchronize();
  hipEventRecord(e0); k<OP><<<256, 1024, WORDS * 4>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  uint32_t* d; CHK(hipMalloc(&d, 256 * 1024 * 4));
  const int iters = 4000;
  const char* names[] = {"ds_add_u32 conflict-free", "ds_add_u32 lanes 0..15 only", "ds_add_u32 every 4th lane", "ds_add_u64 conflict-free", "ds_read_b32", "ds_read_b64",
                         "ds_add_u32 random words", "ds_add_u32 vote-like pitch 31", "ds_add_u32 vote-like pitch 32", "ds_add_u32 packed-half rows pitch 16",
                         "ds_add_rtn_u32 conflict-free", "ds_write_b32", "ds_add_u32 packed-half rows pitch 17", "ds_add_u64 vote-like pitch 32", "ds_add_u64 random 8-byte slots", "ds_read_b32 random words",
                         "ds_add_u32 banks distinct mod 32 per half-wave", "ds_add_u32 banks distinct mod 64 per wave", "ds_add_u32 banks distinct mod 16 per 16 lanes",
                         "ds_add_u32 16 banks, distinct inside each 16 lanes, shared by the 4 groups", "ds_add_u32 32 banks, distinct inside each 32 lanes, shared by the halves", "ds_add_u32 64 distinct banks, interleaved groups"};
  float t[22];
  t[0] = run<0>(d, iters); t[1] = run<1>(d, iters); t[2] = run<2>(d, iters); t[3] = run<3>(d, iters); t[4] = run<4>(d, iters); t[5] = run<5>(d, iters); t[6] = run<6>(d, iters);
  t[7] = run<7>(d, iters); t[8] = run<8>(d, iters); t[9] = run<9>(d, iters); t[10] = run<10>(d, iters); t[11] = run<11>(d, iters); t[12] = run<12>(d, iters); t[13] = run<13>(d, iters); t[14] = run<14>(d, iters); t[15] = run<15>(d, iters); t[16] = run<16>(d, iters); t[17] = run<17>(d, iters); t[18] = run<18>(d, iters); t[19] = run<19>(d, iters); t[20] = run<20>(d, iters); t[21] = run<21>(d, iters);
  // one workgroup of 16 waves per CU: wave-instructions per CU = iters * 8 * 16
  for (int i = 0; i < 22; i++) printf("%-40s %8.3f ms  -> %6.2f cycles per wave-instruction per CU\n", names[i], t[i], t[i] * 1e-3 * 2.4e9 / ((double)iters * 8 * 16));
  return 0;
}
