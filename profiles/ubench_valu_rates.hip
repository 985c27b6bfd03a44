#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
// issue rate of a few VALU instructions: 8 independent chains per wave, 4 waves per SIMD, all CUs
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, int iters) {
  uint32_t v[8];
  for (int i = 0; i < 8; i++) v[i] = seed + threadIdx.x * 7 + i;
  uint32_t c = seed | 1u, d = 0x3f800000u; unsigned long long msk = 0x5555555555555555ull ^ seed;
 double dd = (double)seed, de = 1.0000001;
  asm volatile("s_mov_b64 vcc, %0" :: "s"(msk) : "vcc");
  for (int it = 0; it < iters; it++) {
#define OPX(i) \
    if (OP == 0) asm volatile("v_mad_u32_u16 %0, %0, 4, %1 op_sel:[1,0,0,0]" : "+v"(v[i]) : "v"(c)); \
    else if (OP == 1) asm volatile("v_min3_u16 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c), "v"(d)); \
    else if (OP == 2) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(v[i])); \
    else if (OP == 3) asm volatile("v_fract_f32 %0, %0" : "+v"(v[i])); \
    else if (OP == 4) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(v[i]) : "v"(c)); \
    else if (OP == 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(d), "v"(c)); \
    else if (OP == 6) asm volatile("v_bfe_u32 %0, %0, 16, 7" : "+v"(v[i])); \
    else if (OP == 7) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c), "v"(d)); \
    else if (OP == 8) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c), "v"(d)); \
    else if (OP == 9) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c), "v"(d)); \
    else if (OP == 10) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(c)); \
    else if (OP == 19) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(c)); \
    else if (OP == 20) asm volatile("v_cndmask_b32_e32 %0, %1, %0, vcc" : "+v"(v[i]) : "v"(c)); \
    else if (OP == 21) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(v[i]) : "v"(c) : "vcc"); \
    else if (OP == 22) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(c) : "vcc"); \
    else if (OP == 23) asm volatile("v_cmp_lt_u32_e64 s[10:11], %0, %1\n v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(v[i]) : "v"(c) : "s10", "s11"); \
    else if (OP == 24) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(c) : "vcc"); \
    else if (OP == 11) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c), "s"(msk)); \
    else if (OP == 12) asm volatile("v_min_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c)); \
    else if (OP == 13) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[i]) : "v"(c)); \
    else if (OP == 14) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(c) : "vcc"); \
    else if (OP == 15) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c), "v"(d)); \
    else if (OP == 16) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(c)); \
    else if (OP == 17) asm volatile("v_cvt_f32_f64 %0, %1" : "+v"(v[i]) : "v"(dd)); \
    else if (OP == 18) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(dd) : "v"(de));
    REP8(OPX) REP8(OPX) REP8(OPX) REP8(OPX)
  }
  uint32_t s = (uint32_t)dd; for (int i = 0; i < 8; i++) s ^= v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char* name, uint32_t* d) {
  const int iters = 2000, blocks = 256 * 4; // 4 WGs of 256 per CU -> 4 waves per SIMD
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<OP><<<blocks, 256>>>(d, 12345u, 10); hipDeviceSynchronize();
  hipEventRecord(a); k<OP><<<blocks, 256>>>(d, 12345u, iters); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double instr = (double)blocks * 4 /*waves*/ * iters * 32.0; // wave-instructions
  // per SIMD: blocks*4 waves / (256 CUs * 4 SIMDs) waves per SIMD
  const double per_simd = instr / (256.0 * 4.0);
  printf("%-16s %8.3f ms  %6.2f ns per wave-instruction per SIMD  (x clock GHz = cycles)\n", name, ms, ms * 1e6 / per_simd);
}
int main() {
  uint32_t* d; hipMalloc(&d, 256 * 4 * 256 * 4);
  run<5>("v_fma_f32", d); run<2>("v_cvt_i32_f32", d); run<3>("v_fract_f32", d); run<4>("v_lshl_add_u32", d);
  run<0>("v_mad_u32_u16", d); run<1>("v_min3_u16", d); run<6>("v_bfe_u32", d); run<7>("v_min3_f32", d);
  run<8>("v_mad_u32_u24", d); run<9>("v_perm_b32", d); run<10>("v_cndmask_b32 vcc", d); run<11>("v_cndmask_e64 sgpr", d); run<19>("v_cndmask_e64 vcc", d); run<20>("v_cndmask_e32 swapped", d); run<21>("v_addc_co vcc", d); run<22>("cmp vcc + cnd e32 (2)", d); run<23>("cmp sgpr + cnd e64 (2)", d); run<24>("cmp vcc + cnd e64 vcc (2)", d);
  run<12>("v_min_f32", d); run<13>("v_and_b32", d); run<14>("v_cmp+v_add (2)", d); run<15>("v_bfi_b32", d); run<16>("v_add_u32", d); run<5>("v_fma_f32 again", d);
  run<17>("v_cvt_f32_f64", d); run<18>("v_fma_f64", d);
  return 0;
}
