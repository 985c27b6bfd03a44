#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s\n", hipGetErrorString(e)); return 1;}}while(0)
template<int OP> __global__ __launch_bounds__(256) void k(float* out, float a, float b, int iters, unsigned* lds_dummy) {
  __shared__ unsigned acc[4096];
  float x[8]; int ii[8];
  for (int u=0;u<8;u++){ x[u] = threadIdx.x*0.001f + u; ii[u]=u+threadIdx.x; }
  for (int i = threadIdx.x; i < 4096; i += 256) acc[i]=0;
  __syncthreads();
  unsigned s = 0;
  for (int it=0; it<iters; it++) {
#pragma unroll
    for (int u=0;u<8;u++) {
      if (OP==0) x[u] = __builtin_fmaf(x[u], a, b);
      if (OP==1) { ii[u] = (int)x[u]; x[u] = x[u] + (float)(ii[u]&1); }              // cvt_i32 + (and, cvt_f32, add)
      if (OP==2) x[u] = __builtin_amdgcn_fractf(x[u]) + a;                           // fract + add
      if (OP==3) { s += (x[u] < b) ? 1u : 0u; x[u] += a; }                           // cmp + cndmask/add
      if (OP==4) { ii[u] = (ii[u] << 2) + ii[(u+1)&7]; }                              // lshl_add
      if (OP==5) { x[u] = x[u] + a; }                                                 // add baseline
      if (OP==6) { atomicAdd(&acc[(threadIdx.x*33 + u*257 + it) & 4095], 1u); }      // ds_add conflict-light
      if (OP==7) { ii[u] = (int)x[u]; x[u] = x[u]+a; }                                // cvt_i32 + add
      if (OP==8) { asm volatile("v_fract_f32 %0, %1" : "=v"(x[u]) : "v"(x[u])); }    // pure fract chain
      if (OP==9) { asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(ii[u]) : "v"(x[u])); } // pure cvt
    }
  }
  float r=0; for (int u=0;u<8;u++) r += x[u] + ii[u];
  out[blockIdx.x*256+threadIdx.x] = r + s + acc[threadIdx.x];
}
template<int OP> float run(float* d, int iters){
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<256*8, 256>>>(d, 1.0001f, 0.5f, 10, nullptr); hipDeviceSynchronize();
  hipEventRecord(e0); k<OP><<<256*8, 256>>>(d, 1.0001f, 0.5f, iters, nullptr); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms,e0,e1); return ms;
}
int main(){ float* d; CHK(hipMalloc(&d, 256*8*256*4)); int iters=20000;
  const char* names[]={"fma","cvt_i32+and+cvt_f32+add","fract+add","cmp+cndmask+add(+add)","lshl_add","add","ds_add","cvt_i32+add","fract(asm)","cvt_i32(asm)"};
  float t[10]; t[0]=run<0>(d,iters);t[1]=run<1>(d,iters);t[2]=run<2>(d,iters);t[3]=run<3>(d,iters);t[4]=run<4>(d,iters);t[5]=run<5>(d,iters);t[6]=run<6>(d,iters);t[7]=run<7>(d,iters);t[8]=run<8>(d,iters);t[9]=run<9>(d,iters);
  // per-SIMD: blocks=2048 x 4 waves = 8192 waves over 1024 SIMDs = 8 waves/SIMD sequentially-ish (2 blocks/CU resident? 8 blocks/CU)
  for(int i=0;i<10;i++){ double wave_instr = (double)iters*8*8192; double cyc = t[i]*1e-3*2.4e9*1024; printf("%-28s %8.3f ms  -> %.2f cycles per wave-iteration-op-group per SIMD\n", names[i], t[i], cyc/wave_instr); }
  return 0; }
