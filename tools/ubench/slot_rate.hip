// Microbenchmark: the strip kernel's slot pattern in isolation -- one dependent MFMA (32x32x16 bf16) followed by NQ
// quarters of 12 multiply-adds (4 plain + 8 DPP), operands in registers, no LDS -- at 1 and 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o slot_rate slot_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define DPPW " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define DPPL " wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"

__device__ __forceinline__ void q4(float& a0, float& a1, float& a2, float& a3, float s0, float s1, float s2, float s3, float cw, float cc, float ce) {
  asm volatile(
      "v_fmac_f32_e32 %0, %4, %9\n\tv_fmac_f32_e32 %1, %5, %9\n\tv_fmac_f32_e32 %2, %6, %9\n\tv_fmac_f32_e32 %3, %7, %9\n\t"
      "v_fmac_f32_dpp %0, %4, %8" DPPW "v_fmac_f32_dpp %1, %5, %8" DPPW "v_fmac_f32_dpp %2, %6, %8" DPPW "v_fmac_f32_dpp %3, %7, %8" DPPW
      "v_fmac_f32_dpp %0, %4, %10" DPPL "v_fmac_f32_dpp %1, %5, %10" DPPL "v_fmac_f32_dpp %2, %6, %10" DPPL "v_fmac_f32_dpp %3, %7, %10" DPPL
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
      : "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(cw), "v"(cc), "v"(ce)
      : "memory");
}

template <int NQ>  // quarters per MFMA
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int reps) {
  f32x16 acc, b, s;
  bf16x8 wa, wb;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc[i] = 0.f; b[i] = threadIdx.x + i; s[i] = 0.5f * threadIdx.x + i; }
#pragma unroll
  for (int i = 0; i < 8; ++i) { wa[i] = (__bf16)(0.001f * (threadIdx.x + i)); wb[i] = (__bf16)(0.002f * i); }
  float cw = 0.3f, cc = 0.4f, ce = 0.5f;
  asm volatile("" : "+v"(cw), "+v"(cc), "+v"(ce));
  unsigned long long total = 0;
  for (int r = 0; r < reps; ++r) {
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll
    for (int m = 0; m < 24; ++m) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(wa), "v"(wb) : "memory");
#pragma unroll
      for (int qq = 0; qq < NQ; ++qq) {
        const int k4 = 4 * ((m * NQ + qq) & 3);
        float a0 = b[k4], a1 = b[k4 + 1], a2 = b[k4 + 2], a3 = b[k4 + 3];
        q4(a0, a1, a2, a3, s[k4], s[k4 + 1], s[k4 + 2], s[k4 + 3], cw, cc, ce);
        b[k4] = a0; b[k4 + 1] = a1; b[k4 + 2] = a2; b[k4 + 3] = a3;
      }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    total += t1 - t0;
  }
  float q = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) q += acc[i] + b[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = q;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = total / reps;
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 512 * 256 * 4); hipMalloc(&cyc, 8);
  for (int nq = 0; nq < 3; ++nq)
    for (int threads : {256, 512}) {
      if (nq == 0) k<0><<<256, threads>>>(out, cyc, 20);
      if (nq == 1) k<1><<<256, threads>>>(out, cyc, 20);
      if (nq == 2) k<2><<<256, threads>>>(out, cyc, 20);
      hipDeviceSynchronize();
      unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      printf("1 MFMA + %2d multiply-adds per group, %d wave(s)/SIMD: %6.1f cycles per group (24 groups in %llu)\n", 12 * nq, threads / 256, h / 24.0, h);
    }
  return 0;
}
