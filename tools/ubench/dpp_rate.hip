// Microbenchmark: issue cost of the strip kernel's multiply-add flavours, W waves per SIMD, 512 per timed block, 16 chains:
//   plain v_fmac_f32_e32, v_fmac_f32_dpp wave_shr:1, v_fmac_f32_dpp row_shr:1, and the strip kernel's quarter pattern
//   (4 plain + 8 DPP on four accumulators).    hipcc --offload-arch=gfx950 -O3 -o dpp_rate dpp_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

#define DPPW " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define DPPL " wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define DPPR " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cyc, int reps) {
  float a[16], s[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x + i; s[i] = threadIdx.x * 0.5f + i; }
  float w = 1.0001f, u = 0.5f, v = 0.25f;
  asm volatile("" : "+v"(w), "+v"(u), "+v"(v));
  unsigned long long total = 0;
  for (int r = 0; r < reps; ++r) {
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(s[i]), "v"(u));
      } else if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32_dpp %0, %1, %2" DPPW : "+v"(a[i]) : "v"(s[i]), "v"(u));
      } else if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32_dpp %0, %1, %2" DPPR : "+v"(a[i]) : "v"(s[i]), "v"(u));
      } else {  // the quarter pattern: per group of four accumulators 4 centre, 4 west, 4 east  (16 groups of 12 -> scale below)
#pragma unroll
        for (int i = 0; i < 16; i += 4)
          asm volatile(
              "v_fmac_f32_e32 %0, %4, %9\n\tv_fmac_f32_e32 %1, %5, %9\n\tv_fmac_f32_e32 %2, %6, %9\n\tv_fmac_f32_e32 %3, %7, %9\n\t"
              "v_fmac_f32_dpp %0, %4, %8" DPPW "\n\tv_fmac_f32_dpp %1, %5, %8" DPPW "\n\tv_fmac_f32_dpp %2, %6, %8" DPPW "\n\tv_fmac_f32_dpp %3, %7, %8" DPPW "\n\t"
              "v_fmac_f32_dpp %0, %4, %10" DPPL "\n\tv_fmac_f32_dpp %1, %5, %10" DPPL "\n\tv_fmac_f32_dpp %2, %6, %10" DPPL "\n\tv_fmac_f32_dpp %3, %7, %10" DPPL
              : "+v"(a[i]), "+v"(a[i + 1]), "+v"(a[i + 2]), "+v"(a[i + 3])
              : "v"(s[i]), "v"(s[i + 1]), "v"(s[i + 2]), "v"(s[i + 3]), "v"(w), "v"(u), "v"(v));
      }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    total += t1 - t0;
  }
  float q = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) q += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = q;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = total / reps;
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 8);
  const char* names[4] = {"v_fmac_f32_e32", "v_fmac_f32_dpp wave_shr", "v_fmac_f32_dpp row_shr", "quarter pattern (4 plain + 8 dpp)"};
  for (int mode = 0; mode < 4; ++mode)
    for (int threads : {256, 512, 1024}) {
      if (mode == 0) k<0><<<256, threads>>>(out, cyc, 20);
      if (mode == 1) k<1><<<256, threads>>>(out, cyc, 20);
      if (mode == 2) k<2><<<256, threads>>>(out, cyc, 20);
      if (mode == 3) k<3><<<256, threads>>>(out, cyc, 20);
      hipDeviceSynchronize();
      unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      const int w = threads / 256;
      const double n = mode == 3 ? 32.0 * 48 : 512.0;
      printf("%-36s %d wave(s)/SIMD: %5llu cycles for %4.0f -> %.2f cycles per instruction per wave, %.2f per SIMD\n", names[mode], w, h, n, h / n, h / n / w);
    }
  return 0;
}
