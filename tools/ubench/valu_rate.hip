// Microbenchmark: VALU issue rate per SIMD with W waves per SIMD, scalar v_fma_f32 vs packed v_pk_fma_f32 (one weight
// register broadcast to both halves by op_sel), independent accumulation chains.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int PK>
__global__ __launch_bounds__(1024) void k(float* out, int iters, unsigned long long* cyc) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float w = 1.0001f, u0 = 0.5f, u1 = 0.25f;
  unsigned long long t0, t1;
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int i = 0; i < iters; ++i) {
    if (PK) {
      asm volatile(
          "v_pk_fma_f32 %0, %8, %9, %0 op_sel_hi:[0,1,1]\n\t"
          "v_pk_fma_f32 %2, %8, %9, %2 op_sel_hi:[0,1,1]\n\t"
          "v_pk_fma_f32 %4, %8, %9, %4 op_sel_hi:[0,1,1]\n\t"
          "v_pk_fma_f32 %6, %8, %9, %6 op_sel_hi:[0,1,1]\n\t"
          : "+v"(*(double*)&a0), "+v"(a1), "+v"(*(double*)&a2), "+v"(a3), "+v"(*(double*)&a4), "+v"(a5), "+v"(*(double*)&a6), "+v"(a7)
          : "v"(*(double*)&w), "v"(*(double*)&u0));
    } else {
      asm volatile(
          "v_fma_f32 %0, %8, %9, %0\n\tv_fma_f32 %1, %8, %10, %1\n\tv_fma_f32 %2, %8, %9, %2\n\tv_fma_f32 %3, %8, %10, %3\n\t"
          "v_fma_f32 %4, %8, %9, %4\n\tv_fma_f32 %5, %8, %10, %5\n\tv_fma_f32 %6, %8, %9, %6\n\tv_fma_f32 %7, %8, %10, %7\n\t"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
          : "v"(w), "v"(u0), "v"(u1));
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main2();
int main() {
  main2();
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 8);
  const int iters = 4096;
  for (int threads : {256, 512, 768, 1024})
    for (int pk : {0, 1}) {
      if (pk) k<1><<<256, threads>>>(out, iters, cyc); else k<0><<<256, threads>>>(out, iters, cyc);
      hipDeviceSynchronize();
      unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      const double fma_per_lane = pk ? 8.0 * iters : 8.0 * iters;   // pk: 4 instr x 2 FMAs
      const int waves_per_simd = threads / 256;
      printf("%d waves/SIMD %s: %llu cycles for %d x %d instr per wave -> %.2f cycles per instr per wave, %.2f cycles per instr per SIMD, %.1f FMA/clk/SIMD\n",
             waves_per_simd, pk ? "v_pk_fma_f32" : "v_fma_f32   ", h, iters, pk ? 4 : 8, (double)h / (iters * (pk ? 4 : 8)),
             (double)h / (iters * (pk ? 4 : 8) * waves_per_simd), fma_per_lane * 64 * waves_per_simd / (double)h);
    }
  return 0;
}
// plain C++ variant: 16 independent chains per lane, the compiler schedules
__global__ __launch_bounds__(1024) void kc(float* out, int iters, unsigned long long* cyc, float w, float u) {
  float a[16];
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x + i;
  unsigned long long t0, t1;
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = fmaf(w, u, a[j]) ;
#pragma unroll
    for (int j = 0; j < 16; ++j) asm volatile("" : "+v"(a[j]));
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0; for (int j = 0; j < 16; ++j) s += a[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main2() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 16);
  for (int blocks : {1, 256})
  for (int threads : {256, 512, 768, 1024}) {
    kc<<<blocks, threads>>>(out, 4096, cyc, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("C++ %d blocks, %d waves/SIMD: %llu cycles for 4096 x 16 v_fma per wave -> %.2f cycles per instr per wave\n", blocks, threads / 256, h, (double)h / (4096 * 16));
  }
  return 0;
}
