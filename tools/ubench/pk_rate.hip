// Microbenchmark: issue cost of v_pk_fma_f32 (two channels per instruction, the coefficient broadcast by op_sel) against two
// v_fmac_f32, 1 and 2 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -o pk_rate pk_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int reps) {
  f2 a[8], s[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = f2{(float)threadIdx.x + i, 1.f}; s[i] = f2{threadIdx.x * 0.5f + i, 2.f}; }
  f2 c = f2{0.5f, 0.25f};
  asm volatile("" : "+v"(c));
  unsigned long long total = 0;
  for (int r = 0; r < reps; ++r) {
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll
    for (int j = 0; j < 32; ++j) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "v"(s[i]), "v"(c));
        else {
          float lo = a[i][0], hi = a[i][1];
          asm volatile("v_fmac_f32_e32 %0, %2, %4\n\tv_fmac_f32_e32 %1, %3, %4" : "+v"(lo), "+v"(hi) : "v"(s[i][0]), "v"(s[i][1]), "v"(c[0]));
          a[i][0] = lo; a[i][1] = hi;
        }
      }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    total += t1 - t0;
  }
  float q = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) q += a[i][0] + a[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = q;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = total / reps;
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 512 * 256 * 4); hipMalloc(&cyc, 8);
  for (int mode = 0; mode < 2; ++mode)
    for (int threads : {256, 512}) {
      if (mode == 0) k<0><<<256, threads>>>(out, cyc, 20);
      else k<1><<<256, threads>>>(out, cyc, 20);
      hipDeviceSynchronize();
      unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      printf("%-28s %d wave(s)/SIMD: %5llu cycles for 512 channel multiply-adds -> %.2f cycles per pair of channels\n",
             mode == 0 ? "v_pk_fma_f32 (256 instr)" : "2 x v_fmac_f32 (512 instr)", threads / 256, h, h / 256.0);
    }
  return 0;
}
