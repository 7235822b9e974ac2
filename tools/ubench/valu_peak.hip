// Microbenchmark: v_fma_f32 throughput per SIMD with W waves per SIMD, 512 FMAs per timed block, fully unrolled (no loop
// overhead inside the timed region; valu_rate.hip's 8-instruction loop measured mostly its own branch), 16 independent
// accumulation chains per lane.   hipcc --offload-arch=gfx950 -O3 -o valu_peak valu_peak.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cyc, int reps) {
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x + i;
  const float w = 1.0001f, u = 0.5f;
  unsigned long long total = 0;
  for (int r = 0; r < reps; ++r) {
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll
    for (int j = 0; j < 32; ++j) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(w), "v"(u));
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    total += t1 - t0;
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = total / reps;
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 8);
  for (int threads : {256, 512, 768, 1024}) {
    k<<<256, threads>>>(out, cyc, 20);
    hipDeviceSynchronize();
    unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const int w = threads / 256;
    printf("%d wave(s) per SIMD: 512 v_fma_f32 per wave in %llu cycles -> %.2f cycles per instruction per wave, %.2f per SIMD, %.1f FMA lanes per clock and SIMD\n",
           w, h, h / 512.0, h / 512.0 / w, 512.0 * 64 * w / h);
  }
  return 0;
}
