// Microbenchmark: does the per-instruction cost of a straight-line stream of 8-byte vector instructions (v_fmac_f32_dpp)
// depend on how much code the loop body holds?  BODY = instructions per loop iteration (8 bytes each); 2 waves per SIMD
// (the strip kernel's occupancy), every wave runs the same body.   hipcc --offload-arch=gfx950 -O3 -o icache_rate icache_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define DPPW " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"

template <int BODY16>  // body = 16 * BODY16 instructions
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
  float a[16], s[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x + i; s[i] = threadIdx.x * 0.5f + i; }
  float u = 0.5f;
  asm volatile("" : "+v"(u));
  __syncthreads();
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int r = 0; r < iters; ++r) {
#pragma unroll
    for (int j = 0; j < BODY16; ++j) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32_dpp %0, %1, %2" DPPW : "+v"(a[i]) : "v"(s[i]), "v"(u));
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float q = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) q += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = q;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int B> void run(float* out, unsigned long long* cyc) {
  const int total = 1 << 16;  // instructions per wave
  const int iters = total / (16 * B);
  k<B><<<256, 512>>>(out, cyc, iters);
  hipDeviceSynchronize();
  unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("body %6d instructions (%4d KiB): %.2f cycles per instruction per wave\n", 16 * B, 16 * B * 8 / 1024, (double)h / (iters * 16.0 * B));
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 512 * 256 * 4); hipMalloc(&cyc, 8);
  run<8>(out, cyc); run<32>(out, cyc); run<128>(out, cyc); run<256>(out, cyc); run<512>(out, cyc); run<1024>(out, cyc);
  return 0;
}
