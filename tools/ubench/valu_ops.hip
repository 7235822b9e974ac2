// Microbenchmark: cycles per VALU instruction of one wave's stream (8 independent chains), by opcode / operand kind.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define KERNEL(NAME, STR)                                                                                         \
  __global__ __launch_bounds__(1024) void NAME(float* out, int iters, unsigned long long* cyc, float sw) {         \
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    float w = 1.0001f + threadIdx.x * 1e-9f, u = 0.5f;                                                                 \
    unsigned long long t0, t1;                                                                                     \
    __syncthreads();                                                                                               \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");                                      \
    for (int i = 0; i < iters; ++i)                                                                                \
      asm volatile(STR : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w), "v"(u), "s"(sw)); \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");                                      \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                            \
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;                                                     \
  }
KERNEL(k_fma3, "v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n")
KERNEL(k_fmac, "v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n")
KERNEL(k_fmas, "v_fma_f32 %0, %10, %9, %0\n v_fma_f32 %1, %10, %9, %1\n v_fma_f32 %2, %10, %9, %2\n v_fma_f32 %3, %10, %9, %3\n v_fma_f32 %4, %10, %9, %4\n v_fma_f32 %5, %10, %9, %5\n v_fma_f32 %6, %10, %9, %6\n v_fma_f32 %7, %10, %9, %7\n")
KERNEL(k_add, "v_add_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n")
KERNEL(k_mov, "v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n")
KERNEL(k_nop, "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n")
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(1024) void k_pk(float* out, int iters, unsigned long long* cyc, float sw) {
  v2f a0 = {threadIdx.x + 0.f, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  v2f w = {1.0001f + threadIdx.x * 1e-9f, 0.9999f}, u = {0.5f, 0.25f};
  unsigned long long t0, t1;
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int i = 0; i < iters; ++i)
    asm volatile("v_pk_fma_f32 %0, %8, %9, %0 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %1, %8, %9, %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
                 "v_pk_fma_f32 %2, %8, %9, %2 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %3, %8, %9, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
                 "v_pk_fma_f32 %4, %8, %9, %4 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %5, %8, %9, %5 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
                 "v_pk_fma_f32 %6, %8, %9, %6 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %7, %8, %9, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w), "v"(u));
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  v2f s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// two dependent pk_fma back to back on ONE chain (is a wait state needed? does the result still come out right?)
__global__ __launch_bounds__(64) void k_pkdep(float* out) {
  v2f a = {1.f, 2.f}, w = {3.f, 5.f}, u = {7.f, 11.f};
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n" : "+v"(a) : "v"(w), "v"(u));
  if (threadIdx.x == 0) { out[0] = a.x; out[1] = a.y; }
}
template <int CH>
__global__ __launch_bounds__(1024) void k_pkchain(float* out, int iters, unsigned long long* cyc) {
  v2f a[8];
  for (int i = 0; i < 8; ++i) a[i] = v2f{threadIdx.x + (float)i, 1.f};
  v2f w = {1.0001f, 0.9999f}, u = {0.5f, 0.25f};
  unsigned long long t0, t1;
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8 / CH; ++r)
#pragma unroll
      for (int c = 0; c < CH; ++c) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a[c]) : "v"(w), "v"(u));
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  v2f s = a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int CH>
__global__ __launch_bounds__(1024) void k_fmachain(float* out, int iters, unsigned long long* cyc) {
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + (float)i;
  float w = 1.0001f, u = 0.5f;
  unsigned long long t0, t1;
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8 / CH; ++r)
#pragma unroll
      for (int c = 0; c < CH; ++c) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[c]) : "v"(w), "v"(u));
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  out[blockIdx.x * blockDim.x + threadIdx.x] = a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
#define RUNCH(K, CH, NAME) for (int threads : {256, 768}) { hipLaunchKernelGGL(K<CH>, dim3(256), dim3(threads), 0, 0, out, iters, cyc); hipDeviceSynchronize(); unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); \
    printf("%s, %d independent chains, %d waves/SIMD: %.2f ticks per instruction per wave\n", NAME, CH, threads / 256, (double)h / (iters * 8.0)); }
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 16);
  const int iters = 4096;
  struct { const char* n; void (*f)(float*, int, unsigned long long*, float); } ks[] = {{"v_fma_f32 (3 vgpr)", k_fma3}, {"v_fmac_f32", k_fmac}, {"v_fma_f32 (sgpr mul)", k_fmas}, {"v_add_f32", k_add}, {"v_mov_b32", k_mov}, {"s_nop 0", k_nop}};
  { hipLaunchKernelGGL(k_pkdep, dim3(1), dim3(64), 0, 0, out); hipDeviceSynchronize(); float h[2]; hipMemcpy(h, out, 8, hipMemcpyDeviceToHost);
    printf("dependent pk_fma pair: got (%g, %g), expected (%g, %g)\n", h[0], h[1], (1.f + 3 * 7) + 5 * 7, (2.f + 3 * 11) + 5 * 11); }
  for (int threads : {256, 512, 768}) { hipLaunchKernelGGL(k_pk, dim3(256), dim3(threads), 0, 0, out, iters, cyc, 1.f); hipDeviceSynchronize(); unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("v_pk_fma_f32 (op_sel bcast) %d waves/SIMD: %.2f ticks per instruction per wave\n", threads / 256, (double)h / (iters * 8.0)); }
  RUNCH(k_pkchain, 1, "v_pk_fma_f32") RUNCH(k_pkchain, 2, "v_pk_fma_f32") RUNCH(k_pkchain, 4, "v_pk_fma_f32") RUNCH(k_pkchain, 8, "v_pk_fma_f32")
  RUNCH(k_fmachain, 1, "v_fma_f32") RUNCH(k_fmachain, 2, "v_fma_f32") RUNCH(k_fmachain, 4, "v_fma_f32")
  for (auto& k : ks)
    for (int threads : {256, 512, 768}) {
      hipLaunchKernelGGL(k.f, dim3(256), dim3(threads), 0, 0, out, iters, cyc, 1.0001f);
      hipDeviceSynchronize();
      unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      printf("%-22s %d waves/SIMD: %.2f ticks per instruction per wave\n", k.n, threads / 256, (double)h / (iters * 8.0));
    }
  return 0;
}
