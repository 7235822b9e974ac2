// Microbenchmark: what TWO waves on one SIMD get when both issue the same stream (the other *_rate files time wave 0 only,
// which as the oldest wave wins the arbitration).  Every wave stamps its own start and end; reported: the span from the
// first start to the last end over the waves of SIMD 0, per instruction group, for 1 and 2 waves per SIMD.
//   MODE 0: 16 independent v_fmac_f32        MODE 1: 8 v_pk_fma_f32 (the same 16 multiply-adds)
//   MODE 2: 1 MFMA 32x32x16 bf16 + 12 v_fmac MODE 3: 1 MFMA alone      MODE 4: 1 MFMA + 24 v_fmac
//   MODE 5: 16 v_fmac_f32_dpp wave_shr:1      MODE 6: the strip kernel's quarter (4 plain + 8 dpp) + 2 more plain (14)
//   MODE 7: the quarter with its centre terms packed (2 v_pk_fma + 8 dpp)
// hipcc --offload-arch=gfx950 -O3 -o simd_share simd_share.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f16v __attribute__((ext_vector_type(16)));
typedef short s8v __attribute__((ext_vector_type(8)));
typedef float f2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* stamps, int groups) {
  float a[24], s[24];
  f2v pa[8], ps[8];
#pragma unroll
  for (int i = 0; i < 24; ++i) { a[i] = threadIdx.x + i; s[i] = threadIdx.x * 0.5f + i; }
#pragma unroll
  for (int i = 0; i < 8; ++i) { pa[i] = f2v{a[i], a[i + 8]}; ps[i] = f2v{s[i], s[i + 8]}; }
  f2v pu = {0.5f, 0.25f};
  float u = 0.5f;
  asm volatile("" : "+v"(u), "+v"(pu));
  f16v acc0 = {0}, acc1 = {0};
  s8v fa = {1, 2, 3, 4, 5, 6, 7, 8}, fb = {1, 1, 1, 1, 1, 1, 1, 1};
  asm volatile("" : "+v"(fa), "+v"(fb));
  __syncthreads();
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int g = 0; g < groups; g += 2) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (MODE >= 2) {
        if (h == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(fa), "v"(fb));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(fa), "v"(fb));
      }
      if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(s[i]), "v"(u));
      } else if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pa[i]) : "v"(ps[i]), "v"(pu));
      } else if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 12; ++i) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(s[i]), "v"(u));
      } else if (MODE == 5) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
          asm volatile("v_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(s[i]), "v"(u));
      } else if (MODE == 6 || MODE == 7) {
#pragma unroll
        for (int i = 0; i < 4; i += 4) {
          if (MODE == 6) {
#pragma unroll
            for (int e = 0; e < 4; ++e) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i + e]) : "v"(s[i + e]), "v"(u));
          } else {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(pa[0]) : "v"(ps[0]), "v"(pu));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(pa[1]) : "v"(ps[1]), "v"(pu));
          }
#pragma unroll
          for (int e = 0; e < 4; ++e)
            asm volatile("v_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[4 + e]) : "v"(s[4 + e]), "v"(u));
#pragma unroll
          for (int e = 0; e < 4; ++e)
            asm volatile("v_fmac_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[8 + e]) : "v"(s[8 + e]), "v"(u));
        }
      } else if (MODE == 4) {
#pragma unroll
        for (int i = 0; i < 24; ++i) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(s[i]), "v"(u));
      }
    }
  }
  asm volatile("s_nop 0" ::: "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float q = 0;
#pragma unroll
  for (int i = 0; i < 24; ++i) q += a[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) q += pa[i].x + pa[i].y;
#pragma unroll
  for (int i = 0; i < 16; ++i) q += acc0[i] + acc1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = q;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) {
    stamps[2 * (threadIdx.x >> 6)] = t0;
    stamps[2 * (threadIdx.x >> 6) + 1] = t1;
  }
}

int main() {
  float* out; unsigned long long* st;
  hipMalloc(&out, 512 * 256 * 4); hipMalloc(&st, 16 * 8);
  const char* names[8] = {"16 v_fmac_f32", "8 v_pk_fma_f32 (16 multiply-adds)", "1 MFMA + 12 v_fmac", "1 MFMA alone", "1 MFMA + 24 v_fmac",
                          "16 v_fmac_f32_dpp", "quarter: 4 plain + 8 dpp", "quarter: 2 pk + 8 dpp"};
  const int groups = 256;
  for (int mode = 0; mode < 8; ++mode)
    for (int threads : {256, 512}) {
      for (int rep = 0; rep < 2; ++rep) {
        if (mode == 0) k<0><<<256, threads>>>(out, st, groups);
        if (mode == 1) k<1><<<256, threads>>>(out, st, groups);
        if (mode == 2) k<2><<<256, threads>>>(out, st, groups);
        if (mode == 3) k<3><<<256, threads>>>(out, st, groups);
        if (mode == 4) k<4><<<256, threads>>>(out, st, groups);
        if (mode == 5) k<5><<<256, threads>>>(out, st, groups);
        if (mode == 6) k<6><<<256, threads>>>(out, st, groups);
        if (mode == 7) k<7><<<256, threads>>>(out, st, groups);
        hipDeviceSynchronize();
      }
      unsigned long long h[16]; hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
      const int nw = threads / 64;
      // waves w and w + 4 share a SIMD (w % 4)
      unsigned long long lo = h[0], hi = h[1];
      if (nw == 8) { lo = h[0] < h[8] ? h[0] : h[8]; hi = h[1] > h[9] ? h[1] : h[9]; }
      printf("%-36s %d wave(s)/SIMD: span %6llu cycles / %d groups = %6.1f per group per SIMD; wave 0 alone %6.1f", names[mode], nw / 4,
             hi - lo, groups, (double)(hi - lo) / groups, (double)(h[1] - h[0]) / groups);
      if (nw == 8) printf(", wave 4 %6.1f", (double)(h[9] - h[8]) / groups);
      printf("\n");
    }
  return 0;
}
