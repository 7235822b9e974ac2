// Microbenchmark: does a multiply-add whose accumulator and source register share a VGPR bank (index mod 4) issue more
// slowly?  acc and src are 16-register tuples pinned to fixed registers; MODE = distance of the bases mod 4.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16 __attribute__((ext_vector_type(16)));

#define DPPW " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define Q(A, S, C0, C1, C2)                                                                                                   \
  "v_fmac_f32_e32 v" #A "0, v" #S "0, " C1 "\n\tv_fmac_f32_e32 v" #A "1, v" #S "1, " C1 "\n\tv_fmac_f32_e32 v" #A "2, v" #S "2, " C1 "\n\tv_fmac_f32_e32 v" #A "3, v" #S "3, " C1 "\n\t" \
  "v_fmac_f32_dpp v" #A "0, v" #S "0, " C0 DPPW "\n\tv_fmac_f32_dpp v" #A "1, v" #S "1, " C0 DPPW "\n\tv_fmac_f32_dpp v" #A "2, v" #S "2, " C0 DPPW "\n\tv_fmac_f32_dpp v" #A "3, v" #S "3, " C0 DPPW "\n\t" \
  "v_fmac_f32_dpp v" #A "0, v" #S "0, " C2 DPPW "\n\tv_fmac_f32_dpp v" #A "1, v" #S "1, " C2 DPPW "\n\tv_fmac_f32_dpp v" #A "2, v" #S "2, " C2 DPPW "\n\tv_fmac_f32_dpp v" #A "3, v" #S "3, " C2 DPPW "\n\t"

// registers: acc v40..v43 (written as v4 + digit), sources: v60..63 (same bank), v61..64?  -> use explicit literal blocks instead
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cyc, int reps) {
  unsigned long long total = 0;
  float r = threadIdx.x;
  // initialise the registers used below
  asm volatile("v_mov_b32 v40, %0\n\tv_mov_b32 v41, %0\n\tv_mov_b32 v42, %0\n\tv_mov_b32 v43, %0\n\t"
               "v_mov_b32 v60, %0\n\tv_mov_b32 v61, %0\n\tv_mov_b32 v62, %0\n\tv_mov_b32 v63, %0\n\tv_mov_b32 v64, %0\n\tv_mov_b32 v65, %0\n\tv_mov_b32 v66, %0\n\t"
               "v_mov_b32 v80, 0.5\n\tv_mov_b32 v81, 0.5\n\tv_mov_b32 v82, 0.5\n\tv_mov_b32 v83, 0.5"
               : : "v"(r) : "v40", "v41", "v42", "v43", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v80", "v81", "v82", "v83");
  for (int rr = 0; rr < reps; ++rr) {
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      if (MODE == 0)  // acc v40.., src v60.. (same bank), coefficients v80, v81, v82 (v80: same bank as both)
        asm volatile(Q(4, 6, "v80", "v81", "v82") ::: "v40", "v41", "v42", "v43");
      else if (MODE == 1)  // coefficients in other banks only
        asm volatile(Q(4, 6, "v81", "v82", "v83") ::: "v40", "v41", "v42", "v43");
      else {  // src shifted by one register: v61.. for acc v40.. (different bank), coefficients v82, v83, v82
        asm volatile(
            "v_fmac_f32_e32 v40, v61, v82\n\tv_fmac_f32_e32 v41, v62, v83\n\tv_fmac_f32_e32 v42, v63, v80\n\tv_fmac_f32_e32 v43, v64, v81\n\t"
            "v_fmac_f32_dpp v40, v61, v82" DPPW "\n\tv_fmac_f32_dpp v41, v62, v83" DPPW "\n\tv_fmac_f32_dpp v42, v63, v80" DPPW "\n\tv_fmac_f32_dpp v43, v64, v81" DPPW "\n\t"
            "v_fmac_f32_dpp v40, v61, v82" DPPW "\n\tv_fmac_f32_dpp v41, v62, v83" DPPW "\n\tv_fmac_f32_dpp v42, v63, v80" DPPW "\n\tv_fmac_f32_dpp v43, v64, v81" DPPW
            ::: "v40", "v41", "v42", "v43");
      }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    total += t1 - t0;
  }
  float q;
  asm volatile("v_add_f32 %0, v40, v41\n\tv_add_f32 %0, %0, v42\n\tv_add_f32 %0, %0, v43" : "=v"(q));
  out[blockIdx.x * blockDim.x + threadIdx.x] = q;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = total / reps;
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 8);
  const char* names[3] = {"acc/src same bank, one coef same bank", "acc/src same bank, coefs elsewhere", "acc/src/coef all different banks"};
  for (int mode = 0; mode < 3; ++mode)
    for (int threads : {256, 512}) {
      if (mode == 0) k<0><<<256, threads>>>(out, cyc, 20);
      if (mode == 1) k<1><<<256, threads>>>(out, cyc, 20);
      if (mode == 2) k<2><<<256, threads>>>(out, cyc, 20);
      hipDeviceSynchronize();
      unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      printf("%-42s %d wave(s)/SIMD: %5llu cycles for 384 -> %.2f cycles per instruction per wave\n", names[mode], threads / 256, h, h / 384.0);
    }
  return 0;
}
