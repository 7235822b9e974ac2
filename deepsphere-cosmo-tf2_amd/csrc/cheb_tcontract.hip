// The dense contraction over K planes without an LDS staging of the planes (round 4):
//     y[n, m, o] = act( sum_k sum_f planes[k][n, m, f] * w[f * K + k, o] + bias[o] )            (reference gnn_layers.py:144-150)
// A plane row of 16 channels is 64 contiguous bytes and 32 consecutive rows are 2 KiB: a wave loads them straight into the B-operand
// layout of v_mfma_f32_32x32x16_bf16 -- lane = (row of the block, half g), 8 channels per lane, two 16-byte loads -- splits them
// to bf16 in registers and contracts (cheb_istrip_kernel.h, is_contract: the three arithmetics of the fused kernels; the unfused
// path used to run exact fp32 whatever the layer asked for, padded to 32 channels per order).  The weights are turned into
// A-operand fragments in LDS by the workgroup itself (a few KiB), the accumulator tile (32 rows x 32 columns, lane = row) is stored
// with 16-byte stores.  HBM-bound: K planes in, y out.  Used by the unfused path (graphs the fused kernels do not take) whenever
// Fin * K > 64 and the fragments fit the LDS (channel counts that are not multiples of four: scalar loads); cheb_contract.hip otherwise.
#include <algorithm>

#include "cheb_istrip_kernel.h"

namespace dsph {

constexpr int TC_THREADS = 512;
constexpr int TC_KMAX = 64;
constexpr int TC_LDS_MAX = 96 * 1024;

struct TContractArgs {
  const float* p[TC_KMAX];
  const float* w;
  const float* bias;
  float* y;
  int64_t plane_rows, rows;
  int N, Fin, Fout, K, act;
};

template <int PREC, int NB, bool VEC>  // VEC: channel count a multiple of four, planes 16-byte aligned
__global__ __launch_bounds__(TC_THREADS, 2) void cheb_tcontract_kernel(TContractArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tc_smem[];
  constexpr int TERMS = is_terms(PREC), TB = is_term_bytes(PREC), LEVB = TERMS * TB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 31, g = lane >> 5;
  const int C = (a.Fin + 15) / 16, NBT = (a.Fout + 31) / 32;
  // fragments [slice c][order k][column block nb][term]: element (lane l, slot j) <- w[(16 c + 8 (l >> 5) + j) * K + k][32 nb + (l & 31)]
  for (int e = tid; e < C * a.K * NBT * 512; e += TC_THREADS) {
    const int blk = e >> 9, l = (e >> 3) & 63, j = e & 7;
    const int nb = blk % NBT, k = (blk / NBT) % a.K, c = blk / (NBT * a.K);
    const int ch = 16 * c + 8 * (l >> 5) + j, col = 32 * nb + (l & 31);
    const float v = (ch < a.Fin && col < a.Fout) ? a.w[((int64_t)ch * a.K + k) * a.Fout + col] : 0.f;
    unsigned char* base = tc_smem + (size_t)blk * LEVB;
    if (PREC == DSPH_PREC_FP32) {
      reinterpret_cast<float*>(base)[j * 64 + l] = v;
    } else if (PREC == DSPH_PREC_BF16X3) {
      const __bf16 hi = (__bf16)v;
      reinterpret_cast<__bf16*>(base)[l * 8 + j] = hi;
      reinterpret_cast<__bf16*>(base + 1024)[l * 8 + j] = (__bf16)(v - (float)hi);
    } else {
      const unsigned au = __builtin_bit_cast(unsigned, v);
      const float r = v - __builtin_bit_cast(float, au & 0xffff0000u);
      const unsigned ru = __builtin_bit_cast(unsigned, r);
      reinterpret_cast<unsigned short*>(base)[l * 8 + j] = (unsigned short)(au >> 16);
      reinterpret_cast<unsigned short*>(base + 1024)[l * 8 + j] = (unsigned short)(ru >> 16);
      reinterpret_cast<__bf16*>(base + 2048)[l * 8 + j] = (__bf16)(r - __builtin_bit_cast(float, ru & 0xffff0000u));
    }
  }
  __syncthreads();
  const int64_t blocks_per_map = (a.rows + 31) / 32, total = blocks_per_map * a.N;
  const bool vec = (a.Fout & 3) == 0;
  for (int64_t b = (int64_t)blockIdx.x * (TC_THREADS / 64) + wave; b < total; b += (int64_t)gridDim.x * (TC_THREADS / 64)) {
    const int n = (int)(b / blocks_per_map);
    const int64_t m = (b - (int64_t)n * blocks_per_map) * 32 + px;
    const bool row_ok = m < a.rows;
    for (int nb0 = 0; nb0 < NBT; nb0 += NB) {
      sp_f32x16 acc[NB];
#pragma unroll
      for (int q = 0; q < NB; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
      for (int c = 0; c < C; ++c) {
        const int ch = 16 * c + 8 * g;
        for (int k = 0; k < a.K; ++k) {
          float row[8];
          const float* src = a.p[k] + ((int64_t)n * a.plane_rows + (row_ok ? m : 0)) * a.Fin + ch;
          if (VEC) {
            const sp_f32x4 v0 = (row_ok && ch < a.Fin) ? *reinterpret_cast<const sp_f32x4*>(src) : sp_f32x4{0.f, 0.f, 0.f, 0.f};
            const sp_f32x4 v1 = (row_ok && ch + 4 < a.Fin) ? *reinterpret_cast<const sp_f32x4*>(src + 4) : sp_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) { row[j] = v0[j]; row[4 + j] = v1[j]; }
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) row[j] = (row_ok && ch + j < a.Fin) ? src[j] : 0.f;
          }
#pragma unroll
          for (int q = 0; q < NB; ++q)
            if (nb0 + q < NBT)
              is_contract<8, PREC, false>(acc[q], row, tc_smem + (size_t)((c * a.K + k) * NBT + nb0 + q) * LEVB, lane);
        }
      }
#pragma unroll
      for (int q = 0; q < NB; ++q) {
        if (nb0 + q >= NBT || !row_ok) continue;
        float* dst = a.y + ((int64_t)n * a.rows + m) * a.Fout + 32 * (nb0 + q);
#pragma unroll
        for (int tq = 0; tq < 4; ++tq) {
          const int o = 8 * tq + 4 * g, oc = 32 * (nb0 + q) + o;
          float r[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) r[e] = apply_act(acc[q][4 * tq + e] + ((a.bias != nullptr && oc + e < a.Fout) ? a.bias[oc + e] : 0.f), a.act);
          if (vec && oc + 3 < a.Fout) *reinterpret_cast<sp_f32x4*>(dst + o) = sp_f32x4{r[0], r[1], r[2], r[3]};
          else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (oc + e < a.Fout) dst[o + e] = r[e];
          }
        }
      }
    }
  }
}

// true (and launched) when the shape fits this kernel; false: the caller falls back to cheb_contract.hip's kernels
bool launch_cheb_tcontract(const float* const* planes, int64_t plane_rows, const float* w, const float* bias, float* y, int64_t N,
                           int64_t rows, int32_t Fin, int32_t Fout, int32_t K, int32_t act, int32_t precision, int num_cu,
                           hipStream_t stream, int* rc) {
  *rc = DSPH_OK;
  if (K > TC_KMAX || N > (1 << 24)) return false;
  const int C = (Fin + 15) / 16, NBT = (Fout + 31) / 32;
  const size_t lds = (size_t)C * K * NBT * is_terms(precision) * is_term_bytes(precision);
  if (lds > (size_t)TC_LDS_MAX) return false;
  TContractArgs a;
  for (int k = 0; k < TC_KMAX; ++k) a.p[k] = k < K ? planes[k] : nullptr;
  bool vec_in = Fin % 4 == 0;
  for (int k = 0; k < K; ++k) vec_in = vec_in && (reinterpret_cast<uintptr_t>(planes[k]) & 15) == 0;
  if ((reinterpret_cast<uintptr_t>(y) & 15) && Fout % 4 == 0) return false;
  a.w = w; a.bias = bias; a.y = y;
  a.plane_rows = plane_rows; a.rows = rows;
  a.N = (int)N; a.Fin = Fin; a.Fout = Fout; a.K = K; a.act = act;
  const int64_t blocks = (rows + 31) / 32 * N;
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(2 * (int64_t)num_cu, (blocks + 7) / 8));
  const bool two = NBT >= 2;
#define DSPH_TC(P)                                                                                                             \
  do {                                                                                                                         \
    if (two && vec_in) hipLaunchKernelGGL((cheb_tcontract_kernel<P, 2, true>), dim3(grid), dim3(TC_THREADS), lds, stream, a);   \
    else if (two) hipLaunchKernelGGL((cheb_tcontract_kernel<P, 2, false>), dim3(grid), dim3(TC_THREADS), lds, stream, a);       \
    else if (vec_in) hipLaunchKernelGGL((cheb_tcontract_kernel<P, 1, true>), dim3(grid), dim3(TC_THREADS), lds, stream, a);     \
    else hipLaunchKernelGGL((cheb_tcontract_kernel<P, 1, false>), dim3(grid), dim3(TC_THREADS), lds, stream, a);                \
  } while (0)
  if (precision == DSPH_PREC_FP32) DSPH_TC(DSPH_PREC_FP32);
  else if (precision == DSPH_PREC_BF16X6) DSPH_TC(DSPH_PREC_BF16X6);
  else DSPH_TC(DSPH_PREC_BF16X3);
#undef DSPH_TC
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) *rc = hip_fail(e, "cheb_tcontract launch");
  return true;
}

}  // namespace dsph
