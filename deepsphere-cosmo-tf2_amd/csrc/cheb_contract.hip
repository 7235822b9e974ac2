// Dense feature contraction over K Chebyshev planes, without ever building the reference's
// (N*M) x (Fin*K) matrix (tf.stack / reshape / transpose / tf.matmul, gnn_layers.py:144-150):
//     y[n,m,o] = act( sum_k sum_f planes[k][n,m,f] * w[f*K + k, o] + bias[o] )
// Plane k is read in the caller's (N, rows, Fin) layout; the weight row index f*K + k is the
// reference's (channel-major, order-minor) and is resolved when the weight tile is staged.
//
// gfx950 mapping: one 256-thread workgroup (4 waves, one per SIMD) owns 128 pixels x (32*NB)
// output channels; each wave a 32-pixel slab, accumulated in NB 32x32 fp32 MFMA tiles
// (v_mfma_f32_32x32x2_f32: exact fp32, bitwise an fma chain in the order the products are fed).
// Per (k, 32-channel chunk): the 128 x 32 plane tile and the 32 x (32*NB) weight tile are staged
// in LDS with 16-byte coalesced loads; A fragments are read with ds_read_b128 from rows padded
// by 16 B (conflict-free for the four 16-lane groups), B fragments with ds_read_b32.
// Roofline: reads K planes + writes y once -> HBM-bound for Fin*Fout small, fp32-MFMA-bound
// (157 TFLOP/s) from about Fin = Fout = 64 up.
#include <algorithm>

#include "dsphere_common.h"

namespace dsph {

constexpr int TM = 128;      // pixels per workgroup
constexpr int FC = 32;       // input channels staged per step
constexpr int A_LD = FC + 4; // padded LDS row (floats)
constexpr int KMAX = 64;

struct PlanePtrs {
  const float* p[KMAX];
};

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NB, bool VEC_A, bool VEC_W>
__global__ __launch_bounds__(256) void cheb_contract_f32_kernel(
    PlanePtrs planes, int64_t plane_rows, const float* __restrict__ w,
    const float* __restrict__ bias, float* __restrict__ y, int64_t rows, int Fin, int Fout, int K,
    int act) {
  constexpr int WN = 32 * NB;
  __shared__ __attribute__((aligned(16))) float lds[TM * A_LD + FC * WN];
  float* __restrict__ sA = lds;
  float* __restrict__ sW = lds + TM * A_LD;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 31;
  const int h = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * TM;
  const int n = blockIdx.y;
  const int ob = blockIdx.z * WN;

  f32x16 acc[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

  for (int k = 0; k < K; ++k) {
    const float* __restrict__ pl = planes.p[k] + (int64_t)n * plane_rows * Fin;
    for (int c0 = 0; c0 < Fin; c0 += FC) {
      __syncthreads();
      // ---- stage the plane tile: TM rows x FC channels -------------------------------------
      if (VEC_A) {
#pragma unroll
        for (int i = 0; i < (TM * FC / 4) / 256; ++i) {
          const int idx = i * 256 + tid;
          const int row = idx / (FC / 4);
          const int q = idx % (FC / 4);
          const int64_t m = m0 + row;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (m < rows && c0 + 4 * q < Fin)
            v = *reinterpret_cast<const float4*>(pl + m * Fin + c0 + 4 * q);
          *reinterpret_cast<float4*>(sA + row * A_LD + 4 * q) = v;
        }
      } else {
#pragma unroll 4
        for (int i = 0; i < (TM * FC) / 256; ++i) {
          const int idx = i * 256 + tid;
          const int row = idx / FC;
          const int f = idx % FC;
          const int64_t m = m0 + row;
          float v = 0.f;
          if (m < rows && c0 + f < Fin) v = pl[m * Fin + c0 + f];
          sA[row * A_LD + f] = v;
        }
      }
      // ---- stage the weight tile: rows (c0+f)*K + k, columns ob .. ob+WN ---------------------
      if (VEC_W) {
#pragma unroll
        for (int i = 0; i < (FC * WN / 4 + 255) / 256; ++i) {
          const int idx = i * 256 + tid;
          if (idx < FC * WN / 4) {
            const int f = idx / (WN / 4);
            const int q = idx % (WN / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c0 + f < Fin && ob + 4 * q < Fout)
              v = *reinterpret_cast<const float4*>(w + ((int64_t)(c0 + f) * K + k) * Fout + ob + 4 * q);
            *reinterpret_cast<float4*>(sW + f * WN + 4 * q) = v;
          }
        }
      } else {
#pragma unroll 4
        for (int i = 0; i < (FC * WN) / 256; ++i) {
          const int idx = i * 256 + tid;
          const int f = idx / WN;
          const int o = idx % WN;
          float v = 0.f;
          if (c0 + f < Fin && ob + o < Fout) v = w[((int64_t)(c0 + f) * K + k) * Fout + ob + o];
          sW[f * WN + o] = v;
        }
      }
      __syncthreads();
      // ---- MFMA: lane (li, h) feeds inner index 8t + 4h + s at step (t, s) -------------------
      const float* __restrict__ arow = sA + (wave * 32 + li) * A_LD + 4 * h;
#pragma unroll
      for (int t = 0; t < FC / 8; ++t) {
        const float4 a4 = *reinterpret_cast<const float4*>(arow + 8 * t);
        const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const float* __restrict__ wrow = sW + (8 * t + 4 * h + s) * WN + li;
#pragma unroll
          for (int b = 0; b < NB; ++b)
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], wrow[32 * b], acc[b], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue: bias, activation, store.  C/D map: row = (r&3) + 8*(r>>2) + 4*h, col = li ----
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int o = ob + 32 * b + li;
    if (o >= Fout) continue;
    const float bv = bias ? bias[o] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
      const int64_t m = m0 + wave * 32 + row;
      if (m < rows) y[((int64_t)n * rows + m) * Fout + o] = apply_act(acc[b][r] + bv, act);
    }
  }
}

// Few input channels (Fin*K <= 64: the first layer of every network has Fin = 1): the MFMA kernel above pads the
// inner dimension to 32 per order, i.e. does 32x the work at Fin = 1 and was MFMA-bound at 10 % of the
// roofline.  Here one thread makes four adjacent outputs of one pixel with plain FMAs (order k outer, f inner,
// the same chain as the MFMA kernel), the weights sit in LDS, the Fin*K plane values of a pixel are read by the
// threads of its row as broadcasts: HBM-bound on the y write.
__global__ __launch_bounds__(256) void cheb_contract_small_kernel(PlanePtrs planes, int64_t plane_rows,
                                                                  const float* __restrict__ w,
                                                                  const float* __restrict__ bias, float* __restrict__ y,
                                                                  int64_t rows, int Fin, int Fout, int K, int act,
                                                                  int qshift, int iters, int vec_ok) {
  // LDS: weights [Fin*K][4*Qp] (zero-padded columns), then this workgroup's plane values [K][rows_blk][Fin]:
  // read from global memory as K contiguous runs (a load per pixel and order would be vector-memory-issue bound)
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Qp = 1 << qshift, FK = Fin * K, ldw = 4 * Qp, rpb = 256 >> qshift, rows_blk = iters * rpb;
  float* __restrict__ sw = sm;
  float* __restrict__ sp = sm + FK * ldw;
  for (int i = threadIdx.x; i < FK * ldw; i += 256) {
    const int r = i >> (qshift + 2), o = i & (ldw - 1);
    sw[i] = o < Fout ? w[(int64_t)r * Fout + o] : 0.f;
  }
  const int n = blockIdx.y;
  const int64_t mb = (int64_t)blockIdx.x * rows_blk;
  const int64_t nrow = rows - mb < rows_blk ? rows - mb : rows_blk;
  const int run = (int)nrow * Fin;  // floats of one plane that belong to this workgroup
  for (int k = 0; k < K; ++k) {
    const float* __restrict__ src = planes.p[k] + ((int64_t)n * plane_rows + mb) * Fin;
    for (int i = threadIdx.x; i < run; i += 256) sp[k * rows_blk * Fin + i] = src[i];
  }
  __syncthreads();
  const int oq = threadIdx.x & (Qp - 1), rsub = threadIdx.x >> qshift;
  const int o0 = 4 * oq;
  if (o0 >= Fout) return;
  float bv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bv[j] = (bias && o0 + j < Fout) ? bias[o0 + j] : 0.f;
  for (int it = 0; it < iters; ++it) {
    const int rl = it * rpb + rsub;
    if (rl >= nrow) break;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < K; ++k) {
      const float* __restrict__ pl = sp + (k * rows_blk + rl) * Fin;
      for (int f = 0; f < Fin; ++f) {
        const float v = pl[f];
        const float4 wv = *reinterpret_cast<const float4*>(sw + (f * K + k) * ldw + o0);
        acc.x = fmaf(v, wv.x, acc.x);
        acc.y = fmaf(v, wv.y, acc.y);
        acc.z = fmaf(v, wv.z, acc.z);
        acc.w = fmaf(v, wv.w, acc.w);
      }
    }
    float r[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = apply_act(r[j] + bv[j], act);
    float* __restrict__ yp = y + ((int64_t)n * rows + mb + rl) * Fout + o0;
    if (vec_ok && o0 + 3 < Fout) {
      *reinterpret_cast<float4*>(yp) = make_float4(r[0], r[1], r[2], r[3]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (o0 + j < Fout) yp[j] = r[j];
    }
  }
}

int launch_cheb_contract(const float* const* planes, int64_t plane_rows, const float* w,
                         const float* bias, float* y, int64_t N, int64_t rows, int32_t Fin,
                         int32_t Fout, int32_t K, int32_t act, int32_t precision,
                         hipStream_t stream) {
  if (rows <= 0 || N <= 0) return DSPH_OK;
  // plane rows straight into the MFMA operand layout, in the arithmetic asked for (cheb_tcontract.hip: four input channels and
  // more); the kernels below -- always exact fp32 -- take what that one does not (fewer channels: the plain-FMA kernel)
  if (Fin * K > 64 || Fin >= 4) {
    int rc = DSPH_OK, ncu = 256, dev = 0;
    static int ncu_of[64] = {0};  // (looked up once per device: hipGetDeviceProperties is slow; a racing second writer stores the same value)
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
      if (ncu_of[dev] == 0) {
        hipDeviceProp_t prop;
        ncu_of[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
      }
      ncu = ncu_of[dev];
    }
    if (launch_cheb_tcontract(planes, plane_rows, w, bias, y, N, rows, Fin, Fout, K, act, precision, ncu, stream, &rc)) return rc;
  }
  if (K > KMAX) {
    set_error("cheb_contract: K = %d exceeds %d", K, KMAX);
    return DSPH_E_UNSUPPORTED;
  }
  if (N > 65535) {
    set_error("cheb_contract: batch %lld exceeds 65535", (long long)N);
    return DSPH_E_UNSUPPORTED;
  }
  PlanePtrs pp;
  bool vec_a = (Fin % 4 == 0);
  for (int k = 0; k < K; ++k) {
    pp.p[k] = planes[k];
    vec_a = vec_a && ((reinterpret_cast<uintptr_t>(planes[k]) & 15) == 0);
  }
  for (int k = K; k < KMAX; ++k) pp.p[k] = nullptr;
  const bool vec_w = (Fout % 4 == 0) && ((reinterpret_cast<uintptr_t>(w) & 15) == 0);
  {
    const int Q = (Fout + 3) / 4;
    int qshift = 0;
    while ((1 << qshift) < Q) ++qshift;
    const size_t sw_bytes = (size_t)Fin * K * 4 * ((size_t)1 << qshift) * sizeof(float);
    if (Fin * K <= 64 && qshift <= 6 && sw_bytes <= 24 * 1024) {
      const int rpb = 256 >> qshift;
      int iters = (int)((40 * 1024) / ((size_t)rpb * Fin * K * sizeof(float)));  // plane values of the workgroup: <= 40 KiB
      iters = std::max(1, std::min(16, iters));
      const size_t lds = sw_bytes + (size_t)iters * rpb * Fin * K * sizeof(float);
      const int rows_per_blk = iters * rpb;
      const int64_t nblk = (rows + rows_per_blk - 1) / rows_per_blk;
      const int vec_ok = (Fout % 4 == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
      hipLaunchKernelGGL(cheb_contract_small_kernel, dim3((unsigned)nblk, (unsigned)N), dim3(256), lds, stream, pp,
                         plane_rows, w, bias, y, rows, (int)Fin, (int)Fout, (int)K, (int)act, qshift, iters, vec_ok);
      DSPH_HIP(hipGetLastError());
      return DSPH_OK;
    }
  }
  const int nb = Fout > 32 ? 2 : 1;
  const int wn = 32 * nb;
  dim3 grid((unsigned)((rows + TM - 1) / TM), (unsigned)N, (unsigned)((Fout + wn - 1) / wn));
#define DSPH_LAUNCH_CT(NB, VA, VW)                                                               \
  hipLaunchKernelGGL((cheb_contract_f32_kernel<NB, VA, VW>), grid, dim3(256), 0, stream, pp,      \
                     plane_rows, w, bias, y, rows, (int)Fin, (int)Fout, (int)K, (int)act)
  if (nb == 2) {
    if (vec_a && vec_w) DSPH_LAUNCH_CT(2, true, true);
    else if (vec_a) DSPH_LAUNCH_CT(2, true, false);
    else if (vec_w) DSPH_LAUNCH_CT(2, false, true);
    else DSPH_LAUNCH_CT(2, false, false);
  } else {
    if (vec_a && vec_w) DSPH_LAUNCH_CT(1, true, true);
    else if (vec_a) DSPH_LAUNCH_CT(1, true, false);
    else if (vec_w) DSPH_LAUNCH_CT(1, false, true);
    else DSPH_LAUNCH_CT(1, false, false);
  }
#undef DSPH_LAUNCH_CT
  DSPH_HIP(hipGetLastError());
  return DSPH_OK;
}

}  // namespace dsph
