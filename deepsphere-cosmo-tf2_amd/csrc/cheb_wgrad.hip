// Weight gradient of the Chebyshev / monomial convolution:
//     dw[f*K + k, o] = sum_{n,m} planes[k][n,m,f] * dy[n,m,o]
// The reference gets it from TensorFlow's autodiff of tf.matmul (gnn_layers.py:149); a library GEMM
// is the wrong tool for this shape (a Fin x Fout result reduced over N*M = 5e7 rows: one tile, no
// split-K), so it is a hand-written split-over-pixels MFMA kernel.
//
// gfx950 mapping: a 256-thread workgroup owns a run of pixel rows of one map; wave w owns one
// (32 input channels) x (32 output channels) block for all K orders: K accumulator tiles of
// v_mfma_f32_32x32x2_f32 (exact fp32).  Both operands are read straight from HBM in MFMA operand
// order -- lane (i, kk) of A = planes[k][row 2t+kk][32*fb + i], of B = dy[row 2t+kk][32*ob + i]: each
// half-wave reads one contiguous 128-byte run, dy once per row pair for all K planes.  Every workgroup
// writes its partial K x Fin x Fout slab; a second kernel adds the slabs in a fixed order, so the
// result is bitwise reproducible (float atomics would not be).
// Roofline: K planes + dy read once -> HBM-bound for small Fin*Fout; fp32-MFMA-bound from 64 x 64 up.
#include <algorithm>

#include "dsphere_common.h"

namespace dsph {

constexpr int WG_KC = 8;        // most orders per pass (accumulator tiles per wave)
constexpr int WG_ROWS = 4096;   // least pixel rows per work item
constexpr int WG_ITEMS = 768;   // most work items (= partial slabs) per call: 3 per CU of an MI355X, 60 MB of slabs at K 5, 64 -> 64.
                                // A constant, not the device's CU count: the partition, and with it the rounding, is the same everywhere.
constexpr int WG_KMAX = 64;

struct WgradPlanes {
  const float* p[WG_KMAX];
};

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KC>
__global__ __launch_bounds__(256, 2) void cheb_wgrad_kernel(WgradPlanes planes, int64_t plane_rows,
                                                         const float* __restrict__ dy, float* __restrict__ slabs,
                                                         int64_t rows, int N, int Fin, int Fout, int K, int k0,
                                                         int kc, int nfb, int nob, int chunks_per_map, int rows_per_item) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int combo = blockIdx.y * 4 + wave;  // (fb, ob) block of this wave
  const bool active = combo < nfb * nob;
  const int fb = active ? combo / nob : 0, ob = active ? combo % nob : 0;
  const int i = lane & 31, kk = lane >> 5;
  const int fcol = 32 * fb + i, ocol = 32 * ob + i;
  const bool f_ok = active && fcol < Fin, o_ok = active && ocol < Fout;
  const int work = blockIdx.x;  // (map n, chunk) pairs
  const int n = work / chunks_per_map, chunk = work % chunks_per_map;
  const int64_t m0 = (int64_t)chunk * rows_per_item;
  const int64_t m1 = m0 + rows_per_item < rows ? m0 + rows_per_item : rows;

  f32x16 acc[KC];
#pragma unroll
  for (int k = 0; k < KC; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

  const float* __restrict__ dyp = dy + ((int64_t)n * rows) * Fout + (o_ok ? ocol : 0);
  const int64_t poff = (int64_t)n * plane_rows * Fin + (f_ok ? fcol : 0);
  // lanes kk = 1 run one row ahead of lanes kk = 0.  UN row pairs per trip: all their loads are issued
  // before the first MFMA needs one (clamped addresses, zeros selected afterwards -- no branches)
  constexpr int UN = KC <= 4 ? 8 : 4;
  for (int64_t mb = m0; mb < m1; mb += 2 * UN) {
    float b[UN], av[UN][KC];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t m = mb + 2 * u + kk;
      const int64_t mc = m < m1 ? m : m1 - 1;
      b[u] = dyp[mc * Fout];
#pragma unroll
      for (int k = 0; k < KC; ++k)
        if (k < kc) av[u][k] = planes.p[k0 + k][poff + mc * Fin];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const bool row_ok = mb + 2 * u + kk < m1;
      const float bu = (row_ok && o_ok) ? b[u] : 0.f;
#pragma unroll
      for (int k = 0; k < KC; ++k)
        if (k < kc) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32((row_ok && f_ok) ? av[u][k] : 0.f, bu, acc[k], 0, 0, 0);
    }
  }
  if (!active) return;
  // slab layout [work][k][f][o] (f, o padded to the 32-blocks actually stored)
  float* __restrict__ slab = slabs + (int64_t)work * K * Fin * Fout;
  const int h = lane >> 5;
#pragma unroll
  for (int k = 0; k < KC; ++k) {
    if (k < kc) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int f = 32 * fb + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (f < Fin && ocol < Fout) slab[((int64_t)(k0 + k) * Fin + f) * Fout + ocol] = acc[k][r];
      }
    }
  }
}

// dw[f*K + k, o] = sum over work items (ascending) of slab[work][k][f][o]
__global__ __launch_bounds__(256) void cheb_wgrad_reduce_kernel(const float* __restrict__ slabs,
                                                                float* __restrict__ dw, int nwork, int Fin,
                                                                int Fout, int K) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int total = K * Fin * Fout;
  if (e >= total) return;
  const int o = e % Fout, f = (e / Fout) % Fin, k = e / (Fout * Fin);
  float s = 0.f;
  for (int w = 0; w < nwork; ++w) s += slabs[(int64_t)w * total + e];
  dw[((int64_t)f * K + k) * Fout + o] = s;
}

// Rows per work item: 4096, or as many more as keep the number of items (N x chunks per map) at WG_ITEMS; a multiple of 16
// (the kernel's trip).  At the headline shape: 4 maps x 192 chunks of 65,536 rows = 768 slabs of 80 KiB (12,288 of them before).
static int64_t wgrad_rows_per_item(int64_t N, int64_t rows) {
  const int64_t chunks_max = std::max<int64_t>(1, WG_ITEMS / std::max<int64_t>(N, 1));
  int64_t per = std::max<int64_t>(WG_ROWS, (rows + chunks_max - 1) / chunks_max);
  return (per + 15) / 16 * 16;
}
static int wgrad_chunks(int64_t N, int64_t rows) {
  const int64_t per = wgrad_rows_per_item(N, rows);
  return (int)((rows + per - 1) / per);
}
static int wgrad_work_items(int64_t N, int64_t rows) { return (int)(N * wgrad_chunks(N, rows)); }

size_t wgrad_workspace_bytes(int64_t N, int64_t rows, int32_t Fin, int32_t Fout, int32_t K) {
  return (size_t)wgrad_work_items(N, rows) * K * Fin * Fout * sizeof(float);
}

int launch_cheb_wgrad(const float* const* planes, int64_t plane_rows, const float* dy, float* dw,
                      int64_t N, int64_t rows, int32_t Fin, int32_t Fout, int32_t K, void* workspace,
                      size_t workspace_bytes, hipStream_t stream) {
  if (K > WG_KMAX) {
    set_error("cheb_wgrad: K = %d exceeds %d", K, WG_KMAX);
    return DSPH_E_UNSUPPORTED;
  }
  const size_t need = wgrad_workspace_bytes(N, rows, Fin, Fout, K);
  if (!workspace || workspace_bytes < need) {
    set_error("cheb_wgrad: workspace %zu bytes, need %zu", workspace_bytes, need);
    return DSPH_E_WORKSPACE;
  }
  WgradPlanes pp;
  for (int k = 0; k < WG_KMAX; ++k) pp.p[k] = k < K ? planes[k] : nullptr;
  const int chunks = wgrad_chunks(N, rows);
  const int rows_per_item = (int)wgrad_rows_per_item(N, rows);
  const int nwork = wgrad_work_items(N, rows);
  const int nfb = (Fin + 31) / 32, nob = (Fout + 31) / 32;
  dim3 grid((unsigned)nwork, (unsigned)((nfb * nob + 3) / 4));
  for (int k0 = 0; k0 < K; k0 += WG_KC) {
    const int kc = K - k0 < WG_KC ? K - k0 : WG_KC;
#define DSPH_WGRAD(KC)                                                                                   \
  hipLaunchKernelGGL(cheb_wgrad_kernel<KC>, grid, dim3(256), 0, stream, pp, plane_rows, dy,              \
                     static_cast<float*>(workspace), rows, (int)N, (int)Fin, (int)Fout, (int)K, k0, kc, nfb, nob, \
                     chunks, rows_per_item)
    if (kc <= 2) DSPH_WGRAD(2);
    else if (kc <= 4) DSPH_WGRAD(4);
    else if (kc <= 5) DSPH_WGRAD(5);
    else DSPH_WGRAD(8);
#undef DSPH_WGRAD
    DSPH_HIP(hipGetLastError());
  }
  const int total = K * Fin * Fout;
  hipLaunchKernelGGL(cheb_wgrad_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, stream,
                     static_cast<const float*>(workspace), dw, nwork, (int)Fin, (int)Fout, (int)K);
  DSPH_HIP(hipGetLastError());
  return DSPH_OK;
}

}  // namespace dsph
