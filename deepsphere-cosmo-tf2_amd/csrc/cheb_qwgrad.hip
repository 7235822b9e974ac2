// Host side of the quad-strip weight gradient (cheb_qwgrad_kernel.h): launch and the fixed-order sum of the slabs.
#include <algorithm>

#include "cheb_qwgrad_kernel.h"

namespace dsph {

// dw[(f K + k) ld + o] (+)= the rule of cheb_qwgrad_kernel.h applied to the sums of the five products over all slabs.  Sixteen
// lanes per (f, o): lane q adds the slabs q, q + 16, ... in ascending order, the sixteen partial sums are added in a fixed tree
// (the same scheme as fused_wgrad_reduce_kernel): the result does not depend on the launch's timing.
__global__ __launch_bounds__(256) void qwgrad_reduce_kernel(const float* __restrict__ slabs, int nslabs, float* __restrict__ dw, int ld,
                                                            int cheb, int accumulate) {
  __shared__ float part[QW_PAIRS][16][17];
  const int el = threadIdx.x & 15, q = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + el;  // f * 64 + o
  float s[QW_PAIRS] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (int i = q; i < nslabs; i += 16) {
    // (workgroup i of the kernel: its place `ord` in the XCD-wise order; the odd ones ran on -dy)
    const float sg = (((i & 7) * (nslabs >> 3) + (i >> 3)) & 1) ? -1.f : 1.f;
#pragma unroll
    for (int pr = 0; pr < QW_PAIRS; ++pr) s[pr] = fmaf(sg, slabs[(size_t)i * QW_SLAB + (size_t)pr * 4096 + e], s[pr]);
  }
#pragma unroll
  for (int pr = 0; pr < QW_PAIRS; ++pr) part[pr][q][el] = s[pr];
  __syncthreads();
  if (q == 0) {
    float g[QW_PAIRS];
#pragma unroll
    for (int pr = 0; pr < QW_PAIRS; ++pr) {
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = part[pr][i][el];
#pragma unroll
      for (int h = 8; h >= 1; h >>= 1)
#pragma unroll
        for (int i = 0; i < h; ++i) v[i] += v[i + h];
      g[pr] = v[0];
    }
    // g: G00 G01 G20 G21 G22
    const float d[5] = {g[0], g[1], g[2], cheb ? 2.f * g[3] - g[1] : g[3], cheb ? 2.f * g[4] - g[0] : g[4]};
    const int f = e >> 6, o = e & 63;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      float* out = dw + (size_t)(f * 5 + k) * ld + o;
      *out = accumulate ? *out + d[k] : d[k];
    }
  }
}

bool qwgrad_shape_ok(int32_t Fin, int32_t Fout, int32_t K) { return K == 5 && Fin == 64 && Fout == 64; }

static int qwgrad_grid(const QWgradLaunch& s, int* pieces, int* wg_per_piece) {
  int grid;
  (void)qstrip_split(s.num_cu, s.tape_rows, s.N, s.tape_rows / std::max(1, s.nstrips), &grid, pieces, wg_per_piece);
  return grid;
}

size_t qwgrad_slab_bytes(int num_cu) { return (size_t)std::max(8, num_cu / 8 * 8) * QW_SLAB * sizeof(float); }

int launch_cheb_qwgrad(const QWgradLaunch& s, hipStream_t stream) {
  QWgradArgs a;
  a.x = s.x;
  a.dy = s.dy;
  a.slabs = s.slabs;
  a.gvals8 = s.gvals8;
  a.gdiag = s.gdiag;
  a.strips = s.strips;
  a.tab = s.tab;
  a.prefix = s.prefix;
  a.x_rows = s.x_rows;
  a.dy_rows = s.dy_rows;
  a.nstrips = s.nstrips;
  a.N = (int)s.N;
  a.lddy = s.lddy;
  const int grid = qwgrad_grid(s, &a.pieces, &a.wg_per_piece);
  if (s.cheb) hipLaunchKernelGGL(cheb_qwgrad5_kernel<true>, dim3(grid), dim3(QS_THREADS), 0, stream, a);
  else hipLaunchKernelGGL(cheb_qwgrad5_kernel<false>, dim3(grid), dim3(QS_THREADS), 0, stream, a);
  DSPH_HIP(hipGetLastError());
  hipLaunchKernelGGL(qwgrad_reduce_kernel, dim3(4096 / 16), dim3(256), 0, stream, s.slabs, grid, s.dw, (int)s.lddw, s.cheb ? 1 : 0,
                     s.accumulate ? 1 : 0);
  DSPH_HIP(hipGetLastError());
  return DSPH_OK;
}

}  // namespace dsph
