// Host side of the K = 8 quad-strip kernel (cheb_qstrip8_kernel.h): weight image and launch.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "cheb_qstrip8_kernel.h"

namespace dsph {

// Weight image: one 1 KiB A-operand fragment of v_mfma_f32_16x16x32_bf16 per (role, quarter oq, level of the role, hi | lo): lane
// l, element i <- s_j m_j w[(8 (l >> 4) + i) * K + j][16 oq + (l & 15)], j the level (top 7 - lev, middle 4 - lev, bottom 1 - lev),
// s_j the sign kept with the plane (qs_wsign), m_0 = 2 (level 0 runs doubled: the kernel halves y when it stores it).  The third
// level of `bottom` is a zero block.
__global__ __launch_bounds__(256) void qstrip8_wprep_kernel(const float* __restrict__ w, unsigned char* __restrict__ out, int ld, int f16) {
  constexpr int K = Q8_K;
  const int blk = blockIdx.x;  // (role * 2 + oq) * 3 + lev
  const int lev = blk % 3, oq = (blk / 3) & 1, role = blk / 6;
  const int j = (role == 0 ? 7 : role == 1 ? 4 : 1) - lev;
  const bool have = j >= 0;
  float sc = have ? qs_wsign(true, j) * (j == 0 ? 2.f : 1.f) : 0.f;
  if (f16) {  // f16 hi | lo (11 + 11 mantissa bits): the weights times the power of two that puts the largest in [2048, 4096)
    __shared__ float smax[256];
    float m = 0.f;
    for (int e = threadIdx.x; e < 32 * K * 32; e += 256) m = fmaxf(m, fabsf(w[(int64_t)(e / 32) * ld + e % 32]));
    smax[threadIdx.x] = m;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
      if ((int)threadIdx.x < st) smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + st]);
      __syncthreads();
    }
    int ex = 0;
    const float mx = smax[0];
    float pw = 1.f;
    if (mx > 0.f && mx < 3.0e38f) { (void)frexpf(mx, &ex); pw = ldexpf(1.f, 12 - ex); }
    sc *= pw;
    if (blk == 0 && threadIdx.x == 0) *reinterpret_cast<float*>(out + (size_t)Q8_WIMG) = 1.f / pw;
  }
  unsigned char* base = out + (size_t)blk * 2 * QS_FRAG;
  for (int e = threadIdx.x; e < 512; e += 256) {
    const int l = e >> 3, i = e & 7;
    const int ch = 8 * (l >> 4) + i, col = 16 * oq + (l & 15);
    const float v = have ? sc * w[((int64_t)ch * K + j) * ld + col] : 0.f;
    if (f16) {
      const _Float16 hi = (_Float16)v;
      const _Float16 lo = (_Float16)(v - (float)hi);
      reinterpret_cast<_Float16*>(base)[l * 8 + i] = hi;
      reinterpret_cast<_Float16*>(base + QS_FRAG)[l * 8 + i] = lo;
    } else {
      const __bf16 hi = (__bf16)v;
      const __bf16 lo = (__bf16)(v - (float)hi);
      reinterpret_cast<__bf16*>(base)[l * 8 + i] = hi;
      reinterpret_cast<__bf16*>(base + QS_FRAG)[l * 8 + i] = lo;
    }
  }
}

bool qstrip8_shape_ok(int32_t Fin, int32_t Fout, int32_t K) { return K == Q8_K && Fin == 32 && Fout == 32; }
size_t qstrip8_wimg_bytes() { return (size_t)Q8_WIMG + 256; }  // (+ the f16 image's factor)

// The tape of rows is cut as for the K = 5 kernel (qstrip_split: G workgroups, P pieces, w workgroups per piece each taking
// every w-th map) with this kernel's run-in; a single map (configs[3]) is one workgroup per piece.
int64_t qstrip8_split(int num_cu, int64_t tape_rows, int64_t N, int64_t mean_height, int* grid, int* pieces, int* wg_per_piece) {
  const int g = (int)std::max<int64_t>(8, std::min<int64_t>(num_cu / 8 * 8, tape_rows * N / 64 / 8 * 8));
  auto span_of = [&](int64_t w) {
    const int64_t P = std::max<int64_t>(1, g / w), share = (tape_rows + P - 1) / P, maps = (N + w - 1) / w;
    const int64_t runs = share / std::max<int64_t>(1, mean_height) + 2;
    return (share + runs * (Q8_RUNIN + 1)) * maps;
  };
  int64_t best_w = 1, best = -1;
  if (N <= g) { best_w = N; best = span_of(N); }
  else
    for (int64_t w = 1; w <= g; w *= 2) {
      const int64_t sp = span_of(w);
      if (best < 0 || sp < best) { best = sp; best_w = w; }
    }
  if (grid) *grid = g;
  if (pieces) *pieces = (int)std::max<int64_t>(1, g / best_w);
  if (wg_per_piece) *wg_per_piece = (int)best_w;
  return best;
}

int launch_cheb_qstrip8(const QStrip8Launch& s, hipStream_t stream) {
  if (s.prep_weights) {
    hipLaunchKernelGGL(qstrip8_wprep_kernel, dim3(3 * 2 * 3), dim3(256), 0, stream, s.w, s.wimg, (int)s.ld_w, s.f16 ? 1 : 0);
    DSPH_HIP(hipGetLastError());
  }
  Q8Args a;
  a.x = s.x;
  a.bias = s.bias;
  a.y = s.y;
  a.wimg = s.wimg;
  a.gvals8 = s.gvals8;
  a.gdiag = s.gdiag;
  a.strips = s.strips;
  a.tab = s.tab;
  a.prefix = s.prefix;
  a.x_rows = s.x_rows;
  a.y_rows = s.y_rows;
  a.nstrips = s.nstrips;
  a.N = (int)s.N;
  a.ld = s.ld;
  a.act = s.act;
  a.xsc = s.f16 ? ldexpf(1.f, s.f16_xexp) : 1.f;
  a.xsc_inv = s.f16 ? ldexpf(1.f, -s.f16_xexp) : 1.f;
  int grid;
  (void)qstrip8_split(s.num_cu, s.tape_rows, s.N, s.tape_rows / std::max(1, s.nstrips), &grid, &a.pieces, &a.wg_per_piece);
#ifdef DSPH_Q8_SIX_WAVES  // (tuning: the six-wave variant, `top` fetching)
  if (s.f16) hipLaunchKernelGGL((cheb_qstrip8_kernel<0, true>), dim3(grid), dim3(Q8_THREADS), 0, stream, a);
  else hipLaunchKernelGGL((cheb_qstrip8_kernel<0, false>), dim3(grid), dim3(Q8_THREADS), 0, stream, a);
#else
  if (s.f16) hipLaunchKernelGGL((cheb_qstrip8_kernel<1, true>), dim3(grid), dim3(512), 0, stream, a);
  else hipLaunchKernelGGL((cheb_qstrip8_kernel<1, false>), dim3(grid), dim3(512), 0, stream, a);
#endif
  DSPH_HIP(hipGetLastError());
  return DSPH_OK;
}

}  // namespace dsph
