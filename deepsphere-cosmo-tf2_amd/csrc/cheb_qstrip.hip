// Host side of the quad-strip kernel (cheb_qstrip_kernel.h): weight image and launch.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "cheb_qstrip_kernel.h"

namespace dsph {

// Weight image: one 1 KiB A-operand fragment of v_mfma_f32_16x16x32_{bf16,f16} per (role, quarter oq, level of the role,
// 32-channel block kb, hi | lo): lane l, element i <- s_j m_j w[(32 kb + 8 (l >> 4) + i) * K + j][16 oq + (l & 15)], j the level
// (H: 4 - lev, L: 1 - lev), s_j the sign kept with the plane (qs_wsign), m_0 = 2 in the Chebyshev basis (level 0 runs doubled:
// the kernel halves y when it stores it).  The third level of role L is a zero block.
// f16 (DSPH_PREC_F16X3): hi | lo are f16 (11 + 11 mantissa bits), the weights times the power of two that puts the largest
// one in [2048, 4096) -- lo parts stay normal numbers --; the inverse factor sits behind the fragments for the kernel's store.
__global__ __launch_bounds__(256) void qstrip_wprep_kernel(const float* __restrict__ w, unsigned char* __restrict__ out, int Fin,
                                                           int Fout, int cheb, int ld, int f16) {
  constexpr int K = 5;
  const int blk = blockIdx.x;  // ((role * 4 + oq) * 3 + lev) * 2 + kb
  const int kb = blk & 1, lev = (blk >> 1) % 3, oq = ((blk >> 1) / 3) & 3, role = (blk >> 1) / 12;
  const int j = role == 0 ? K - 1 - lev : 1 - lev;
  const bool have = role == 0 || lev < 2;
  float sc = qs_wsign(cheb != 0, j) * ((cheb != 0 && j == 0) ? 2.f : 1.f);
  if (f16) {  // (every block finds the same maximum: 20,480 values)
    __shared__ float smax[256];
    float m = 0.f;
    for (int e = threadIdx.x; e < Fin * K * Fout; e += 256) m = fmaxf(m, fabsf(w[(int64_t)(e / Fout) * ld + e % Fout]));
    smax[threadIdx.x] = m;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
      if ((int)threadIdx.x < st) smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + st]);
      __syncthreads();
    }
    int ex = 0;
    const float mx = smax[0];
    float pw = 1.f;
    if (mx > 0.f && mx < 3.0e38f) { (void)frexpf(mx, &ex); pw = ldexpf(1.f, 12 - ex); }  // mx = f 2^ex, f in [0.5, 1): mx pw in [2048, 4096)
    sc *= pw;
    if (blk == 0 && threadIdx.x == 0) *reinterpret_cast<float*>(out + (size_t)2 * 4 * 3 * 2 * 2 * QS_FRAG) = 1.f / pw;
  }
  unsigned char* base = out + (size_t)blk * 2 * QS_FRAG;
  for (int e = threadIdx.x; e < 512; e += 256) {
    const int l = e >> 3, i = e & 7;
    const int ch = 32 * kb + 8 * (l >> 4) + i, col = 16 * oq + (l & 15);
    const float v = (have && ch < Fin && col < Fout) ? sc * w[((int64_t)ch * K + j) * ld + col] : 0.f;
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

bool qstrip_shape_ok(int32_t Fin, int32_t Fout, int32_t K) { return K == 5 && Fin == 64 && Fout == 64; }

// How the kernel's workgroups share the work (cheb_qstrip_kernel.h): G workgroups (a multiple of 8: the kernel deals XCD by
// XCD; one per CU, fewer when a workgroup would get under 64 rows), the per-map tape of `tape_rows` rows cut into P pieces, w
// workgroups per piece taking every w-th map.  w = N (one map each, in step on the same rows of L~) when the batch fits;
// otherwise the w in {1, 2, 4, ...} with the shortest busiest workgroup.  Returns that workgroup's steps: its rows plus nine
// run-in steps per run of rows, every run rounded up to whole triples of steps (the step body is unrolled three times).
int64_t qstrip_split(int num_cu, int64_t tape_rows, int64_t N, int64_t mean_height, int* grid, int* pieces, int* wg_per_piece) {
  const int g = (int)std::max<int64_t>(8, std::min<int64_t>(num_cu / 8 * 8, tape_rows * N / 64 / 8 * 8));
  auto span_of = [&](int64_t w) {
    const int64_t P = std::max<int64_t>(1, g / w), share = (tape_rows + P - 1) / P, maps = (N + w - 1) / w;
    const int64_t runs = share / std::max<int64_t>(1, mean_height) + 2;
    return (share + runs * (2 * QS_D + 1 + 1)) * maps;
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

size_t qstrip_wimg_bytes() { return (size_t)2 * 4 * 3 * 2 * 2 * QS_FRAG + 256; }  // 96 KiB + the f16 image's factor

int launch_cheb_qstrip(const QStripLaunch& s, hipStream_t stream) {
  if (s.prep_weights) {
    hipLaunchKernelGGL(qstrip_wprep_kernel, dim3(2 * 4 * 3 * 2), dim3(256), 0, stream, s.w, s.wimg, (int)s.Fin, (int)s.Fout, s.cheb ? 1 : 0,
                       (int)s.ld, s.f16 ? 1 : 0);
    DSPH_HIP(hipGetLastError());
  }
  QStripArgs a;
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
  a.Fin = s.Fin;
  a.Fout = s.Fout;
  a.ld = s.ld;
  a.act = s.act;
  a.xsc = s.f16 ? ldexpf(1.f, s.f16_xexp) : 1.f;
  a.xsc_inv = s.f16 ? ldexpf(1.f, -s.f16_xexp) : 1.f;
  int grid;
  (void)qstrip_split(s.num_cu, s.tape_rows, s.N, s.tape_rows / std::max(1, s.nstrips), &grid, &a.pieces, &a.wg_per_piece);
  void (*kern)(QStripArgs) = s.f16 ? (s.cheb ? cheb_qstrip5_kernel<true, true> : cheb_qstrip5_kernel<false, true>)
                                   : (s.cheb ? cheb_qstrip5_kernel<true, false> : cheb_qstrip5_kernel<false, false>);
#ifdef DSPH_QS_STAMPS
  static unsigned* d_stamps = nullptr;
  constexpr size_t NST = 8 * 4 * 10;
  if (!d_stamps) DSPH_HIP(hipMalloc(&d_stamps, NST * 4));
  DSPH_HIP(hipMemsetAsync(d_stamps, 0, NST * 4, stream));
  a.stamps = d_stamps;
#endif
  hipLaunchKernelGGL(kern, dim3(grid), dim3(QS_THREADS), 0, stream, a);
  DSPH_HIP(hipGetLastError());
#ifdef DSPH_QS_STAMPS
  if (getenv("DSPH_STAMPS_DUMP")) {
    std::vector<unsigned> h(NST);
    if (hipStreamSynchronize(stream) == hipSuccess && hipMemcpy(h.data(), d_stamps, NST * 4, hipMemcpyDeviceToHost) == hipSuccess)
      for (int w = 0; w < 8; ++w)
        for (int it = 0; it < 4; ++it) {
          const unsigned* r = &h[((size_t)w * 4 + it) * 10];
          fprintf(stderr, "QSSTAMP wave %d step %2d:", w, it);
          unsigned prev = r[0];
          for (int i = 1; i <= 8; ++i) {
            if (r[i] == 0) { fprintf(stderr, "      -"); continue; }
            fprintf(stderr, " %6u", r[i] - prev);
            prev = r[i];
          }
          fprintf(stderr, " | step %u | t0 %u\n", r[8] - r[0], r[0]);
        }
  }
#endif
  return DSPH_OK;
}

}  // namespace dsph
