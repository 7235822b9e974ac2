// Host side of the strip kernel (cheb_strip_kernel.h): weight images, the cut of rectangles of class-R tiles
// into strip pairs, and the launch.
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "cheb_strip_kernel.h"

namespace dsph {

// Weight image of the strip kernel: one 1 KiB MFMA A-operand fragment per (role, ob, level of the role, ib, hi | lo);
// lane l, element e <- s_j * w[(16 ib + 8 (l >> 5) + e) * K + j][32 ob + (l & 31)], j the level (H: K-1-lev, L: S-1-lev),
// s_j the sign of the kept plane (sp_wsign).  Levels a role does not have are zero blocks.
__global__ __launch_bounds__(256) void strip_wprep_kernel(const float* __restrict__ w, unsigned char* __restrict__ out, int Fin,
                                                          int Fout, int K, int NIB, int NLEV, int cheb, int ld) {
  const int blk = blockIdx.x;  // ((role * 2 + ob) * NLEV + lev) * NIB + ib
  const int ib = blk % NIB, lev = (blk / NIB) % NLEV, ob = (blk / (NIB * NLEV)) & 1, role = blk / (NIB * NLEV * 2);
  const int S = K / 2;
  const int j = role == 0 ? K - 1 - lev : S - 1 - lev;
  const bool have = role == 0 ? lev < K - S : lev < S;
  const float sgn = sp_wsign(cheb != 0, j);
  unsigned char* base = out + (size_t)blk * 2 * SP_FRAG;
  for (int e = threadIdx.x; e < 512; e += 256) {
    const int l = e >> 3, i = e & 7;
    const int ch = 16 * ib + 8 * (l >> 5) + i, col = 32 * ob + (l & 31);
    const float v = (have && ch < Fin && col < Fout) ? sgn * w[((int64_t)ch * K + j) * ld + col] : 0.f;
    const __bf16 hi = (__bf16)v;
    const __bf16 lo = (__bf16)(v - (float)hi);
    reinterpret_cast<__bf16*>(base)[l * 8 + i] = hi;
    reinterpret_cast<__bf16*>(base + SP_FRAG)[l * 8 + i] = lo;
  }
}

bool strip_shape_ok(int32_t Fin, int32_t Fout, int32_t K) {
  return K == 5 && Fin == 64 && Fout == 64;  // (the generic kernel takes K = 2 .. 5: see launch_cheb_strip)
}

size_t strip_wimg_bytes(int32_t Fin, int32_t Fout, int32_t K) {
  (void)Fout;
  const int NIB = (Fin + 15) / 16, S = K / 2, NLEV = std::max(K - S, S);
  return (size_t)2 * 2 * NLEV * NIB * 2 * SP_FRAG;
}

int launch_cheb_strip(const StripLaunch& s, hipStream_t stream) {
  const int NIB = (s.Fin + 15) / 16, S = s.K / 2, NLEV = std::max(s.K - S, S);
  if (s.prep_weights) {
    hipLaunchKernelGGL(strip_wprep_kernel, dim3(2 * 2 * NLEV * NIB), dim3(256), 0, stream, s.w, s.wimg, (int)s.Fin, (int)s.Fout,
                       (int)s.K, NIB, NLEV, s.cheb ? 1 : 0, (int)s.ld);
    DSPH_HIP(hipGetLastError());
  }
  StripArgs a;
  a.x = s.x;
  a.bias = s.bias;
  a.y = s.y;
  a.wimg = s.wimg;
  a.gvals8 = s.gvals8;
  a.gdiag = s.gdiag;
  a.pairs = s.pairs;
  a.x_rows = s.x_rows;
  a.y_rows = s.y_rows;
  a.npairs = s.npairs;
  a.N = (int)s.N;
  a.Fin = s.Fin;
  a.Fout = s.Fout;
  a.ld = s.ld;
  a.act = s.act;
  // (one workgroup per CU, fewer when there are fewer (pair, map) items; cheb_fused.hip's strip_makespan mirrors this)
  const int grid = (int)std::max<int64_t>(8, std::min<int64_t>(s.num_cu, ((int64_t)s.npairs * s.N + 7) / 8 * 8));
  void (*kern)(StripArgs) = nullptr;
  // K = 5: the hand-ordered instantiation; DSPH_OPT_STRIP_GENERIC (diagnosis) and the other K: the generic kernel
  const bool generic = s.generic;
#define DSPH_SP_PICK(KK) (s.cheb ? cheb_strip_kernel<KK, 4, true> : cheb_strip_kernel<KK, 4, false>)
  switch (s.K) {
    case 2: kern = DSPH_SP_PICK(2); break;
    case 3: kern = DSPH_SP_PICK(3); break;
    case 4: kern = DSPH_SP_PICK(4); break;
    case 5: kern = generic ? DSPH_SP_PICK(5) : (s.cheb ? cheb_strip5_kernel<true> : cheb_strip5_kernel<false>); break;
    default: set_error("cheb_strip: K = %d", s.K); return DSPH_E_UNSUPPORTED;
  }
#undef DSPH_SP_PICK
#ifdef DSPH_SP_STAMPS
  static unsigned* d_stamps = nullptr;
  constexpr size_t NST = 8 * 4 * 9;
  if (!d_stamps) DSPH_HIP(hipMalloc(&d_stamps, NST * 4));
  DSPH_HIP(hipMemsetAsync(d_stamps, 0, NST * 4, stream));
  a.stamps = d_stamps;
#endif
  hipLaunchKernelGGL(kern, dim3(grid), dim3(SP_THREADS), 0, stream, a);
  DSPH_HIP(hipGetLastError());
#ifdef DSPH_SP_STAMPS
  if (getenv("DSPH_STAMPS_DUMP")) {
    std::vector<unsigned> h(NST);
    if (hipStreamSynchronize(stream) == hipSuccess && hipMemcpy(h.data(), d_stamps, NST * 4, hipMemcpyDeviceToHost) == hipSuccess)
      for (int w = 0; w < 8; ++w)
        for (int it = 0; it < 4; ++it) {
          const unsigned* r = &h[((size_t)w * 4 + it) * 9];
          fprintf(stderr, "SPSTAMP wave %d step %2d:", w, it);
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
