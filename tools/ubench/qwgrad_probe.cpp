// Stand-alone probe of the quad-strip weight-gradient kernel (csrc/cheb_qwgrad_kernel.h): a square of S x S pixels in Morton
// order with a random SYMMETRIC 9-point operator, the interior cut into 64-column strips; (1) small cases against a float64
// restatement on this host (the probe's own, not oracle/: a tuning tool, not a test), (2) a large case timed with HIP events.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I../../include -I../../deepsphere-cosmo-tf2_amd/csrc \
//         -x hip qwgrad_probe.cpp -o qwgrad_probe
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include <random>
#include <vector>

#include "cheb_qstrip.hip"
#include "cheb_qwgrad.hip"

namespace dsph {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int hip_fail(hipError_t e, const char* what) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); return DSPH_E_HIP; }
}  // namespace dsph
using namespace dsph;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static const int DX[8] = {-1, -1, 0, 1, 1, 1, 0, -1}, DY[8] = {0, 1, 1, 1, 0, -1, -1, -1};

struct Case {
  int S, N, border, seg;
  bool cheb;
};

static std::vector<QStrip> cut(int S, int border, int seg) {
  std::vector<QStrip> v;
  const int X0 = border, X1 = S - border, Y0 = border, Y1 = S - border;
  const int H = Y1 - Y0, nseg = (H + seg - 1) / seg;
  for (int sg = 0; sg < nseg; ++sg) {
    const int ya = Y0 + (int)((long)H * sg / nseg), yb = Y0 + (int)((long)H * (sg + 1) / nseg);
    for (int x0 = X0; x0 < X1; x0 += QS_USE) {
      QStrip s{};
      s.x0 = x0; s.w = std::min(QS_USE, X1 - x0); s.xs = x0 - QS_D;
      s.y0 = ya; s.y1 = yb; s.xlo = X0 - QS_D; s.xhi = X1 - 1 + QS_D; s.ylo = Y0 - QS_D; s.yhi = Y1 - 1 + QS_D;
      v.push_back(s);
    }
  }
  return v;
}

static double run(const Case& c, bool check, int reps) {
  const int S = c.S, N = c.N, K = 5, F = 64;
  const size_t M = (size_t)S * S;
  std::mt19937 rng(4321 + S);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  auto rid = [&](int X, int Y) { return (size_t)(st_spread((unsigned)X) | (st_spread((unsigned)Y) << 1)); };
  std::vector<float> u(M), g8(M * 8, 0.f), gd(M), x((size_t)N * M * F), dy((size_t)N * M * F);
  for (auto& v : u) v = U(rng);
  for (auto& v : gd) v = 0.3f * U(rng);
  for (int Y = 1; Y < S - 1; ++Y)
    for (int X = 1; X < S - 1; ++X)
      for (int d = 0; d < 8; ++d) g8[rid(X, Y) * 8 + d] = 0.06f * (u[rid(X, Y)] + u[rid(X + DX[d], Y + DY[d])]) + 0.01f * (float)((d & 3) + 1);
  for (auto& v : x) v = U(rng);
  for (auto& v : dy) v = U(rng);
  std::vector<QStrip> strips = cut(S, c.border, c.seg);
  float *d_g8, *d_gd, *d_x, *d_dy, *d_dw, *d_slabs;
  QStrip* d_s;
  CK(hipMalloc(&d_g8, g8.size() * 4)); CK(hipMalloc(&d_gd, gd.size() * 4)); CK(hipMalloc(&d_x, x.size() * 4));
  CK(hipMalloc(&d_dy, dy.size() * 4)); CK(hipMalloc(&d_dw, (size_t)F * K * F * 4)); CK(hipMalloc(&d_slabs, qwgrad_slab_bytes(256)));
  CK(hipMalloc(&d_s, strips.size() * sizeof(QStrip)));
  CK(hipMemcpy(d_g8, g8.data(), g8.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_gd, gd.data(), gd.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_x, x.data(), x.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_dy, dy.data(), dy.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_s, strips.data(), strips.size() * sizeof(QStrip), hipMemcpyHostToDevice));
  std::vector<int32_t> prefix(1, 0);
  for (const QStrip& q : strips) prefix.push_back(prefix.back() + (q.y1 - q.y0));
  int32_t* d_p;
  CK(hipMalloc(&d_p, prefix.size() * 4));
  CK(hipMemcpy(d_p, prefix.data(), prefix.size() * 4, hipMemcpyHostToDevice));
  QWgradLaunch L;
  L.x = d_x; L.dy = d_dy; L.dw = d_dw; L.slabs = d_slabs; L.strips = d_s; L.prefix = d_p; L.tape_rows = prefix.back();
  L.gvals8 = d_g8; L.gdiag = d_gd; L.x_rows = (int64_t)M; L.dy_rows = (int64_t)M; L.N = N; L.nstrips = (int)strips.size();
  L.lddy = F; L.lddw = F; L.num_cu = 256; L.cheb = c.cheb; L.accumulate = false;
  if (launch_cheb_qwgrad(L, nullptr) != DSPH_OK) exit(1);
  CK(hipDeviceSynchronize());
  double ms = 0;
  if (reps > 0) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 2; ++r) launch_cheb_qwgrad(L, nullptr);
    CK(hipEventRecord(e0, nullptr));
    for (int r = 0; r < reps; ++r) launch_cheb_qwgrad(L, nullptr);
    CK(hipEventRecord(e1, nullptr));
    CK(hipEventSynchronize(e1));
    float t;
    CK(hipEventElapsedTime(&t, e0, e1));
    ms = t / reps;
    int grid, pieces, wpp;
    const int64_t span = qstrip_split(256, L.tape_rows, N, L.tape_rows / (int64_t)strips.size(), &grid, &pieces, &wpp);
    printf("S %d N %d: %zu strips, tape of %lld rows in %d pieces x %d workgroups (grid %d), about %lld steps each: %.3f ms per launch = %.3f us per step\n",
           S, N, strips.size(), (long long)L.tape_rows, pieces, wpp, grid, (long long)span, ms, ms * 1e3 / span);
  }
  if (check) {
    std::vector<float> dw((size_t)F * K * F);
    CK(hipMemcpy(dw.data(), d_dw, dw.size() * 4, hipMemcpyDeviceToHost));
    std::vector<double> ref((size_t)F * K * F, 0.0);
    for (int n = 0; n < N; ++n) {
      std::vector<std::vector<double>> T(K, std::vector<double>(M * F, 0.0));
      for (size_t i = 0; i < M * F; ++i) T[0][i] = x[(size_t)n * M * F + i];
      for (int k = 1; k < K; ++k)
        for (int Y = 1; Y < S - 1; ++Y)
          for (int X = 1; X < S - 1; ++X) {
            const size_t r = rid(X, Y);
            for (int f = 0; f < F; ++f) {
              double s = (double)gd[r] * T[k - 1][r * F + f];
              for (int d = 0; d < 8; ++d) s += (double)g8[r * 8 + d] * T[k - 1][rid(X + DX[d], Y + DY[d]) * F + f];
              T[k][r * F + f] = (c.cheb && k >= 2) ? 2 * s - T[k - 2][r * F + f] : s;
            }
          }
      for (int Y = c.border; Y < S - c.border; ++Y)
        for (int X = c.border; X < S - c.border; ++X) {
          const size_t r = rid(X, Y);
          for (int f = 0; f < F; ++f)
            for (int k = 0; k < K; ++k) {
              const double tv = T[k][r * F + f];
              for (int o = 0; o < F; ++o) ref[((size_t)f * K + k) * F + o] += tv * (double)dy[((size_t)n * M + r) * F + o];
            }
        }
    }
    double worst = 0, dmax = 0, worst_k[5] = {0, 0, 0, 0, 0};
    for (int f = 0; f < F; ++f)
      for (int k = 0; k < K; ++k)
        for (int o = 0; o < F; ++o) {
          const size_t i = ((size_t)f * K + k) * F + o;
          const double e = fabs((double)dw[i] - ref[i]);
          worst = std::max(worst, e);
          worst_k[k] = std::max(worst_k[k], e);
          dmax = std::max(dmax, fabs(ref[i]));
        }
    printf("check S %d N %d %s seg %d: max |err| %.3e of max |dW| %.3e -> %.2e   (by order: %.1e %.1e %.1e %.1e %.1e)\n", S, N,
           c.cheb ? "chebyshev" : "monomial", c.seg, worst, dmax, worst / dmax, worst_k[0] / dmax, worst_k[1] / dmax, worst_k[2] / dmax,
           worst_k[3] / dmax, worst_k[4] / dmax);
  }
  hipFree(d_g8); hipFree(d_gd); hipFree(d_x); hipFree(d_dy); hipFree(d_dw); hipFree(d_slabs); hipFree(d_s); hipFree(d_p);
  return ms;
}

int main(int argc, char** argv) {
  if (argc > 1 && atoi(argv[1]) == 2) {
    run({1024, 4, 16, 1 << 30, true}, false, argc > 2 ? atoi(argv[2]) : 5);
    return 0;
  }
  if (argc > 1 && atoi(argv[1]) == 3) {  // (the full-chip timing alone: eight maps)
    run({1024, 8, 16, 1 << 30, true}, false, 3);
    return 0;
  }
  run({128, 2, 16, 1 << 30, true}, true, 0);
  run({128, 1, 16, 40, false}, true, 0);
  run({128, 1, 16, 40, true}, true, 0);
  if (argc > 1 && atoi(argv[1]) == 0) return 0;
  run({1024, 4, 16, 1 << 30, true}, false, 5);
  run({1024, 8, 16, 1 << 30, true}, false, 3);
  return 0;
}
