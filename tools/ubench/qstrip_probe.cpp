// Stand-alone probe of the quad-strip kernel (csrc/cheb_qstrip_kernel.h): a square "base pixel" of S x S pixels in Morton
// order with a random 9-point operator, the interior cut into 64-column strips; (1) a small case against a float64
// restatement of the layer on this host (the probe's own, not oracle/: this is a tuning tool, not a test), (2) a large case
// timed with HIP events and reported per strip step.  Not part of the library, not part of the test suite.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I../../include -I../../deepsphere-cosmo-tf2_amd/csrc \
//         -x hip qstrip_probe.cpp -o qstrip_probe
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include <random>
#include <vector>

#include "cheb_qstrip.hip"

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
  bool f16 = false;
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
      s.tab = 0; s.tws = S / 16;  // (round 6: rows through a table of tile bases -- here the Morton plane itself)
      v.push_back(s);
    }
  }
  return v;
}

static double run(const Case& c, bool check, int reps) {
  const int S = c.S, N = c.N, K = 5, F = 64;
  const size_t M = (size_t)S * S;
  std::mt19937 rng(1234 + S);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  std::vector<float> g8(M * 8), gd(M), x((size_t)N * M * F), w((size_t)F * K * F), bias(F);
  for (auto& v : g8) v = 0.12f * U(rng);
  for (auto& v : gd) v = 0.3f * U(rng);
  for (auto& v : x) v = U(rng);
  for (auto& v : w) v = 0.06f * U(rng);
  for (auto& v : bias) v = 0.1f * U(rng);
  std::vector<QStrip> strips = cut(S, c.border, c.seg);
  float *d_g8, *d_gd, *d_x, *d_w, *d_b, *d_y;
  unsigned char* d_img;
  QStrip* d_s;
  CK(hipMalloc(&d_g8, g8.size() * 4)); CK(hipMalloc(&d_gd, gd.size() * 4)); CK(hipMalloc(&d_x, x.size() * 4));
  CK(hipMalloc(&d_w, w.size() * 4)); CK(hipMalloc(&d_b, bias.size() * 4)); CK(hipMalloc(&d_y, x.size() * 4));
  CK(hipMalloc(&d_img, qstrip_wimg_bytes())); CK(hipMalloc(&d_s, strips.size() * sizeof(QStrip)));
  CK(hipMemcpy(d_g8, g8.data(), g8.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_gd, gd.data(), gd.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_x, x.data(), x.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_w, w.data(), w.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_b, bias.data(), bias.size() * 4, hipMemcpyHostToDevice)); CK(hipMemset(d_y, 0, x.size() * 4));
  CK(hipMemcpy(d_s, strips.data(), strips.size() * sizeof(QStrip), hipMemcpyHostToDevice));
  std::vector<int32_t> prefix(1, 0);
  for (const QStrip& q : strips) prefix.push_back(prefix.back() + (q.y1 - q.y0));
  std::vector<int32_t> tab((size_t)(S / 16) * (S / 16) + 8, 0);
  for (int ty = 0; ty < S / 16; ++ty)
    for (int tx = 0; tx < S / 16; ++tx) tab[(size_t)ty * (S / 16) + tx] = (int32_t)(st_morton((unsigned)tx, (unsigned)ty) * 256u);
  int32_t* d_tab;
  CK(hipMalloc(&d_tab, tab.size() * 4));
  CK(hipMemcpy(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
  int32_t* d_p;
  CK(hipMalloc(&d_p, prefix.size() * 4));
  CK(hipMemcpy(d_p, prefix.data(), prefix.size() * 4, hipMemcpyHostToDevice));
  QStripLaunch L;
  L.prefix = d_p; L.tape_rows = prefix.back(); L.tab = d_tab;
  L.x = d_x; L.w = d_w; L.bias = d_b; L.y = d_y; L.wimg = d_img; L.strips = d_s; L.gvals8 = d_g8; L.gdiag = d_gd;
  L.x_rows = (int64_t)M; L.y_rows = (int64_t)M; L.N = N; L.nstrips = (int)strips.size(); L.Fin = F; L.Fout = F; L.act = DSPH_ACT_RELU;
  L.ld = F; L.num_cu = getenv("QS_NUM_CU") ? atoi(getenv("QS_NUM_CU")) : 256; /* (tuning: fewer workgroups than CUs) */ L.cheb = c.cheb; L.f16 = c.f16; L.prep_weights = true;
  // (tuning: QS_ALIAS=1: every map reads map 0's x and writes map 0's y -- the same instructions and the same data toggling with
  // an eighth of the memory traffic; QS_ZERO=1: x = 0 -- the same instructions and traffic, no toggling in the x operands)
  if (getenv("QS_ALIAS")) { L.x_rows = 0; L.y_rows = 0; }
  if (getenv("QS_ZERO")) CK(hipMemset(d_x, 0, x.size() * 4));
  if (launch_cheb_qstrip(L, nullptr) != DSPH_OK) exit(1);
  CK(hipDeviceSynchronize());
  double ms = 0;
  if (reps > 0) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    L.prep_weights = false;
    for (int r = 0; r < 2; ++r) launch_cheb_qstrip(L, nullptr);
    CK(hipEventRecord(e0, nullptr));
    for (int r = 0; r < reps; ++r) launch_cheb_qstrip(L, nullptr);
    CK(hipEventRecord(e1, nullptr));
    CK(hipEventSynchronize(e1));
    float t;
    CK(hipEventElapsedTime(&t, e0, e1));
    ms = t / reps;
    int64_t px = 0;
    for (const QStrip& s : strips) px += (int64_t)(s.y1 - s.y0) * s.w;
    int grid, pieces, wpp;
    const int64_t span = qstrip_split(L.num_cu, L.tape_rows, N, L.tape_rows / (int64_t)strips.size(), &grid, &pieces, &wpp);
    printf("S %d N %d: %zu strips, tape of %lld rows in %d pieces x %d workgroups (grid %d), about %lld steps each: %.3f ms per launch = %.3f us per step; %.1f Mpix-maps/s, "
           "%.1f GB/s of x + y\n", S, N, strips.size(), (long long)L.tape_rows, pieces, wpp, grid, (long long)span, ms, ms * 1e3 / span, px * N / ms * 1e-3,
           px * N * 512.0 / ms * 1e-6);
  }
  if (check) {
    std::vector<float> y(x.size());
    CK(hipMemcpy(y.data(), d_y, y.size() * 4, hipMemcpyDeviceToHost));
    // float64 restatement on the rectangle and its halo: T_0 = x, T_1 = L~ x, T_k = 2 L~ T_{k-1} - T_{k-2} (monomial: L~ T_{k-1})
    auto rid = [&](int X, int Y) { return (size_t)(st_spread((unsigned)X) | (st_spread((unsigned)Y) << 1)); };
    double worst = 0, ymax = 0;
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
          for (int o = 0; o < F; ++o) {
            double s = bias[o];
            for (int f = 0; f < F; ++f)
              for (int k = 0; k < K; ++k) s += T[k][r * F + f] * (double)w[((size_t)f * K + k) * F + o];
            s = s > 0 ? s : 0;
            const double got = y[((size_t)n * M + r) * F + o];
            worst = std::max(worst, fabs(got - s));
            ymax = std::max(ymax, fabs(s));
          }
        }
    }
    printf("check S %d N %d %s %s: max |err| %.3e of max |y| %.3e -> %.2e\n", S, N, c.cheb ? "chebyshev" : "monomial", c.f16 ? "f16x3" : "bf16x3", worst, ymax,
           worst / ymax);
  }
  hipFree(d_g8); hipFree(d_gd); hipFree(d_x); hipFree(d_w); hipFree(d_b); hipFree(d_y); hipFree(d_img); hipFree(d_s);
  return ms;
}

int main(int argc, char** argv) {
  if (argc > 1 && atoi(argv[1]) == 2) {  // (ablation / stamp builds: the timed case alone)
    run({1024, 4, 16, 1 << 30, true}, false, argc > 2 ? atoi(argv[2]) : 5);
    return 0;
  }
  if (argc > 1 && atoi(argv[1]) == 3) {  // (the full-chip timing alone: eight maps)
    run({1024, 8, 16, 1 << 30, true}, false, 3);
    return 0;
  }
  if (argc > 1 && atoi(argv[1]) == 4) {  // (the full-chip timing in the monomial basis)
    run({1024, 8, 16, 1 << 30, false}, false, 3);
    return 0;
  }
  run({128, 2, 16, 1 << 30, true}, true, 0);
  run({128, 1, 16, 40, false}, true, 0);
  run({128, 2, 16, 1 << 30, true, true}, true, 0);
  run({128, 1, 16, 1 << 30, false, true}, true, 0);
  if (argc > 1 && atoi(argv[1]) == 0) return 0;
  // one base pixel of nside 1024, four maps: 18 strips x 4 maps; then cut so that the items fill 256 CUs
  run({1024, 4, 16, 1 << 30, true}, false, 5);
  run({1024, 8, 16, 1 << 30, true}, false, 3);
  run({1024, 8, 16, 1 << 30, true, true}, false, 3);
  return 0;
}
