// Host side of the input-side strip kernel (cheb_istrip_kernel.h): weight image and launch.
#include <algorithm>
#include <vector>

#include "cheb_istrip_kernel.h"

namespace dsph {

// A-operand fragments of one 32-column block: [level k][term][64 lanes][16 B] (bf16 arithmetics) or [k][step][64 lanes][4 B]
// (exact fp32).  Lane (m = lane & 31 -> column 32 nb + m, kg = lane >> 5), slot j <- channel CH kg + j (zero for j >= CH and
// for channels the layer does not have); w is the layer's kernel [Fin_w * K, ld], row f K + k.
// CH == 1 (cheb_istrip1_kernel): ONE image, slot j <- level j of channel kg.
// pair (cheb_istrip1_kernel, two maps per wave): rows 0 .. 15 carry W in the slots of half 0, rows 16 .. 31 in those of half 1.
__global__ __launch_bounds__(256) void istrip_wprep_kernel(const float* __restrict__ w, unsigned char* __restrict__ out, int Fin_w,
                                                           int Fout, int K, int CH, int prec, int ld, int pair) {
  const int k = blockIdx.x;
  const int terms = is_terms(prec), tb = is_term_bytes(prec);
  unsigned char* base = out + (size_t)k * terms * tb;
  for (int e = threadIdx.x; e < 512; e += 256) {
    const int l = e >> 3, j = e & 7;
    const int ch = CH * (l >> 5) + j, col = l & 31;
    float v;
    if (CH == 1 && pair) v = (j < K && (l >> 5) == (col >> 4) && (col & 15) < Fout) ? w[(int64_t)j * ld + (col & 15)] : 0.f;
    else if (CH == 1) v = (j < K && (l >> 5) < Fin_w && col < Fout) ? w[((int64_t)(l >> 5) * K + j) * ld + col] : 0.f;
    else if (pair) v = (j < CH && j < Fin_w && (l >> 5) == (col >> 4) && (col & 15) < Fout) ? w[((int64_t)j * K + k) * ld + (col & 15)] : 0.f;
    else v = (j < CH && ch < Fin_w && col < Fout) ? w[((int64_t)ch * K + k) * ld + col] : 0.f;
    if (prec == DSPH_PREC_FP32) {
      reinterpret_cast<float*>(base)[j * 64 + l] = v;  // step j, lane l
    } else if (prec == DSPH_PREC_BF16X3) {
      const __bf16 hi = (__bf16)v;
      reinterpret_cast<__bf16*>(base)[l * 8 + j] = hi;
      reinterpret_cast<__bf16*>(base + 1024)[l * 8 + j] = (__bf16)(v - (float)hi);
    } else {
      const unsigned au = __builtin_bit_cast(unsigned, v);
      const float h = __builtin_bit_cast(float, au & 0xffff0000u);
      const float r = v - h;
      const unsigned ru = __builtin_bit_cast(unsigned, r);
      const float m = __builtin_bit_cast(float, ru & 0xffff0000u);
      reinterpret_cast<unsigned short*>(base)[l * 8 + j] = (unsigned short)(au >> 16);
      reinterpret_cast<unsigned short*>(base + 1024)[l * 8 + j] = (unsigned short)(ru >> 16);
      reinterpret_cast<__bf16*>(base + 2048)[l * 8 + j] = (__bf16)(r - m);
    }
  }
}

bool istrip_narrow(int32_t Fin_w) { return Fin_w <= 2; }  // real input channels: the level-packed kernel's layers
// layers that run two maps per wave (at most 16 output columns): one input channel on the level-packed kernel (1 -> 16), three
// or four on the four-channel form (4 -> 8, the layers behind a pseudo-convolution) -- half of each wave would idle otherwise
bool istrip_pairs(int32_t Fin_w, int32_t Fout) { return (Fin_w == 1 || (Fin_w >= 3 && Fin_w <= 4)) && Fout <= 16; }
bool istrip_shape_ok(int32_t Fin, int32_t K) { return K >= 2 && K <= 5 && Fin >= 4 && Fin <= 16 && Fin % 4 == 0; }

size_t istrip_wimg_bytes(int32_t K, int32_t precision) { return (size_t)K * is_terms(precision) * is_term_bytes(precision); }

void (*istrip_kernel_k2(int ch, int prec))(IStripArgs);  // cheb_istrip_inst.hip, one translation unit per K
void (*istrip_kernel_k3(int ch, int prec))(IStripArgs);
void (*istrip_kernel_k4(int ch, int prec))(IStripArgs);
void (*istrip_kernel_k5(int ch, int prec))(IStripArgs);

// Row segments per strip for a batch of N maps: the count whose busiest worker has the fewest steps when the kernel deals its
// items (a contiguous eighth per XCD, in turn to the XCD's waves).  heights: rows of every pair's strips; D: run-in rows.
int istrip_segments(const std::vector<int32_t>& heights, const std::vector<unsigned char>& second, int64_t N, int num_cu, int D,
                    bool narrow) {
  const int IS_WAVES = narrow ? IS1_WAVES : dsph::IS_WAVES;  // workers per workgroup
  if (narrow) num_cu *= IS1_WG_PER_CU;
  static const int cand[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32};
  int best = 1;
  int64_t best_span = -1;
  std::vector<int64_t> load;
  for (int S : cand) {
    const int64_t Q = (int64_t)heights.size() * 2 * S * N;
    if (Q > (1ll << 26)) break;
    const int G = (int)std::max<int64_t>(8, std::min<int64_t>(num_cu, ((Q + IS_WAVES - 1) / IS_WAVES + 7) / 8 * 8));
    int64_t worst = 0;
    for (int xcd = 0; xcd < 8; ++xcd) {
      const int nw = (G + 7 - xcd) / 8 * IS_WAVES;
      const int64_t q0 = Q * xcd / 8, q1 = Q * (xcd + 1) / 8;
      if (nw <= 0 || q1 <= q0) continue;
      load.assign((size_t)nw, 0);
      for (int64_t q = q0; q < q1; ++q) {
        const int64_t pe = q / (N * S);  // pair * 2 + strip
        const size_t p = (size_t)(pe >> 1);
        if ((pe & 1) && !second[p]) continue;
        const int sg = (int)((q / N) % S);
        const int H = heights[p], rows = (int)((int64_t)H * (sg + 1) / S - (int64_t)H * sg / S);
        if (rows > 0) load[(size_t)((q - q0) % nw)] += rows + 2 * D + 2;
      }
      for (int64_t v : load) worst = std::max(worst, v);
    }
    if (best_span < 0 || worst < best_span) { best_span = worst; best = S; }
    bool fine = true;  // no point cutting below ~16 rows
    for (int32_t h : heights) fine = fine && h / (S * 2) >= 16;
    if (!fine) break;
  }
  return best;
}

// one launch per 32-column block of the (at most 64-column) block the caller handles
int launch_cheb_istrip(const IStripLaunch& s, hipStream_t stream) {
  const bool narrow = istrip_narrow(s.Fin_w);
  const int CH = narrow ? 1 : (s.Fin <= 8 ? 4 : 8);
  const int waves = narrow ? IS1_WAVES : IS_WAVES, ncu = narrow ? IS1_WG_PER_CU * s.num_cu : s.num_cu;
  void (*kern)(IStripArgs) = nullptr;
  switch (s.K) {
    case 2: kern = istrip_kernel_k2(CH, s.precision); break;
    case 3: kern = istrip_kernel_k3(CH, s.precision); break;
    case 4: kern = istrip_kernel_k4(CH, s.precision); break;
    case 5: kern = istrip_kernel_k5(CH, s.precision); break;
    default: set_error("cheb_istrip: K = %d", s.K); return DSPH_E_UNSUPPORTED;
  }
  const size_t wb = istrip_wimg_bytes(s.K, s.precision);
  // (one input channel: the level-packed kernel; three or four: the four-channel form -- two channels fill both halves of
  // the level-packed kernel already)
  const bool pair = istrip_pairs(s.Fin_w, s.Fout) && (narrow || (s.Fin == 4 && CH == 4));
  const int64_t items = (int64_t)s.npairs * 2 * s.nseg * (pair ? (s.N + 1) / 2 : s.N);
  // eight workers per workgroup, one workgroup per CU; fewer when there are fewer items
  const int grid = (int)std::max<int64_t>(8, std::min<int64_t>(ncu, ((items + waves - 1) / waves + 7) / 8 * 8));
  for (int32_t cb = 0; cb < s.Fout; cb += 32) {
    unsigned char* img = s.wimg + (size_t)(cb / 32) * wb;
    const int32_t fo = std::min<int32_t>(32, s.Fout - cb);
    if (s.prep_weights) {
      hipLaunchKernelGGL(istrip_wprep_kernel, dim3(narrow ? 1 : s.K), dim3(256), 0, stream, s.w + cb, img, (int)s.Fin_w, (int)fo, (int)s.K, CH,
                         (int)s.precision, (int)s.ld, pair ? 1 : 0);
      DSPH_HIP(hipGetLastError());
    }
    IStripArgs a;
    a.x = s.x;
    a.bias = s.bias ? s.bias + cb : nullptr;
    a.y = s.y + cb;
    a.wimg = img;
    a.gvals8 = s.gvals8;
    a.gdiag = s.gdiag;
    a.pairs = s.pairs;
    a.x_rows = s.x_rows;
    a.y_rows = s.y_rows;
    a.npairs = s.npairs;
    a.N = (int)s.N;
    a.Fin = s.Fin;
    a.Fout = fo;
    a.ld = s.ld;
    a.act = s.act;
    a.nseg = s.nseg;
    a.cheb = s.cheb ? 1 : 0;
    a.pair = pair ? 1 : 0;
    a.pool = s.pool;  // (both kernels have the pooled epilogue; the caller has checked the rest: fused_pool_ok)
    a.ypool = s.ypool ? s.ypool + cb : nullptr;
    a.ypool_rows = s.ypool_rows;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(narrow ? IS1_THREADS : IS_THREADS), 0, stream, a);
    DSPH_HIP(hipGetLastError());
  }
  return DSPH_OK;
}

}  // namespace dsph
