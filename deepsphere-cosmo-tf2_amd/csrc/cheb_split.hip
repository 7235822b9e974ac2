// K > 5 on the fast kernels: a layer with more than five terms as a chain of passes with at most five.
//
// The fast fused kernels (strip, structured-tile) hold a 4-ring halo: K <= 5.  The reference's tutorials use K = 10
// (examples/quick_start.ipynb:118-127), BASELINE configs[3] K = 8; gnn_layers.py:17-28 takes any K.  The product identity of
// the Chebyshev polynomials,   T_{4+j} = 2 T_4 T_j - T_{|4-j|},   turns the tail of the sum of gnn_layers.py:131-150 into
// one more application of T_4:
//     sum_{k<K} T_k x W_k = sum_{k<=4} T_k x A_k + 2 T_4 u,     u = sum_{j=1..J} T_j x V_j,   J = K - 5,
//     V_j = W_{4+j} (K <= 13; beyond: split_next_kernel),   A_k = W_k - [1 <= 4-k <= J] V_{4-k} - [k >= 1 and 4+k <= J] V_{4+k}
// i.e. a (J+1)-term layer x -> u (its order-0 weights zero) followed by a 5-term layer on the concatenation [x | u] whose
// u-rows carry 2 I at order 4 and nothing else.  u has J + 1 = K - 4 terms: if that is still more than five, the same split
// again (K = 10: passes of 2, 5 and 5 terms; K = 13: 5, 5, 5).  Monomial basis (gnn_layers.py:283-286): L^{4+j} = L^4 L^j, the
// same chain with A_k = W_k and I in place of 2 I.
// The concatenation costs no copy: the inner pass writes [x | u] itself -- its weight matrix starts with Fin identity columns
// at order 0 (T_0 x I = x; exact in the fp32 and six-term arithmetics, x rounded to its bf16 hi + lo in the three-term one,
// which is what the next pass's contraction would make of x anyway).  Every pass is the ordinary fused forward
// (launch_cheb_fused): strip / structured / BFS-tile kernels as the plan and the shape say; bias and activation in the last.
// Whole graphs only (no halo columns, no level schedule): a sharded plan keeps the breadth-first-table kernel for K <= 9.
#include <algorithm>

#include "dsphere_common.h"

namespace dsph {

namespace {

constexpr int SPLIT_MAX_PASSES = 8;  // K <= 5 + 4 * 7 = 33

struct SplitShape {
  int L = 0;                      // passes after the first: levels 0 .. L, pass order L, L-1, .., 0
  int Kl[SPLIT_MAX_PASSES] = {};  // terms of level l: K, K-4, K-8, ...  (Kl[L] <= 5 < Kl[L-1])
  int32_t Cz = 0;                 // channels of the concatenation [x | u | zero padding to a multiple of four]
};

bool split_shape(int32_t Fin, int32_t Fout, int32_t K, SplitShape* s) {
  if (K <= 5 || Fin < 1 || Fout < 1) return false;
  int l = 0, k = K;
  s->Kl[0] = K;
  while (k > 5) {
    k -= 4;
    if (++l >= SPLIT_MAX_PASSES) return false;
    s->Kl[l] = k;
  }
  s->L = l;
  s->Cz = (Fin + Fout + 3) & ~3;
  return true;
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// Terms of the next level from those of this one: matching the coefficients of T_m on both sides of
//     sum_{k<Kl} T_k W_k = sum_{k<=4} T_k A_k + 2 T_4 sum_{j=1..J} T_j V_j,      2 T_4 T_j = T_{4+j} + T_{|4-j|},   J = Kl - 5
// gives, for m >= 5,  W_m = V_{m-4} + [m+4 <= J] V_{m+4}:   V_j = W_{4+j} - [j+8 <= J] V_{j+8}  (solved from the top order
// down; plain V_j = W_{4+j} while Kl <= 13), V_0 = 0.  Monomial basis: L^{4+j} = L^4 L^j, V_j = W_{4+j}.
//   wl [Fin * Kl, Fout] -> wn [Fin * (Kl - 4), Fout]
__global__ __launch_bounds__(256) void split_next_kernel(const float* __restrict__ wl, float* __restrict__ wn, int Fin, int Fout,
                                                         int Kl, int cheb) {
  const int J = Kl - 5, Kn = Kl - 4;
  const int64_t total = (int64_t)Fin * Fout * 8;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int r = (int)(e & 7);
    const int o = (int)((e >> 3) % Fout), f = (int)((e >> 3) / Fout);
    if (r == 0) wn[((int64_t)f * Kn) * Fout + o] = 0.f;
    int j = r == 0 ? 8 : r;
    while (j + 8 <= J) j += 8;
    float above = 0.f;  // V_{j+8}
    for (; j >= 1; j -= 8) {
      if (j > J) continue;
      const float v = wl[((int64_t)f * Kl + 4 + j) * Fout + o] - (cheb ? above : 0.f);
      wn[((int64_t)f * Kn + j) * Fout + o] = v;
      above = v;
    }
  }
}

// Weight matrix of one pass from the terms wl of its level (and wn = V, the next level's, for a pass that is not the deepest):
//   deepest pass (first = 1): rows (f, k), f < Fin, k < Kl      -- the layer x -> [x | u]: identity columns at k = 0, then W^l
//   other passes:             rows (c, k), c < Cz,  k < 5       -- on [x | u]: x rows carry A_k, u rows m I at k = 4,
//                             A_k = W_k - [1 <= 4-k <= J] V_{4-k} - [k >= 1 and 4+k <= J] V_{4+k}  (Chebyshev; monomial A_k = W_k)
//   columns: ident = 1: [Fin identity | Fout | zero padding] = Cz of them; ident = 0 (the last pass): Fout
__global__ __launch_bounds__(256) void split_weights_kernel(const float* __restrict__ wl, const float* __restrict__ wn,
                                                            float* __restrict__ out, int Fin, int Fout, int Kl, int first, int ident,
                                                            int Cz, int cheb) {
  const int Kp = first ? Kl : 5, rows_c = first ? Fin : Cz, ncol = ident ? Cz : Fout;
  const int64_t total = (int64_t)rows_c * Kp * ncol;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int col = (int)(e % ncol);
    const int rk = (int)(e / ncol);
    const int c = rk / Kp, k = rk % Kp;
    const int o = ident ? col - Fin : col;  // output channel of the layer, < 0 on the identity columns
    float v = 0.f;
    if (c < Fin) {  // a row of x
      if (o < 0) v = (k == 0 && col == c) ? 1.f : 0.f;
      else if (o < Fout) {
        v = wl[((int64_t)c * Kl + k) * Fout + o];
        if (!first && cheb) {
          const int J = Kl - 5, Kn = Kl - 4;
          if (4 - k >= 1 && 4 - k <= J) v -= wn[((int64_t)c * Kn + 4 - k) * Fout + o];
          if (k >= 1 && 4 + k <= J) v -= wn[((int64_t)c * Kn + 4 + k) * Fout + o];
        }
      }
    } else if (!first && c < Fin + Fout) {  // a row of u
      if (o >= 0 && o < Fout && k == 4 && o == c - Fin) v = cheb ? 2.f : 1.f;
    }
    out[e] = v;
  }
}

struct SplitLayout {
  size_t z[2] = {0, 0};  // the two concatenation buffers (the second only when there are three or more passes)
  size_t wgt = 0;        // weight matrices of the passes, back to back
  size_t wgt_pass[SPLIT_MAX_PASSES] = {};
  size_t lvl[SPLIT_MAX_PASSES] = {};  // terms W^l of the levels 1 .. L ([Fin * Kl, Fout]; level 0 is the layer's kernel)
  size_t sub_pass[SPLIT_MAX_PASSES] = {}, sub_bytes[SPLIT_MAX_PASSES] = {};  // every pass's own workspace (its packed weight
                                                                             // images survive the call: DSPH_FWD_KEEP_WEIGHTS)
  size_t total = 0;
};

// Fin / Fout / K of pass `l` (levels run L .. 0)
void pass_shape(const SplitShape& s, int32_t Fin, int32_t Fout, int l, int32_t* fin, int32_t* fout, int32_t* k) {
  *fin = l == s.L ? Fin : s.Cz;
  *fout = l == 0 ? Fout : s.Cz;
  *k = l == s.L ? s.Kl[l] : 5;
}

bool split_layout(const dsph_plan* p, int64_t N, int32_t Fin, int32_t Fout, int32_t K, int32_t precision, SplitShape* s, SplitLayout* lay) {
  if (!split_shape(Fin, Fout, K, s)) return false;
  // Everything a kept call (DSPH_FWD_KEEP_WEIGHTS) reads again sits at offsets that do not depend on the batch: the terms of
  // the levels, the weight matrices of the passes, the passes' own workspaces (their packed weight images come first in
  // them; only the deepest pass, whose input is x itself, may append a zero-padded copy of x -- so it goes last); the
  // concatenation buffers, N maps each, follow.
  size_t off = 0;
  for (int l = 1; l <= s->L; ++l) {
    lay->lvl[l] = off;
    off += align256((size_t)Fin * s->Kl[l] * Fout * sizeof(float));
  }
  lay->wgt = off;
  for (int l = s->L; l >= 0; --l) {
    int32_t fi, fo, k;
    pass_shape(*s, Fin, Fout, l, &fi, &fo, &k);
    lay->wgt_pass[l] = off;
    off += align256((size_t)fi * k * fo * sizeof(float));
  }
  for (int l = 0; l <= s->L; ++l) {
    int32_t fi, fo, k;
    pass_shape(*s, Fin, Fout, l, &fi, &fo, &k);
    lay->sub_pass[l] = off;
    lay->sub_bytes[l] = align256(fused_workspace_bytes(p, N, fi, fo, k, precision));
    off += lay->sub_bytes[l];
  }
  const size_t zb = align256((size_t)N * (size_t)p->n_cols * (size_t)s->Cz * sizeof(float));
  lay->z[0] = off; off += zb;
  if (s->L >= 2) { lay->z[1] = off; off += zb; }
  lay->total = off;
  return true;
}

}  // namespace

// whole graph, more than five terms, and every pass on the fused kernels
bool split_applicable(const dsph_plan* p, int32_t Fin, int32_t Fout, int32_t K) {
  SplitShape s;
  if (!p->fused || p->n_cols != p->n_rows || !p->levels.empty() || !split_shape(Fin, Fout, K, &s)) return false;
  for (int l = s.L; l >= 0; --l) {
    int32_t fi, fo, k;
    pass_shape(s, Fin, Fout, l, &fi, &fo, &k);
    if (k < 2 || !fused_supported(p, fi, fo, k)) return false;  // (K = 6 .. : the deepest pass has at least two terms)
  }
  return true;
}

size_t split_workspace_bytes(const dsph_plan* p, int64_t N, int32_t Fin, int32_t Fout, int32_t K, int32_t precision) {
  SplitShape s;
  SplitLayout lay;
  return split_layout(p, N, Fin, Fout, K, precision, &s, &lay) ? lay.total : 0;
}

// every pass's tables, so that a prepared forward allocates nothing
int split_prepare(const dsph_plan* p, int32_t K, int32_t Fin, int32_t Fout, int32_t flags) {
  SplitShape s;
  if (!split_shape(Fin, std::max(Fout, 1), K, &s)) return DSPH_OK;
  int rc = fused_prepare(p, s.Kl[s.L], Fin, flags & ~DSPH_PREPARE_RELEASE_HOST);
  if (rc == DSPH_OK) rc = fused_prepare(p, 5, s.Cz, flags);
  return rc;
}

int launch_split_forward(const dsph_plan* p, const float* x, const float* w, const float* bias, float* y, int64_t N, int32_t Fin,
                         int32_t Fout, int32_t K, int32_t basis, int32_t act, int32_t precision, void* workspace,
                         size_t workspace_bytes, hipStream_t stream, bool keep_weights) {
  SplitShape s;
  SplitLayout lay;
  if (!split_layout(p, N, Fin, Fout, K, precision, &s, &lay)) { set_error("split_forward: K = %d does not split", K); return DSPH_E_UNSUPPORTED; }
  if (!workspace || workspace_bytes < lay.total) { set_error("split_forward: workspace %zu bytes, need %zu", workspace_bytes, lay.total); return DSPH_E_WORKSPACE; }
  if (reinterpret_cast<uintptr_t>(workspace) & 15) { set_error("split_forward: the workspace must be 16-byte aligned"); return DSPH_E_BADARG; }
  unsigned char* ws = static_cast<unsigned char*>(workspace);
  const bool cheb = basis == DSPH_BASIS_CHEBYSHEV;
  const float alpha_rest = cheb ? 2.f : 1.f, beta_rest = cheb ? 1.f : 0.f;
  // the terms of the levels, then the weight matrices of all passes (tiny), then the passes
  auto blocks = [](int64_t total) { return dim3((unsigned)std::min<int64_t>((total + 255) / 256, 4096)); };
  auto level_w = [&](int l) -> const float* { return l == 0 ? w : reinterpret_cast<const float*>(ws + lay.lvl[l]); };
  // (the derived matrices are kept like the packed images: the library's own record says whether this workspace holds them)
  fused_images_begin(p, ws, fused_images_key(Fin, Fin, Fout & 0xff, K & 0x3f, Fout, precision, cheb, false, false) ^ ((uint64_t)K << 20), keep_weights);
  const bool derive = fused_images_claim(p, ws, IMG_SPLIT);
  for (int l = 0; l < s.L && derive; ++l) {
    hipLaunchKernelGGL(split_next_kernel, blocks((int64_t)Fin * Fout * 8), dim3(256), 0, stream, level_w(l),
                       reinterpret_cast<float*>(ws + lay.lvl[l + 1]), (int)Fin, (int)Fout, s.Kl[l], cheb ? 1 : 0);
    DSPH_HIP(hipGetLastError());
  }
  for (int l = s.L; l >= 0 && derive; --l) {
    int32_t fi, fo, k;
    pass_shape(s, Fin, Fout, l, &fi, &fo, &k);
    hipLaunchKernelGGL(split_weights_kernel, blocks((int64_t)fi * k * fo), dim3(256), 0, stream, level_w(l),
                       l == s.L ? (const float*)nullptr : level_w(l + 1), reinterpret_cast<float*>(ws + lay.wgt_pass[l]), (int)Fin,
                       (int)Fout, s.Kl[l], l == s.L ? 1 : 0, l == 0 ? 0 : 1, (int)s.Cz, cheb ? 1 : 0);
    DSPH_HIP(hipGetLastError());
  }
  const float* in = x;
  for (int l = s.L; l >= 0; --l) {
    int32_t fi, fo, k;
    pass_shape(s, Fin, Fout, l, &fi, &fo, &k);
    float* out = l == 0 ? y : reinterpret_cast<float*>(ws + lay.z[(s.L - l) & 1]);
    const int rc = launch_cheb_fused(p, in, reinterpret_cast<const float*>(ws + lay.wgt_pass[l]), l == 0 ? bias : nullptr, out, N, fi, fo,
                                     k, l == 0 ? act : DSPH_ACT_NONE, precision, alpha_rest, beta_rest, ws + lay.sub_pass[l],
                                     lay.sub_bytes[l], stream, DSPH_PART_ALL, keep_weights && !derive);
    if (rc != DSPH_OK) { fused_images_forget(p, ws); return rc; }
    in = out;
  }
  return DSPH_OK;
}

}  // namespace dsph
