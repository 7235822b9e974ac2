// Strip form of the fused Chebyshev forward (round 3): the Clenshaw recurrence in registers, fed by the MFMA.
//
// Same mathematics as the other kernels (reference gnn_layers.py:131-150), evaluated in the other order.  Because
// T_k(L~) acts on the pixel axis and the weights on the channel axis, the two commute:
//     y = sum_k T_k(L~) x W_k  =  sum_k T_k(L~) z_k,   z_k = x W_k   (a dense product per pixel, no neighbours)
// and the sum is evaluated by Clenshaw's backward recurrence
//     b_{K-1} = z_{K-1},  b_j = z_j + 2 L~ b_{j+1} - b_{j+2},  y = z_0 + L~ b_1 - b_2
// (monomial basis: Horner, b_j = z_j + L~ b_{j+1}).  What that buys on gfx950:
//   * the MFMA comes FIRST and reads x, not recurrence results: its B operand is x split hi/lo, its A operand the
//     weights, and its 32x32 accumulator tile leaves the matrix pipe in exactly the layout the stencil wants --
//     lane = pixel column of a 32-pixel strip row, registers = 16 output channels -- so the recurrence runs on
//     the accumulators where they stand: no plane ever goes to LDS, none is re-read from it;
//   * the strip is streamed along y (one pixel row per step, every level lagging one row behind the level above):
//     the y-neighbours of a pixel are the previous / next rows held in the same lane's registers, the x-neighbours
//     are the adjacent lanes, fetched by DPP wave shifts folded into the multiply-adds (v_fmac_f32_dpp);
//   * x is read once per strip row (1.33 x in all: 24 of a strip's 32 columns are output, four on either side are
//     halo that is recomputed) by coalesced loads and goes into LDS once, as bf16 hi / lo MFMA fragments; y goes
//     from the accumulators through a free ring slot so that its stores are coalesced too; weights never move:
//     they live in the waves' registers for the whole launch (that is what the two roles are for, below);
//   * no halo in y at all inside a segment: a strip walks hundreds of rows.
// Roles.  A strip of 32 pixel columns is served by four waves: (role H | L) x (output-channel half ob).  H
// evaluates the upper levels K-1 .. S of the recurrence, L the lower ones S-1 .. 0 (S = K / 2) one step later;
// what crosses from H to L per step is one row of b_S and the dying row of b_{S+1}, 8 KiB through LDS.  Splitting
// the levels -- not the pixels -- between waves is what lets every wave keep the weights of ITS levels in
// registers (96 of them for three levels of a 64 x 32 block, hi and lo) and still fit two waves per SIMD; the
// LDS then holds only the ring of the last K+2 rows of x (as bf16 hi / lo MFMA fragments, 8 KiB per row), the
// hand-over rows and a ring of the last K+1 rows of L~ (nine values per pixel).
// What bounds it, and everything that was tried on it: DESIGN.md section 4.0, profiles/r3_strip_ablations.txt.
// A workgroup is two strips = eight waves, wave w and w + 4 share a SIMD: one H and one L, whose MFMA / VALU mix
// is complementary.
//
// Which pixels: rectangles of class-R tiles (cheb_struct.hip: tiles whose region is a verified Morton square of a
// 9-point stencil), cut into strips by the host (cheb_strip.hip); everything else stays with the tile kernels.
// Coordinates are the "virtual Morton decode" of the row index (x = even bits, y = odd bits), as there.
#pragma once

#include <type_traits>

#include "cheb_struct_kernel.h"
#include "dsphere_common.h"

namespace dsph {

constexpr int SP_PX = 32;       // pixel columns per strip = lanes & 31
constexpr int SP_DMAX = 4;      // halo columns on either side: K <= 5
constexpr int SP_USE = SP_PX - 2 * SP_DMAX;  // 24 output columns per strip
constexpr int SP_THREADS = 512;
constexpr int SP_STRIPS = 2;    // strips per workgroup
constexpr int SP_FRAG = 1024;   // bytes of one MFMA operand fragment (64 lanes x 16 B)

typedef float sp_f32x16 __attribute__((ext_vector_type(16)));
typedef float sp_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 sp_bf16x8 __attribute__((ext_vector_type(8)));

// One workgroup item: two strips side by side over the same rows.
struct StripPair {
  int32_t x0[SP_STRIPS];  // virtual x of the first OUTPUT column of each strip
  int32_t w[SP_STRIPS];   // output columns of each strip (<= SP_USE; 0: nothing to store)
  int32_t xs[SP_STRIPS];  // virtual x of lane 0 (x0 - D, or further left for the last, narrower strip of a rectangle, so that
                          // all 32 columns lie inside [xlo, xhi]: the hand-ordered kernel does not clamp x)
  int32_t y0, y1;         // output rows [y0, y1) (virtual y)
  int32_t xlo, xhi;       // loads are clamped to [xlo, xhi] x [ylo, yhi] (the rectangle and its halo)
  int32_t ylo, yhi;
};

struct StripArgs {
  const float* x;
  const float* bias;
  float* y;
  const unsigned char* wimg;  // strip_wprep_kernel: [role][ob][level of the role][ib][hi | lo][64 lanes][16 B]
  const float* gvals8;        // [rows][8] values of L~ by direction (kDirX / kDirY order)
  const float* gdiag;         // [rows]
  const StripPair* pairs;
  int64_t x_rows, y_rows;
  int npairs, N, Fin, Fout, ld, act;
#ifdef DSPH_SP_STAMPS
  unsigned* stamps;  // diagnostic build only: [8 waves][4 steps][9 points] s_memtime values (low words)
#endif
};

// Diagnostic build (-DDSPH_SP_STAMPS, tools/build_strip_ab.sh stamps; never the shipped library): s_memtime at the slot
// boundaries of sixteen steps of one workgroup, into a buffer nothing else reads.
#ifdef DSPH_SP_STAMPS
// (into LDS, copied out after the last stamped step: a stamp written to memory would sit in vmcnt and lengthen every wait
// that follows it)
#define SP_STAMP(id)                                                                                   \
  do {                                                                                                 \
    if (stamp_on) {                                                                                    \
      unsigned long long t_;                                                                           \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                       \
      if (lane == 0) reinterpret_cast<unsigned*>(smem + LDS_FLAG + 64)[(wave * 4 + (stamp_step & 3)) * 9 + (id)] = (unsigned)t_; \
      if ((id) == 8 && stamp_step == 3 && lane < 36)                                                   \
        a.stamps[wave * 36 + lane] = reinterpret_cast<unsigned*>(smem + LDS_FLAG + 64)[wave * 36 + lane]; \
    }                                                                                                  \
  } while (0)
#else
#define SP_STAMP(id)
#endif

// acc += src[lane - 1] * coef / src[lane + 1] * coef: the x-neighbours of a pixel sit in the adjacent lanes and come in
// through the DPP operand of the multiply-add itself (wave_shr:1 / wave_shl:1; lanes 0 / 63 read zero: bound_ctrl).
// hipcc's DPP combiner folds a dpp move into v_add / v_mul but not into v_fmac (it would emit v_mov_b32_dpp + v_fmac:
// 1.67 x the instructions), hence the asm.  Hazards the assembler statements hide from hipcc's hazard recogniser, and
// who covers them: (1) an MFMA result read by a VALU too early -- every chain on an accumulator starts with the
// compiler-visible centre term (sp_row), the asm terms depend on it; (2) a VALU result read through DPP, or as an
// MFMA operand, within two wait states -- sp_fence() after the last asm write of a finished row.
__device__ __forceinline__ void sp_fmac_left(float& acc, float src, float coef) {
  asm("v_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(src), "v"(coef));
}
__device__ __forceinline__ void sp_fmac_right(float& acc, float src, float coef) {
  asm("v_fmac_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(src), "v"(coef));
}
__device__ __forceinline__ void sp_fence(sp_f32x16& v) {
  asm volatile("s_nop 1"
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                 "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
}

// acc += cw * src[x-1] + cc * src[x] + ce * src[x+1] for the lane's 16 channels
__device__ __forceinline__ void sp_row(sp_f32x16& acc, const sp_f32x16& src, float cw, float cc, float ce) {
#pragma unroll
  for (int c = 0; c < 16; ++c) acc[c] = fmaf(cc, src[c], acc[c]);
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    float t = acc[c];
    sp_fmac_left(t, src[c], cw);
    sp_fmac_right(t, src[c], ce);
    acc[c] = t;
  }
}

// The nine values of L~ of this lane's pixel in row `rid`, times m: v[0] the diagonal, v[1 + d] direction d.
struct SpCoef {
  sp_f32x4 a, b;
  float d;
};
__device__ __forceinline__ SpCoef sp_coef_load(const StripArgs& a, unsigned rid) {
  SpCoef c;
  c.a = *reinterpret_cast<const sp_f32x4*>(reinterpret_cast<const char*>(a.gvals8) + (size_t)rid * 32u);
  c.b = *reinterpret_cast<const sp_f32x4*>(reinterpret_cast<const char*>(a.gvals8) + (size_t)rid * 32u + 16u);
  c.d = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.gdiag) + (size_t)rid * 4u);
  return c;
}

// stencil of one level: acc += m * L~ applied to the three rows of the window (w0: y-1, w1: y, w2: y+1)
// direction d -> (kDirX[d], kDirY[d]): d = 0 W, 1 NW(+y), 2 N(+y), 3 NE(+y), 4 E, 5 SE(-y), 6 S(-y), 7 SW(-y)
__device__ __forceinline__ void sp_stencil_lo(sp_f32x16& acc, const sp_f32x16& w0, const sp_f32x16& w1, const SpCoef& c, float m) {
  sp_row(acc, w0, m * c.b[3], m * c.b[2], m * c.b[1]);  // y-1: d = 7, 6, 5
  sp_row(acc, w1, m * c.a[0], m * c.d, m * c.b[0]);     // y  : d = 0, diag, 4
}
__device__ __forceinline__ void sp_stencil_hi(sp_f32x16& acc, const sp_f32x16& w2, const SpCoef& c, float m) {
  sp_row(acc, w2, m * c.a[1], m * c.a[2], m * c.a[3]);  // y+1: d = 1, 2, 3
}

// z += W'_level . x[row in ring slot]: 3 NIB MFMAs (hi.lo + lo.hi + hi.hi), B fragments from the ring
template <int NIB>
__device__ __forceinline__ void sp_mfma(sp_f32x16& acc, const sp_bf16x8 (&wr)[NIB][2], const unsigned char* __restrict__ smem,
                                        unsigned slot_addr) {
#pragma unroll
  for (int ib = 0; ib < NIB; ++ib) {
    const sp_bf16x8 bh = *reinterpret_cast<const sp_bf16x8*>(smem + slot_addr + ib * 2 * SP_FRAG);
    const sp_bf16x8 bl = *reinterpret_cast<const sp_bf16x8*>(smem + slot_addr + ib * 2 * SP_FRAG + SP_FRAG);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[ib][0], bl, acc, 0, 0, 0);  // small terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[ib][1], bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[ib][0], bh, acc, 0, 0, 0);
  }
}

template <int K> struct SpShape {
  static constexpr int D = K - 1;
  static constexpr int S = K / 2;        // L evaluates levels S-1 .. 0, H levels K-1 .. S
  static constexpr int GH = K - S;       // levels of H (the top one has no stencil)
  static constexpr int GL = S;           // levels of L
  static constexpr int RING = K + 2;     // rows of x held (as fragments): y_top - K .. y_top + 1
};

// multiplier of L~ in level j: Chebyshev 2 (1 in the last step), with the sign that turns "- b_{j+2}" into "+" (the
// planes are kept as s_j b_j with s = + + - - + + ...; the weights of level j carry s_j, strip_wprep_kernel)
template <bool CHEB> __device__ __forceinline__ constexpr float sp_mult(int j) {
  return CHEB ? (j == 0 ? 1.f : ((j & 1) ? -2.f : 2.f)) : 1.f;
}
__host__ __device__ constexpr float sp_wsign(bool cheb, int j) { return cheb && ((j & 3) >= 2) ? -1.f : 1.f; }

template <int K, int NIB, bool CHEB>
__global__ __launch_bounds__(SP_THREADS, 2) void cheb_strip_kernel(StripArgs a) {
  using SH = SpShape<K>;
  constexpr int D = SH::D, S = SH::S, GH = SH::GH, GL = SH::GL, RING = SH::RING;
  constexpr int ROWB = NIB * 2 * SP_FRAG;           // bytes of one ring row (hi + lo fragments of every inner block)
  constexpr int RINGB = RING * ROWB;                // per strip
  constexpr int HANDB = 2 * 2 * 4 * SP_FRAG;        // per strip: [ob][b_S row | b_{S+1} row][4 fragments]
  constexpr int STRIPB = RINGB + HANDB;
  constexpr int LDS_BIAS = SP_STRIPS * STRIPB;
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BIAS + 256];
  float* const sBias = reinterpret_cast<float*>(smem + LDS_BIAS);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool roleL = wave >= 4;
  const int strip = (wave >> 1) & 1, ob = wave & 1;
  const int q = (roleL ? 2 : 0) + ob;  // this wave's share of the strip's x rows: inner block q
  const int px = lane & 31, g = lane >> 5;
  const unsigned sbase = (unsigned)strip * STRIPB;
  const unsigned hand = sbase + RINGB + (unsigned)ob * (2 * 4 * SP_FRAG) + (unsigned)lane * 16u;

  if (tid < 64) sBias[tid] = (a.bias != nullptr && tid < a.Fout) ? a.bias[tid] : 0.f;
  for (int i = tid; i < LDS_BIAS / 16; i += SP_THREADS) reinterpret_cast<sp_f32x4*>(smem)[i] = sp_f32x4{0.f, 0.f, 0.f, 0.f};

  // weights of this wave's levels: registers for the whole launch
  constexpr int NLEV = GH > GL ? GH : GL;
  sp_bf16x8 wr[NLEV][NIB][2];
  {
    const int nlev = roleL ? GL : GH;
    const unsigned char* wp = a.wimg + ((size_t)((roleL ? 1 : 0) * 2 + ob) * NLEV) * (NIB * 2 * SP_FRAG) + lane * 16;
#pragma unroll
    for (int l = 0; l < NLEV; ++l)
#pragma unroll
      for (int ib = 0; ib < NIB; ++ib)
#pragma unroll
        for (int p = 0; p < 2; ++p)
          wr[l][ib][p] = l < nlev ? *reinterpret_cast<const sp_bf16x8*>(wp + ((size_t)(l * NIB + ib) * 2 + p) * SP_FRAG)
                                  : sp_bf16x8{};
  }
  const int G = gridDim.x, xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3;
  const int nslots = (G + 7 - xcd) / 8;
  const int p_begin = (int)((int64_t)a.npairs * xcd / 8), p_end = (int)((int64_t)a.npairs * (xcd + 1) / 8);
  const float floor_v = a.act == DSPH_ACT_RELU ? 0.f : -__builtin_huge_valf();
  const unsigned xrowb = (unsigned)a.Fin * 4u, yrowb = (unsigned)a.ld * 4u;
  __syncthreads();

  for (int p = p_begin + slot0; p < p_end; p += nslots) {
    const StripPair pr = a.pairs[p];
    const int x0 = strip ? pr.x0[1] : pr.x0[0], wuse = strip ? pr.w[1] : pr.w[0];
    const int X = min(max(x0 - D + px, pr.xlo), pr.xhi);
    const unsigned sX = st_spread((unsigned)X);
    const bool col_ok = px >= D && px < D + wuse;
    const int T = (pr.y1 - pr.y0) + 2 * D + 1;
    for (int n = 0; n < a.N; ++n) {
      const char* __restrict__ xmap = reinterpret_cast<const char*>(a.x) + (size_t)n * a.x_rows * xrowb;
      char* __restrict__ ymap = reinterpret_cast<char*>(a.y) + (size_t)n * a.y_rows * yrowb;
      // windows: wl[i] = the three newest rows of the level i steps below this role's first list entry
      //   H: wl[0] = level K-1 (its new row comes from the MFMA alone), wl[i] = level K-1-i, i < GH-1
      //   L: wl[0] = level S (rows arrive from H through LDS),           wl[i] = level S-i,   i < GL
      constexpr int NW = (GH - 1 > GL ? GH - 1 : GL) > 0 ? (GH - 1 > GL ? GH - 1 : GL) : 1;
      sp_f32x16 wl[NW][3];
      sp_f32x16 cin;  // L: the row of b_{S+1} that level S-1 adds
#pragma unroll
      for (int i = 0; i < NW; ++i)
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
          for (int c = 0; c < 16; ++c) wl[i][s][c] = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) cin[c] = 0.f;

      int ytop = pr.y0 - D;   // row whose x arrives in the ring for this step
      int slot_top = 0;       // ring slot of row ytop
      sp_f32x4 xa, xb;        // this lane's 8 channels of the row being fetched
      auto xfetch = [&](int yrow) __attribute__((always_inline)) {
        if (q < NIB) {
          const int Yc = min(max(yrow, pr.ylo), pr.yhi);
          const unsigned rid = sX | (st_spread((unsigned)Yc) << 1);
          const char* src = xmap + (size_t)rid * xrowb + (unsigned)(16 * q + 8 * g) * 4u;
          xa = *reinterpret_cast<const sp_f32x4*>(src);
          xb = *reinterpret_cast<const sp_f32x4*>(src + 16);
        }
      };
      auto xstore = [&](int slot) __attribute__((always_inline)) {
        if (q < NIB) {
          const float v[8] = {xa[0], xa[1], xa[2], xa[3], xb[0], xb[1], xb[2], xb[3]};
          sp_bf16x8 hi, lo;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const __bf16 h = (__bf16)v[j];
            hi[j] = h;
            lo[j] = (__bf16)(v[j] - (float)h);
          }
          unsigned char* dst = smem + sbase + (unsigned)slot * ROWB + (unsigned)q * 2 * SP_FRAG + (unsigned)lane * 16u;
          *reinterpret_cast<sp_bf16x8*>(dst) = hi;
          *reinterpret_cast<sp_bf16x8*>(dst + SP_FRAG) = lo;
        }
      };
      // prologue: row ytop into slot 0
      __syncthreads();  // (the previous map's last reads of the ring and of the hand-over block)
      xfetch(ytop);
      xstore(0);
      __syncthreads();

      for (int t = 0; t < T; ++t) {
        // ---- phase A -----------------------------------------------------------------------------------------------
        xfetch(ytop + 1);  // lands during the phase
        auto slot_of = [&](int back) __attribute__((always_inline)) -> unsigned {  // ring address of row ytop - back
          int s = slot_top - back;
          s += s < 0 ? RING : 0;
          return sbase + (unsigned)s * ROWB + (unsigned)lane * 16u;
        };
        auto rid_of = [&](int back) __attribute__((always_inline)) -> unsigned {
          const int Yc = min(max(ytop - back, pr.ylo), pr.yhi);
          return sX | (st_spread((unsigned)Yc) << 1);
        };
        if (!roleL) {
          // H: level K-1 = z alone
          sp_f32x16 top;
#pragma unroll
          for (int c = 0; c < 16; ++c) top[c] = 0.f;
          sp_mfma<NIB>(top, wr[0], smem, slot_of(0));
          sp_f32x16 out_row = top, out_c;
          if (GH >= 2) wl[0][2] = top;
#pragma unroll
          for (int i = 1; i < GH; ++i) {  // level j = K-1-i on row ytop - i
            constexpr int dummy = 0; (void)dummy;
            const int j = K - 1 - i;
            const SpCoef cf = sp_coef_load(a, rid_of(i));
            sp_f32x16 acc;
            if (CHEB && i >= 2) acc = wl[i - 2][0];
            else {
#pragma unroll
              for (int c = 0; c < 16; ++c) acc[c] = 0.f;
            }
            const float m = sp_mult<CHEB>(j);
            sp_stencil_lo(acc, wl[i - 1][0], wl[i - 1][1], cf, m);
            sp_fence(acc);
            sp_mfma<NIB>(acc, wr[i], smem, slot_of(i));
            sp_stencil_hi(acc, wl[i - 1][2], cf, m);
            sp_fence(acc);
            if (i < GH - 1) wl[i][2] = acc;
            else out_row = acc;
          }
          // hand-over: the new row of b_S, and the dying row of b_{S+1} (what level S-1 adds)
          out_c = (GH >= 2) ? wl[GH - 2][0] : out_row;
          unsigned char* hp = smem + hand;
#pragma unroll
          for (int f = 0; f < 4; ++f) {
            *reinterpret_cast<sp_f32x4*>(hp + f * SP_FRAG) = sp_f32x4{out_row[4 * f], out_row[4 * f + 1], out_row[4 * f + 2], out_row[4 * f + 3]};
            if (CHEB && GH >= 2 && S >= 1)
              *reinterpret_cast<sp_f32x4*>(hp + (4 + f) * SP_FRAG) = sp_f32x4{out_c[4 * f], out_c[4 * f + 1], out_c[4 * f + 2], out_c[4 * f + 3]};
          }
#pragma unroll
          for (int i = 0; i < GH - 1; ++i) {
            wl[i][0] = wl[i][1];
            wl[i][1] = wl[i][2];
          }
        } else {
          // L: level S-i on row ytop - (K-1-S) - i - 1, i = 1 .. GL
          sp_f32x16 yrow;
#pragma unroll
          for (int i = 1; i <= GL; ++i) {
            const int j = S - i;
            const int back = (K - 1 - S) + i + 1;
            const SpCoef cf = sp_coef_load(a, rid_of(back));
            sp_f32x16 acc;
            if (CHEB) acc = i >= 2 ? wl[i - 2][0] : cin;
            else {
#pragma unroll
              for (int c = 0; c < 16; ++c) acc[c] = 0.f;
            }
            const float m = sp_mult<CHEB>(j);
            sp_stencil_lo(acc, wl[i - 1][0], wl[i - 1][1], cf, m);
            sp_fence(acc);
            sp_mfma<NIB>(acc, wr[i - 1], smem, slot_of(back));
            sp_stencil_hi(acc, wl[i - 1][2], cf, m);
            sp_fence(acc);
            if (i < GL) wl[i][2] = acc;
            else yrow = acc;
          }
          // y of row ytop - K
          const int yr = ytop - K;
          if (col_ok && yr >= pr.y0 && yr < pr.y1) {
            const unsigned rid = sX | (st_spread((unsigned)yr) << 1);
            char* dst = ymap + (size_t)rid * yrowb + (unsigned)(32 * ob + 4 * g) * 4u;
#pragma unroll
            for (int tq = 0; tq < 4; ++tq) {
              const sp_f32x4 bv = *reinterpret_cast<const sp_f32x4*>(sBias + 32 * ob + 8 * tq + 4 * g);
              sp_f32x4 o;
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = fmaxf(yrow[4 * tq + e] + bv[e], floor_v);
              *reinterpret_cast<sp_f32x4*>(dst + tq * 32) = o;
            }
          }
#pragma unroll
          for (int i = 0; i < GL; ++i) {
            wl[i][0] = wl[i][1];
            wl[i][1] = wl[i][2];
          }
        }
        __syncthreads();
        // ---- phase B: the fetched row becomes fragments; L takes over this step's rows from H ------------------------
        {
          int s = slot_top + 1;
          s = s == RING ? 0 : s;
          xstore(s);
          slot_top = s;
        }
        if (roleL) {
          const unsigned char* hp = smem + hand;
#pragma unroll
          for (int f = 0; f < 4; ++f) {
            const sp_f32x4 r0 = *reinterpret_cast<const sp_f32x4*>(hp + f * SP_FRAG);
#pragma unroll
            for (int e = 0; e < 4; ++e) wl[0][2][4 * f + e] = r0[e];
            if (CHEB && GH >= 2) {
              const sp_f32x4 r1 = *reinterpret_cast<const sp_f32x4*>(hp + (4 + f) * SP_FRAG);
#pragma unroll
              for (int e = 0; e < 4; ++e) cin[4 * f + e] = r1[e];
            }
          }
        }
        ++ytop;
        __syncthreads();
      }
    }
  }
}



// =====================================================================================================================
// The K = 5, 64 -> 64 instantiation with a hand-ordered instruction stream (the headline shape).
//
// hipcc left to itself serialises the generic kernel above completely (ds_read -> wait -> MFMA, a 12-cycle s_nop behind
// every MFMA chain, the stencil behind that; a step took ~7.6 k cycles for ~2 k of work).  Here every hot instruction
// sits in an `asm volatile (... ::: "memory")` statement -- one MFMA, or the 12 multiply-adds of four channels of one
// source row ("quarter") -- and the statements alternate in SOURCE order; volatile statements and the plain LDS / global
// loads between them keep that order, so the source is the schedule.  hipcc still allocates the registers, computes the
// addresses and places the s_waitcnt for the loads.  Hazards are this code's business (see sp_settle()).
//
// Per step and wave, slots of one MFMA chain (12 MFMAs) beside the stencil work that does not depend on it:
//   H:  s0  z_4 -> b4[new]          |  b3[new]  = -2L~ (b4[-2], b4[-1])            (96, initialising)
//       s1  z_3 -> b3[new]          |  b2[new]  = b4[-2] + 2L~ (b3[-2], b3[-1])    (96, in place in the dying row of b4)
//       s2  z_2 -> b2[new]          |  b3[new] += -2L~ b4[new]                     (48)
//       s3                          |  b2[new] += 2L~ b3[new]                      (48)   -> LDS: b2[new], b3[-2]
//   L:  s0  z_0 + b2[-1] -> Y       |  b1[new]  = b3 + -2L~ (b2[-1], b2[0], b2[+1])(144, in place in the row from H)
//       s1  z_1 -> b1[new]          |  Y       += L~ (b1[-2], b1[-1])              (96)
//       s2                          |  Y       += L~ b1[new]                       (48)   -> y
// (signs: the planes are kept as s_j b_j, s = + + - - +, so that every "- b_{j+2}" is an accumulate-in-place.)
// Rows rotate through three register sets per plane (the step body is instantiated for the three phases), nothing is
// copied.  x arrives by LDS-DMA into the free slot of the ring as raw fp32 and is split into hi / lo fragments in place
// in phase B; the rows of L~ are fetched a step ahead.
// =====================================================================================================================

// Tuning builds only (tools/ab_strip.sh; results wrong by construction, the shipped library defines none of them):
// -DDSPH_SP_ABL=bits: 1 row shifts instead of wave shifts, 2 no MFMA, 4 no stencil quarters, 8 no x DMA, 16 no y store,
// 32 no rows of L~ fetched, 64 plain multiply-adds for the DPP ones, 128 no touches, 256 touches six rows ahead
#ifdef DSPH_SP_ABL
#define SP_ABL DSPH_SP_ABL
#else
#define SP_ABL 0
#endif
#if SP_ABL & 64  // (timing only: plain multiply-adds in place of the DPP ones)
#define SP_FD "v_fmac_f32_e32 "
#define SP_DPPL "\n\t"
#define SP_DPPR "\n\t"
#elif SP_ABL & 1
#define SP_DPPL " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define SP_DPPR " row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#else
#define SP_DPPL " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define SP_DPPR " wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#endif
#ifndef SP_FD
#define SP_FD "v_fmac_f32_dpp "
#endif
// cache-policy bits of the x loads and the y stores (tuning: -DDSPH_SP_LDF=1..4, -DDSPH_SP_STNT)
#if DSPH_SP_LDF == 1
#define SP_LDF " nt"
#elif DSPH_SP_LDF == 2
#define SP_LDF " sc0 sc1"
#elif DSPH_SP_LDF == 3
#define SP_LDF " sc1"
#elif DSPH_SP_LDF == 4
#define SP_LDF " sc0 sc1 nt"
#else
#define SP_LDF ""
#endif

// one MFMA: acc += wa * bb
__device__ __forceinline__ void sp_m(sp_f32x16& acc, const sp_bf16x8& wa, const sp_bf16x8& bb) {
  if (SP_ABL & 2) { asm volatile("" : "+v"(acc) : "v"(wa), "v"(bb) : "memory"); return; }
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(wa), "v"(bb) : "memory");
}
// the first MFMA of a chain: acc = wa * bb (+ c)
__device__ __forceinline__ void sp_m0(sp_f32x16& acc, const sp_bf16x8& wa, const sp_bf16x8& bb) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(wa), "v"(bb) : "memory");
}
__device__ __forceinline__ void sp_mc(sp_f32x16& acc, const sp_bf16x8& wa, const sp_bf16x8& bb, const sp_f32x16& c) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(acc) : "v"(wa), "v"(bb), "v"(c) : "memory");
}
// Hazard cover.  `n` wait states with the row as an operand: what follows in source order -- asm or compiler-made copies
// alike -- comes after them.  11: an MFMA chain's result before its first VALU reader; 1: VALU results before a DPP or
// MFMA reader.
template <int N> __device__ __forceinline__ void sp_settle(sp_f32x16& v) {
  if (SP_ABL & 131072) { asm volatile("" : "+v"(v) : : "memory"); return; }
  if (N == 11) asm volatile("s_nop 11" : "+v"(v) : : "memory");
  else asm volatile("s_nop 1" : "+v"(v) : : "memory");
}

// Quarter k of a row unit: channels 4k .. 4k+3 of acc (+)= c . (src[x-1], src[x], src[x+1]).
// INIT: the centre term initialises the accumulator; NEG: the coefficients enter negated.
template <bool INIT, bool NEG>
__device__ __forceinline__ void sp_q4(float& a0, float& a1, float& a2, float& a3, float s0, float s1, float s2, float s3,
                                      float cw, float cc, float ce) {
  if ((SP_ABL & 64) && NEG) { sp_q4<INIT, false>(a0, a1, a2, a3, s0, s1, s2, s3, cw, cc, ce); return; }
  if (!INIT && !NEG)
    asm volatile(
        "v_fmac_f32_e32 %0, %4, %9\n\tv_fmac_f32_e32 %1, %5, %9\n\tv_fmac_f32_e32 %2, %6, %9\n\tv_fmac_f32_e32 %3, %7, %9\n\t"
        SP_FD "%0, %4, %8" SP_DPPL SP_FD "%1, %5, %8" SP_DPPL SP_FD "%2, %6, %8" SP_DPPL SP_FD "%3, %7, %8" SP_DPPL
        SP_FD "%0, %4, %10" SP_DPPR SP_FD "%1, %5, %10" SP_DPPR SP_FD "%2, %6, %10" SP_DPPR SP_FD "%3, %7, %10" SP_DPPR
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
        : "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(cw), "v"(cc), "v"(ce)
        : "memory");
  else if (!INIT && NEG)
    asm volatile(
        "v_fma_f32 %0, %4, -%9, %0\n\tv_fma_f32 %1, %5, -%9, %1\n\tv_fma_f32 %2, %6, -%9, %2\n\tv_fma_f32 %3, %7, -%9, %3\n\t"
        SP_FD "%0, %4, -%8" SP_DPPL SP_FD "%1, %5, -%8" SP_DPPL SP_FD "%2, %6, -%8" SP_DPPL SP_FD "%3, %7, -%8" SP_DPPL
        SP_FD "%0, %4, -%10" SP_DPPR SP_FD "%1, %5, -%10" SP_DPPR SP_FD "%2, %6, -%10" SP_DPPR SP_FD "%3, %7, -%10" SP_DPPR
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
        : "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(cw), "v"(cc), "v"(ce)
        : "memory");
  else if (INIT && !NEG)
    asm volatile(
        "v_mul_f32_e32 %0, %4, %9\n\tv_mul_f32_e32 %1, %5, %9\n\tv_mul_f32_e32 %2, %6, %9\n\tv_mul_f32_e32 %3, %7, %9\n\t"
        SP_FD "%0, %4, %8" SP_DPPL SP_FD "%1, %5, %8" SP_DPPL SP_FD "%2, %6, %8" SP_DPPL SP_FD "%3, %7, %8" SP_DPPL
        SP_FD "%0, %4, %10" SP_DPPR SP_FD "%1, %5, %10" SP_DPPR SP_FD "%2, %6, %10" SP_DPPR SP_FD "%3, %7, %10" SP_DPPR
        : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3)
        : "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(cw), "v"(cc), "v"(ce)
        : "memory");
  else
    asm volatile(
        "v_mul_f32_e64 %0, %4, -%9\n\tv_mul_f32_e64 %1, %5, -%9\n\tv_mul_f32_e64 %2, %6, -%9\n\tv_mul_f32_e64 %3, %7, -%9\n\t"
        SP_FD "%0, %4, -%8" SP_DPPL SP_FD "%1, %5, -%8" SP_DPPL SP_FD "%2, %6, -%8" SP_DPPL SP_FD "%3, %7, -%8" SP_DPPL
        SP_FD "%0, %4, -%10" SP_DPPR SP_FD "%1, %5, -%10" SP_DPPR SP_FD "%2, %6, -%10" SP_DPPR SP_FD "%3, %7, -%10" SP_DPPR
        : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3)
        : "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(cw), "v"(cc), "v"(ce)
        : "memory");
}
template <bool INIT, bool NEG>
__device__ __forceinline__ void sp_q(sp_f32x16& acc, const sp_f32x16& src, int k, float cw, float cc, float ce) {
  if ((SP_ABL & 4) && !INIT) { asm volatile("" : "+v"(acc) : "v"(src), "v"(cw), "v"(cc), "v"(ce) : "memory"); return; }
  // (k is a constant after unrolling)
#define SP_QK(K4)                                                                                              \
  {                                                                                                            \
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;                                                              \
    if (!INIT) { a0 = acc[K4]; a1 = acc[K4 + 1]; a2 = acc[K4 + 2]; a3 = acc[K4 + 3]; }                         \
    sp_q4<INIT, NEG>(a0, a1, a2, a3, src[K4], src[K4 + 1], src[K4 + 2], src[K4 + 3], cw, cc, ce);              \
    acc[K4] = a0; acc[K4 + 1] = a1; acc[K4 + 2] = a2; acc[K4 + 3] = a3;                                        \
  }
  if (k == 0) SP_QK(0)
  else if (k == 1) SP_QK(4)
  else if (k == 2) SP_QK(8)
  else SP_QK(12)
#undef SP_QK
}

// The same quarter with its four centre terms as two packed multiply-adds (v_pk_fma_f32 takes no DPP operand, so only the
// centre column can be packed): the coefficient is the LOW half of the aligned register pair `cc` (op_sel_hi:[1,0,1]), the
// accumulators and the source row go in as aligned pairs and come back out as single registers for the DPP terms.
typedef float sp_f32x2 __attribute__((ext_vector_type(2)));
template <bool INIT, bool NEG>
__device__ __forceinline__ void sp_q4p(sp_f32x2& A01, sp_f32x2& A23, sp_f32x2 S01, sp_f32x2 S23, float cw, sp_f32x2 cc, float ce) {
  if (!INIT && !NEG)
    asm volatile("v_pk_fma_f32 %0, %2, %4, %0 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %1, %3, %4, %1 op_sel_hi:[1,0,1]"
                 : "+v"(A01), "+v"(A23) : "v"(S01), "v"(S23), "v"(cc) : "memory");
  else if (!INIT && NEG)
    asm volatile("v_pk_fma_f32 %0, %2, %4, %0 op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]\n\t"
                 "v_pk_fma_f32 %1, %3, %4, %1 op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]"
                 : "+v"(A01), "+v"(A23) : "v"(S01), "v"(S23), "v"(cc) : "memory");
  else if (INIT && !NEG)
    asm volatile("v_pk_mul_f32 %0, %2, %4 op_sel_hi:[1,0]\n\tv_pk_mul_f32 %1, %3, %4 op_sel_hi:[1,0]"
                 : "=&v"(A01), "=&v"(A23) : "v"(S01), "v"(S23), "v"(cc) : "memory");
  else
    asm volatile("v_pk_mul_f32 %0, %2, %4 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                 "v_pk_mul_f32 %1, %3, %4 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]"
                 : "=&v"(A01), "=&v"(A23) : "v"(S01), "v"(S23), "v"(cc) : "memory");
  float a0 = A01[0], a1 = A01[1], a2 = A23[0], a3 = A23[1];
  const float s0 = S01[0], s1 = S01[1], s2 = S23[0], s3 = S23[1];
  if (!NEG)
    asm volatile(
        SP_FD "%0, %4, %8" SP_DPPL SP_FD "%1, %5, %8" SP_DPPL SP_FD "%2, %6, %8" SP_DPPL SP_FD "%3, %7, %8" SP_DPPL
        SP_FD "%0, %4, %9" SP_DPPR SP_FD "%1, %5, %9" SP_DPPR SP_FD "%2, %6, %9" SP_DPPR SP_FD "%3, %7, %9" SP_DPPR
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
        : "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(cw), "v"(ce)
        : "memory");
  else
    asm volatile(
        SP_FD "%0, %4, -%8" SP_DPPL SP_FD "%1, %5, -%8" SP_DPPL SP_FD "%2, %6, -%8" SP_DPPL SP_FD "%3, %7, -%8" SP_DPPL
        SP_FD "%0, %4, -%9" SP_DPPR SP_FD "%1, %5, -%9" SP_DPPR SP_FD "%2, %6, -%9" SP_DPPR SP_FD "%3, %7, -%9" SP_DPPR
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
        : "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(cw), "v"(ce)
        : "memory");
  A01 = sp_f32x2{a0, a1};
  A23 = sp_f32x2{a2, a3};
}
template <bool INIT, bool NEG>
__device__ __forceinline__ void sp_qp(sp_f32x16& acc, const sp_f32x16& src, int k, float cw, sp_f32x2 cc, float ce) {
#if (SP_ABL & (4 | 64)) || !defined(DSPH_SP_PK)  // (measured: 12.22 ms packed against 11.94 plain, one box -- off unless -DDSPH_SP_PK)
  sp_q<INIT, NEG>(acc, src, k, cw, cc[0], ce);
#else
#define SP_QKP(K4)                                                                                               \
  {                                                                                                              \
    sp_f32x2 A01 = {0.f, 0.f}, A23 = {0.f, 0.f};                                                                 \
    if (!INIT) { A01 = sp_f32x2{acc[K4], acc[K4 + 1]}; A23 = sp_f32x2{acc[K4 + 2], acc[K4 + 3]}; }               \
    sp_q4p<INIT, NEG>(A01, A23, sp_f32x2{src[K4], src[K4 + 1]}, sp_f32x2{src[K4 + 2], src[K4 + 3]}, cw, cc, ce); \
    acc[K4] = A01[0]; acc[K4 + 1] = A01[1]; acc[K4 + 2] = A23[0]; acc[K4 + 3] = A23[1];                          \
  }
  if (k == 0) SP_QKP(0)
  else if (k == 1) SP_QKP(4)
  else if (k == 2) SP_QKP(8)
  else SP_QKP(12)
#undef SP_QKP
#endif
}

// the nine values of a row of L~ of this lane's pixel, by source row of the stencil:
//   y-1: (west, centre, east) = directions 7, 6, 5;  y: 0, diagonal, 4;  y+1: 1, 2, 3   (kDirX / kDirY)
struct SpC9 {
  sp_f32x4 a, b;  // directions 0..3, 4..7
  float d;        // diagonal
};
#define SP_LO0(c) (c).b[3], (c).b[2], (c).b[1]
#define SP_LO1(c) (c).a[0], (c).d, (c).b[0]
#define SP_HI(c) (c).a[1], (c).a[2], (c).a[3]
// the same with the centre coefficient as the aligned pair it is the low half of (sp_qp)
#define SP_LO0P(c) (c).b[3], sp_f32x2{(c).b[2], (c).b[3]}, (c).b[1]
#define SP_HIP(c) (c).a[1], sp_f32x2{(c).a[2], (c).a[3]}, (c).a[3]

// The row of L~ fetched during the previous step, made ready at the TOP of a step: doubled (MODE 1), halved (MODE 2) or
// only touched (MODE 0).  An asm statement on purpose: hipcc waits for the fetch where the statement stands -- before this
// step's own requests go out, when nothing younger is in flight -- instead of in front of the first use, where its
// vmcnt arithmetic (which cannot see the LDS-DMA requests) would wait for this step's requests too.
template <int MODE> __device__ __forceinline__ void sp_c9_ready(SpC9& c) {
  float v[9] = {c.a[0], c.a[1], c.a[2], c.a[3], c.b[0], c.b[1], c.b[2], c.b[3], c.d};
  if (MODE == 1)
    asm volatile(
        "v_add_f32_e32 %0, %0, %0\n\tv_add_f32_e32 %1, %1, %1\n\tv_add_f32_e32 %2, %2, %2\n\tv_add_f32_e32 %3, %3, %3\n\t"
        "v_add_f32_e32 %4, %4, %4\n\tv_add_f32_e32 %5, %5, %5\n\tv_add_f32_e32 %6, %6, %6\n\tv_add_f32_e32 %7, %7, %7\n\t"
        "v_add_f32_e32 %8, %8, %8"
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8])
        :
        : "memory");
  else if (MODE == 2)
    asm volatile(
        "v_mul_f32_e32 %0, 0.5, %0\n\tv_mul_f32_e32 %1, 0.5, %1\n\tv_mul_f32_e32 %2, 0.5, %2\n\tv_mul_f32_e32 %3, 0.5, %3\n\t"
        "v_mul_f32_e32 %4, 0.5, %4\n\tv_mul_f32_e32 %5, 0.5, %5\n\tv_mul_f32_e32 %6, 0.5, %6\n\tv_mul_f32_e32 %7, 0.5, %7\n\t"
        "v_mul_f32_e32 %8, 0.5, %8"
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8])
        :
        : "memory");
  else
    asm volatile("s_nop 0"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8])
                 :
                 : "memory");
  c.a = sp_f32x4{v[0], v[1], v[2], v[3]};
  c.b = sp_f32x4{v[4], v[5], v[6], v[7]};
  c.d = v[8];
}

template <bool CHEB>
__global__ __launch_bounds__(SP_THREADS, 2) void cheb_strip5_kernel(StripArgs a) {
  constexpr int K = 5, NIB = 4, D = 4, RING = K + 2;
  constexpr int ROWB = NIB * 2 * SP_FRAG;     // 8 KiB: one ring row of x (hi | lo fragments of the four inner blocks)
  constexpr int RINGB = RING * ROWB;          // 56 KiB per strip
  constexpr int HANDB = 2 * 2 * 4 * SP_FRAG;  // 16 KiB per strip: [ob][b2 row | b3 row][4 fragments]
  constexpr int CROWB = 32 * 32 + 32 * 4;     // one ring row of L~: [px][8 directions] + [px] diagonal
  constexpr int CRING = K + 1;                // rows of L~ held: ytop-5 .. ytop (written at the end of step ytop)
  constexpr int CRINGB = CRING * CROWB;       // 6,912 B per strip
  constexpr int STRIPB = RINGB + HANDB + CRINGB;
  constexpr int LDS_BIAS = SP_STRIPS * STRIPB;
  constexpr int LDS_FLAG = LDS_BIAS + 256;    // [strip][ob] hand-over counters
#ifdef DSPH_SP_STAMPS
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_FLAG + 64 + 8 * 4 * 9 * 4];
#else
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_FLAG + 64];
#endif
  float* const sBias = reinterpret_cast<float*>(smem + LDS_BIAS);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool roleL = wave >= 4;
  const int strip = (wave >> 1) & 1, ob = wave & 1;
  const int px = lane & 31, g = lane >> 5;
  const unsigned sbase = (unsigned)strip * STRIPB;
  const unsigned lane16 = (unsigned)lane * 16u;
  const unsigned hand = sbase + RINGB + (unsigned)ob * (2 * 4 * SP_FRAG) + lane16;
  const unsigned cbase = sbase + RINGB + HANDB;
  // hand-over counter of this (strip, ob): written by the L wave, polled by the H wave.  Through asm, not `volatile`:
  // hipcc drains vmcnt in front of a volatile access, i.e. it would wait for the x rows just requested.
  const unsigned flag_addr = (unsigned)LDS_FLAG + 8u * (unsigned)strip;
  auto flag_set = [&](int v) __attribute__((always_inline)) {
    asm volatile("ds_write_b32 %0, %1" : : "v"(flag_addr + 4u * (unsigned)ob), "v"(v) : "memory");
  };
  auto flag_get = [&](int which) __attribute__((always_inline)) -> int {
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(flag_addr + 4u * (unsigned)which) : "memory");
    return v;
  };

  if (tid < 64) sBias[tid] = (a.bias != nullptr && tid < a.Fout) ? a.bias[tid] : 0.f;
  for (int i = tid; i < LDS_BIAS / 16; i += SP_THREADS) reinterpret_cast<sp_f32x4*>(smem)[i] = sp_f32x4{0.f, 0.f, 0.f, 0.f};
  if (tid < 16) reinterpret_cast<int*>(smem + LDS_FLAG)[tid] = 0;

  const int G = gridDim.x, xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3;
  const int nslots = (G + 7 - xcd) / 8;
  // work items are (strip pair, map): a contiguous range of them per XCD (neighbouring pairs and the maps of a pair share an
  // L2), dealt to the XCD's workgroups in turn -- with the pairs sorted by height (cheb_fused.hip) every workgroup gets its
  // share of tall and short ones
  const int64_t n_items = (int64_t)a.npairs * a.N;
  const int p_begin = (int)(n_items * xcd / 8), p_end = (int)(n_items * (xcd + 1) / 8);
  const unsigned xrowb = (unsigned)a.Fin * 4u, yrowb = (unsigned)a.ld * 4u;

  // ---- pieces shared by the two roles -------------------------------------------------------------------------------
  // One barrier per step.  What it orders: the fragments of row ytop+1 and the row of L~ written at the end of a step
  // against their readers in the next one; H's hand-over rows against L's reads at the top of the next step.  (The other
  // direction -- L has taken the rows before H overwrites them a step later -- is the counter sFlag.)
  auto step_barrier = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  // Vector-memory instructions whose lanes touch 32 or 64 different cache lines are what this kernel cannot afford: the
  // texture addresser looks up about one line per cycle, and ~50 such instructions per step (x with lane = pixel, y straight
  // from the accumulators) kept it busy for ~3 k cycles per step.  So both directions are coalesced:
  //   x: wave (L, ob) fetches the pixels 16 ob .. 16 ob + 15 of row ytop+1, four lanes per pixel row and instruction (lane & 15
  //      = the pixel, lane >> 4 = one of the four 16-byte chunks of inner block k: 64 contiguous bytes of sixteen rows per
  //      instruction) at the top of the step, and at its end splits its four channels into hi / lo bf16 and writes the two
  //      8-byte halves where the MFMA fragments of ring slot `slot` want them;
  //   y: see the L role (through the free ring slot, eight lanes per 128-byte half row).
  // The requests are asm statements: hipcc must not move them, and xw_wait() is the one wait.
  auto morton_add = [](unsigned s, unsigned d_spread) __attribute__((always_inline)) -> unsigned {  // spread(x + d) from spread(x), spread(d)
    return ((s | 0xAAAAAAAAu) + d_spread) & 0x55555555u;
  };
  auto xfetch = [&](const char* xmap, unsigned sXc, unsigned sY, sp_f32x4 (&xv)[4]) __attribute__((always_inline)) {
    if (SP_ABL & 8) { xv[0] = sp_f32x4{0.f, 0.f, 0.f, 0.f}; xv[1] = xv[2] = xv[3] = xv[0]; return; }
    const char* src = xmap + (size_t)((sXc | sY) * xrowb + (unsigned)(lane >> 4) * 16u);
    asm volatile(
        "global_load_dwordx4 %0, %4, off" SP_LDF "\n\tglobal_load_dwordx4 %1, %4, off offset:64" SP_LDF "\n\t"
        "global_load_dwordx4 %2, %4, off offset:128" SP_LDF "\n\tglobal_load_dwordx4 %3, %4, off offset:192" SP_LDF
        : "=&v"(xv[0]), "=&v"(xv[1]), "=&v"(xv[2]), "=&v"(xv[3])
        : "v"(src)
        : "memory");
  };
  // byte offset, inside a ring row, of the 8-byte half this lane's chunk of instruction k = 0 fills: pixel 16 ob + (lane & 15),
  // chunk lane >> 4 -> fragment lane (pixel, (chunk >> 1) & 1), half chunk & 1 of inner block 0; instruction k is inner
  // block k, 2 KiB further on.  (Sixteen consecutive lanes = sixteen consecutive 16-byte fragment slots: no bank conflicts.)
  const unsigned xw_off = (unsigned)(16 * ob + (lane & 15) + 32 * ((lane >> 5) & 1)) * 16u + (unsigned)((lane >> 4) & 1) * 8u;
  auto xstore = [&](int slot, const sp_f32x4 (&xv)[4]) __attribute__((always_inline)) {
    if (SP_ABL & 32768) return;
    unsigned char* p = smem + sbase + (unsigned)slot * ROWB + xw_off;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // per pair of values: one packed convert for the two hi halves, a shift and a mask to get them back as floats, two
      // subtractions, one packed convert for the lo halves (six instructions; element by element hipcc makes eight of it)
      typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      u32x2 hi, lo;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float a0 = xv[k][2 * j], a1 = xv[k][2 * j + 1];
        const bf16x2 h = __builtin_convertvector(f32x2{a0, a1}, bf16x2);
        const unsigned hu = __builtin_bit_cast(unsigned, h);
        const float h0 = __builtin_bit_cast(float, hu << 16), h1 = __builtin_bit_cast(float, hu & 0xffff0000u);
        const bf16x2 l = __builtin_convertvector(f32x2{a0 - h0, a1 - h1}, bf16x2);
        hi[j] = hu;
        lo[j] = __builtin_bit_cast(unsigned, l);
      }
      *reinterpret_cast<u32x2*>(p + 2 * SP_FRAG * k) = hi;
      *reinterpret_cast<u32x2*>(p + 2 * SP_FRAG * k + SP_FRAG) = lo;
    }
  };
  // The rows of L~ go through a ring in LDS next to the x ring (slot of row ytop: cs_top), fetched once per strip by the wave
  // (L, ob 0) and stored as 2 L~ (Chebyshev) / L~ (monomial): lanes 0..31 fetch directions 0..3 and the diagonal of
  // pixel px, lanes 32..63 directions 4..7.
  auto cfetch = [&](unsigned rid, sp_f32x4& cv, float& cd) __attribute__((always_inline)) {
    if (SP_ABL & 262144) { cv = sp_f32x4{0.1f, 0.1f, 0.1f, 0.1f}; cd = 0.2f; return; }
    if (SP_ABL & 32) rid = 0;
    const char* pv = reinterpret_cast<const char*>(a.gvals8) + (size_t)rid * 32u + (unsigned)g * 16u;
    const char* pd = reinterpret_cast<const char*>(a.gdiag) + (size_t)rid * 4u;
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dword %1, %3, off" : "=&v"(cv), "=&v"(cd) : "v"(pv), "v"(pd) : "memory");
  };
  // every request of this wave has landed; the fetched registers are operands so that nothing that reads them moves in
  // front of the wait.  Straight-line code on purpose: around a branch hipcc copies the operands of an asm statement
  // BEFORE it, i.e. before the data is there.
  auto xw_wait = [&](sp_f32x4 (&xv)[4]) __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(xv[0]), "+v"(xv[1]), "+v"(xv[2]), "+v"(xv[3]) : : "memory");
  };
  auto cw_wait = [&](sp_f32x4& cv, float& cd) __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(cv), "+v"(cd) : : "memory");
  };
  auto cstore = [&](int slot, sp_f32x4 cv, float cd) __attribute__((always_inline)) {
    if (SP_ABL & 262144) return;
    if (CHEB) { cv = cv + cv; cd = cd + cd; }
    unsigned char* p = smem + cbase + (unsigned)slot * CROWB;
    *reinterpret_cast<sp_f32x4*>(p + (unsigned)px * 32u + (unsigned)g * 16u) = cv;
    if (g == 0) *reinterpret_cast<float*>(p + 1024 + (unsigned)px * 4u) = cd;
  };
  auto c9_read = [&](int slot) __attribute__((always_inline)) -> SpC9 {
    const unsigned char* p = smem + cbase + (unsigned)slot * CROWB;
    SpC9 c;
    if (SP_ABL & 16384) { c.a = sp_f32x4{0.1f, 0.1f, 0.1f, 0.1f}; c.b = c.a; c.d = 0.2f; asm volatile("" : "+v"(c.a), "+v"(c.b), "+v"(c.d)); return c; }
    c.a = *reinterpret_cast<const sp_f32x4*>(p + (unsigned)px * 32u);
    c.b = *reinterpret_cast<const sp_f32x4*>(p + (unsigned)px * 32u + 16u);
    c.d = *reinterpret_cast<const float*>(p + 1024 + (unsigned)px * 4u);
    return c;
  };
  // the MFMA chain of one level beside `NQ` quarters of stencil work: MFMA m is followed by the quarters that fall to
  // it; the fragments of inner block ib + 2 are requested once those of ib have been consumed (set: [lo | hi], ib & 1)
#if SP_ABL & 512  // (timing only: the B fragments are not read from LDS)
#define SP_FRLOAD(P) (wr[0][0][0])
#else
#define SP_FRLOAD(P) (*reinterpret_cast<const sp_bf16x8*>(P))
#endif
// order of the three terms of an inner block: (weights hi | lo = 0 | 1, x lo | hi = 0 | 1)
#ifdef DSPH_SP_ORDER2  // W_lo.x_hi, W_hi.x_hi, W_hi.x_lo: one operand stays where it is between consecutive MFMAs
#define SP_ORD_A(j) ((j) == 0 ? 1 : 0)
#define SP_ORD_B(j) ((j) == 2 ? 0 : 1)
#else                  // W_hi.x_lo, W_lo.x_hi, W_hi.x_hi: the small terms first
#define SP_ORD_A(j) ((j) == 1 ? 1 : 0)
#define SP_ORD_B(j) ((j) == 0 ? 0 : 1)
#endif
#define SP_CHAIN(ACC, FIRST_STMT, WLEV, FADDR, NQ, ...)                                                                \
  {                                                                                                                    \
    sp_bf16x8 fr[2][2];                                                                                                \
    _Pragma("unroll") for (int ib = 0; ib < 2; ++ib) {                                                                 \
      fr[ib][0] = SP_FRLOAD(smem + (FADDR) + ib * 2 * SP_FRAG + SP_FRAG);                                              \
      fr[ib][1] = SP_FRLOAD(smem + (FADDR) + ib * 2 * SP_FRAG);                                                        \
    }                                                                                                                  \
    _Pragma("unroll") for (int ib = 0; ib < NIB; ++ib) {                                                               \
      _Pragma("unroll") for (int j = 0; j < 3; ++j) {                                                                  \
        const int m = 3 * ib + j;                                                                                      \
        const sp_bf16x8& wa_ = wr[WLEV][ib][SP_ORD_A(j)];                                                              \
        const sp_bf16x8& bb_ = fr[ib & 1][SP_ORD_B(j)];                                                                \
        if (m == 0) { FIRST_STMT; }                                                                                    \
        else sp_m(ACC, wa_, bb_);                                                                                      \
        _Pragma("unroll") for (int qq = (m * (NQ)) / 12; qq < ((m + 1) * (NQ)) / 12; ++qq) { __VA_ARGS__; }            \
      }                                                                                                                \
      if (ib + 2 < NIB) {                                                                                              \
        fr[ib & 1][0] = SP_FRLOAD(smem + (FADDR) + (ib + 2) * 2 * SP_FRAG + SP_FRAG);                                  \
        fr[ib & 1][1] = SP_FRLOAD(smem + (FADDR) + (ib + 2) * 2 * SP_FRAG);                                            \
      }                                                                                                                \
    }                                                                                                                  \
  }

  __syncthreads();

  if (!roleL) {
    // =================================================================================================================
    // H: levels 4, 3, 2.  R[0] = b4 rows, R[1] = b3 rows; logical row s of a plane in phase PH = R[.][(s + PH) % 3].
    // =================================================================================================================
    sp_bf16x8 wr[3][NIB][2];
    {
      const unsigned char* wp = a.wimg + ((size_t)(0 * 2 + ob) * 3) * (NIB * 2 * SP_FRAG) + lane16;
#pragma unroll
      for (int l = 0; l < 3; ++l)
#pragma unroll
        for (int ib = 0; ib < NIB; ++ib)
#pragma unroll
          for (int p = 0; p < 2; ++p) wr[l][ib][p] = *reinterpret_cast<const sp_bf16x8*>(wp + ((size_t)(l * NIB + ib) * 2 + p) * SP_FRAG);
    }
    int handed = 0;  // hand-overs written so far by this wave; its L partner counts its top-of-step reads in the flag: the write
                     // at the end of step j (hand-over number j) may go ahead once L's read of step j has happened (flag >= j + 1)
    for (int q = p_begin + slot0; q < p_end; q += nslots) {
      const int p = q / a.N, nq = q - p * a.N;
      const StripPair pr = a.pairs[p];
      const int xs = strip ? pr.xs[1] : pr.xs[0];
      const unsigned sX = st_spread((unsigned)(xs + px));  // this lane's pixel column (the rows of L~)
      const int T3 = ((pr.y1 - pr.y0) + 2 * D + 1 + 3) / 3;
      auto spread_y = [&](int yrow) __attribute__((always_inline)) -> unsigned {
        return st_spread((unsigned)min(max(yrow, pr.ylo), pr.yhi)) << 1;
      };
      for (int n = nq; n <= nq; ++n) {
        sp_f32x16 R[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int c = 0; c < 16; ++c) R[i][s][c] = 0.f;
        int ytop = pr.y0 - D, slot_top = 0, cs_top = 0;
        step_barrier();  // (the previous map's last reads of the rings)
        {
          sp_f32x4 cv;
          float cd;
          cfetch(sX | spread_y(ytop - 1), cv, cd);
          cw_wait(cv, cd);
          if (ob == 0) cstore(CRING - 1, cv, cd);
        }
        step_barrier();  // (L: row ytop of x is in the ring)
        auto step = [&](auto ph_c) __attribute__((always_inline)) {
          constexpr int PH = decltype(ph_c)::value;
          constexpr int L0 = PH % 3, L1 = (PH + 1) % 3, L2 = (PH + 2) % 3;
#ifdef DSPH_SP_STAMPS
          const int stamp_step = ytop - (pr.y0 - D) - 60;
          const bool stamp_on = blockIdx.x == 72 && n == 1 && stamp_step >= 0 && stamp_step < 4;
#endif
          SP_STAMP(0);
          int snew = slot_top + 1;
          snew = snew == RING ? 0 : snew;
          auto slot_ix = [&](int back) __attribute__((always_inline)) -> int {
            int s = slot_top - back;
            s += s < 0 ? RING : 0;
            return s;
          };
          const unsigned f0 = sbase + (unsigned)slot_top * ROWB + lane16, f1 = sbase + (unsigned)slot_ix(1) * ROWB + lane16,
                         f2 = sbase + (unsigned)slot_ix(2) * ROWB + lane16;
          auto cslot_ix = [&](int back) __attribute__((always_inline)) -> int {
            int s = cs_top - back;
            s += s < 0 ? CRING : 0;
            return s;
          };
          const SpC9 c3 = c9_read(cslot_ix(1)), c2 = c9_read(cslot_ix(2));  // rows ytop-1 (level 3) and ytop-2 (level 2)
          constexpr bool N3 = CHEB;  // level 3 enters with -2 L~ (Chebyshev), level 2 with +2 L~
          SP_STAMP(1);
#if !(SP_ABL & 2048)
          // s0: z_4 -> b4[new] | b3[new] = -+ (b4[-2], b4[-1])
          SP_CHAIN(R[0][L2], sp_m0(R[0][L2], wa_, bb_), 0, f0, 8,
                   { if (qq < 4) sp_qp<true, N3>(R[1][L2], R[0][L0], qq, SP_LO0P(c3)); else sp_q<false, N3>(R[1][L2], R[0][L1], qq - 4, SP_LO1(c3)); })
          sp_settle<11>(R[0][L2]);
          sp_settle<1>(R[1][L2]);
          SP_STAMP(2);
          // s1: z_3 -> b3[new] | b2[new] = b4[-2] + (b3[-2], b3[-1]), in place in R[0][L0]
          SP_CHAIN(R[1][L2], sp_m(R[1][L2], wa_, bb_), 1, f1, 8,
                   { if (qq < 4) sp_qp<!CHEB, false>(R[0][L0], R[1][L0], qq, SP_LO0P(c2)); else sp_q<false, false>(R[0][L0], R[1][L1], qq - 4, SP_LO1(c2)); })
          sp_settle<11>(R[1][L2]);
          sp_settle<1>(R[0][L0]);
          SP_STAMP(3);
          // s2: z_2 -> b2[new] | b3[new] += b4[new]
          SP_CHAIN(R[0][L0], sp_m(R[0][L0], wa_, bb_), 2, f2, 4, { sp_qp<false, N3>(R[1][L2], R[0][L2], qq, SP_HIP(c3)); })
          sp_settle<11>(R[0][L0]);
          sp_settle<1>(R[1][L2]);
          SP_STAMP(4);
#endif
          // the row ytop of L~ (the c3 registers are free from here on; both waves of the strip ask for it -- the second hits
          // the L1 --, only ob 0 stores it: no branch between the request and cw_wait).  Its latency is H's to wait out:
          // H reaches the barrier well before L.
          sp_f32x4 cv;
          float cd;
          cfetch(sX | spread_y(ytop), cv, cd);
#if !(SP_ABL & 2048)
          // s3: b2[new] += b3[new]
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) sp_qp<false, false>(R[0][L0], R[1][L2], qq, SP_HIP(c2));
#endif
          SP_STAMP(5);
          // hand-over: b2[new] and the dying row of b3 -- once L has taken the previous pair
          while (flag_get(ob) <= handed) {}
          if (!(SP_ABL & 8192)) {
            unsigned char* hp = smem + hand;
#pragma unroll
            for (int f = 0; f < 4; ++f) {
              *reinterpret_cast<sp_f32x4*>(hp + f * SP_FRAG) = sp_f32x4{R[0][L0][4 * f], R[0][L0][4 * f + 1], R[0][L0][4 * f + 2], R[0][L0][4 * f + 3]};
              if (CHEB)
                *reinterpret_cast<sp_f32x4*>(hp + (4 + f) * SP_FRAG) = sp_f32x4{R[1][L0][4 * f], R[1][L0][4 * f + 1], R[1][L0][4 * f + 2], R[1][L0][4 * f + 3]};
            }
          }
          ++handed;
          SP_STAMP(6);
          cw_wait(cv, cd);
          if (ob == 0) cstore(cs_top, cv, cd);
          SP_STAMP(7);
          slot_top = snew;
          cs_top = cs_top + 1 == CRING ? 0 : cs_top + 1;
          ++ytop;
          step_barrier();
          SP_STAMP(8);
        };
        for (int t3 = 0; t3 < T3; ++t3) {
          step(std::integral_constant<int, 0>{});
          step(std::integral_constant<int, 1>{});
          step(std::integral_constant<int, 2>{});
        }
      }
    }
  } else {
    // =================================================================================================================
    // L: levels 1, 0.  R[0] = b2 rows (from H), R[1] = b1 rows (the new one starts as the b3 row from H), Y = y.
    // =================================================================================================================
    sp_bf16x8 wr[2][NIB][2];
    {
      const unsigned char* wp = a.wimg + ((size_t)(1 * 2 + ob) * 3) * (NIB * 2 * SP_FRAG) + lane16;
#pragma unroll
      for (int l = 0; l < 2; ++l)
#pragma unroll
        for (int ib = 0; ib < NIB; ++ib)
#pragma unroll
          for (int p = 0; p < 2; ++p) wr[l][ib][p] = *reinterpret_cast<const sp_bf16x8*>(wp + ((size_t)(l * NIB + ib) * 2 + p) * SP_FRAG);
    }
    const float floor_v = a.act == DSPH_ACT_RELU ? 0.f : -__builtin_huge_valf();
    // (Tried: s_setprio 2 here -- the L waves are the younger ones of their SIMDs and lose the issue arbitration to their H
    // partner: their slots shrink from 1.8 k + 1.4 k to 1.2 k + 1.2 k cycles, H's grow by as much, the step stays the same.)
#if SP_ABL & 4096
    asm volatile("s_setprio 2");
#endif
    int taken = 0;  // top-of-step reads done so far by this wave
    for (int q = p_begin + slot0; q < p_end; q += nslots) {
      const int p = q / a.N, nq = q - p * a.N;
      const StripPair pr = a.pairs[p];
      const int xs = strip ? pr.xs[1] : pr.xs[0], x0 = strip ? pr.x0[1] : pr.x0[0], wuse = strip ? pr.w[1] : pr.w[0];
      const unsigned sXc = st_spread((unsigned)(xs + 16 * ob + (lane & 15)));  // the pixel row its x chunks belong to
      const unsigned sXs = st_spread((unsigned)(xs + (lane >> 3)));            // the pixel row its y chunk belongs to (k = 0)
      const int pfirst = x0 - xs, plast = x0 - xs + wuse;                      // output pixels of the strip: [pfirst, plast)
      const int T3 = ((pr.y1 - pr.y0) + 2 * D + 1 + 3) / 3;
      auto spread_y = [&](int yrow) __attribute__((always_inline)) -> unsigned {
        return st_spread((unsigned)min(max(yrow, pr.ylo), pr.yhi)) << 1;
      };
      for (int n = nq; n <= nq; ++n) {
        const char* __restrict__ xmap = reinterpret_cast<const char*>(a.x) + (size_t)n * a.x_rows * xrowb;
        char* __restrict__ ymap = reinterpret_cast<char*>(a.y) + (size_t)n * a.y_rows * yrowb;
        sp_f32x16 R[2][3], Y;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int c = 0; c < 16; ++c) R[i][s][c] = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) Y[c] = 0.f;
        int ytop = pr.y0 - D, slot_top = 0, cs_top = 0;
        step_barrier();  // (the previous map's last reads of the rings)
        {
          sp_f32x4 xv[4];
          xfetch(xmap, sXc, spread_y(ytop), xv);
          xw_wait(xv);
          xstore(0, xv);
        }
        step_barrier();
        auto step = [&](auto ph_c) __attribute__((always_inline)) {
          constexpr int PH = decltype(ph_c)::value;
          constexpr int L0 = PH % 3, L1 = (PH + 1) % 3, L2 = (PH + 2) % 3;
#ifdef DSPH_SP_STAMPS
          const int stamp_step = ytop - (pr.y0 - D) - 60;
          const bool stamp_on = blockIdx.x == 72 && n == 1 && stamp_step >= 0 && stamp_step < 4;
#endif
          SP_STAMP(0);
          int snew = slot_top + 1;
          snew = snew == RING ? 0 : snew;
          auto slot_ix = [&](int back) __attribute__((always_inline)) -> int {
            int s = slot_top - back;
            s += s < 0 ? RING : 0;
            return s;
          };
          // the request of the step, a whole step ahead of its use (HBM latency under this load is ~2.5 k cycles, and the L
          // waves are the ones with registers to spare): this wave's sixteen pixels of x row ytop+1
          sp_f32x4 xv[4];
          xfetch(xmap, sXc, spread_y(ytop + 1), xv);
          // the rows H left at the end of the previous step: b2[new] -> R[0][L2] (the set that died then), b3 -> R[1][L2]
          if (!(SP_ABL & 8192)) {  // (at the first step of a map these are the last rows of the previous map: finite, and never reach an output)
            const unsigned char* hp = smem + hand;
            // (rows put together by concatenation: element by element hipcc loads into scratch registers and copies)
            typedef float sp_f32x8 __attribute__((ext_vector_type(8)));
            auto row16 = [&](const unsigned char* q) __attribute__((always_inline)) -> sp_f32x16 {
              const sp_f32x4 a0 = *reinterpret_cast<const sp_f32x4*>(q), a1 = *reinterpret_cast<const sp_f32x4*>(q + SP_FRAG),
                             a2 = *reinterpret_cast<const sp_f32x4*>(q + 2 * SP_FRAG), a3 = *reinterpret_cast<const sp_f32x4*>(q + 3 * SP_FRAG);
              const sp_f32x8 lo8 = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7), hi8 = __builtin_shufflevector(a2, a3, 0, 1, 2, 3, 4, 5, 6, 7);
              return __builtin_shufflevector(lo8, hi8, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
            };
            R[0][L2] = row16(hp);
            if (CHEB) R[1][L2] = row16(hp + 4 * SP_FRAG);
          }
          ++taken;
          flag_set(taken);  // (LDS operations of a wave complete in order: the reads above are done first)
          const unsigned f1 = sbase + (unsigned)slot_ix(4) * ROWB + lane16, f0 = sbase + (unsigned)slot_ix(5) * ROWB + lane16;
          auto cslot_ix = [&](int back) __attribute__((always_inline)) -> int {
            int s = cs_top - back;
            s += s < 0 ? CRING : 0;
            return s;
          };
          const SpC9 c1 = c9_read(cslot_ix(4));  // row ytop-4: level 1, 2 L~ as stored
          constexpr bool N1 = CHEB;  // level 1 enters with -2 L~, level 0 with +L~
          SP_STAMP(1);
#if !(SP_ABL & 1024)
          // s0: z_0 (+ b2[-1]) -> Y | b1[new] (= b3 row from H) -+= (b2[-1], b2[0], b2[+1])
          SP_CHAIN(Y, (CHEB ? sp_mc(Y, wa_, bb_, R[0][L0]) : sp_m0(Y, wa_, bb_)), 1, f0, 12,
                   { if (qq < 4) sp_qp<!CHEB, N1>(R[1][L2], R[0][L0], qq, SP_LO0P(c1));
                     else if (qq < 8) sp_q<false, N1>(R[1][L2], R[0][L1], qq - 4, SP_LO1(c1));
                     else sp_qp<false, N1>(R[1][L2], R[0][L2], qq - 8, SP_HIP(c1)); })
          sp_settle<11>(Y);
          sp_settle<1>(R[1][L2]);
          SP_STAMP(2);
          SpC9 c0 = c9_read(cslot_ix(5));  // row ytop-5: level 0, L~ (read here: c1 is dead, the L waves need the registers)
          if (CHEB) sp_c9_ready<2>(c0);
          // s1: z_1 -> b1[new] | Y += (b1[-2], b1[-1])
          SP_CHAIN(R[1][L2], sp_m(R[1][L2], wa_, bb_), 0, f1, 8,
                   { if (qq < 4) sp_qp<false, false>(Y, R[1][L0], qq, SP_LO0P(c0)); else sp_q<false, false>(Y, R[1][L1], qq - 4, SP_LO1(c0)); })
          sp_settle<11>(R[1][L2]);
          SP_STAMP(3);
          // s2: Y += b1[new]: y of row ytop - K
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) sp_qp<false, false>(Y, R[1][L2], qq, SP_HIP(c0));
#endif
          SP_STAMP(4);
          xw_wait(xv);  // (here, in front of this step's y stores: the wait is for everything in flight)
          SP_STAMP(5);
          // y of row ytop - K leaves through LDS so that the stores are coalesced (lane = pixel stores touch 32 rows per
          // instruction and hold the wave ~400 cycles each): into the 16 runs of 256 bytes that THIS wave will overwrite with
          // the fragments of row ytop+1 a few instructions further down -- pixels 16 ob .. 16 ob + 15 of every fragment of the
          // free ring slot: nobody else touches them, and the LDS operations of a wave complete in order -- as [pixel][32
          // channels], two pixels per run, the 16-byte chunks of a pixel XOR-ed with the run number (bank spread); back as eight
          // lanes per pixel (instruction k: pixels 8 k .. 8 k + 7), bias and activation floor, eight half rows of 128
          // contiguous bytes per store instruction.
          if (SP_ABL & 65536) {
            asm volatile("" : : "v"(Y) : "memory");
          } else {
            unsigned char* slot = smem + sbase + (unsigned)snew * ROWB + (unsigned)ob * 256u;
            auto run_base = [](unsigned run) -> unsigned { return (run >> 1) * 1024u + (run & 1u) * 512u; };
            {
              const unsigned run = (unsigned)px >> 1;
              unsigned char* wp = slot + run_base(run) + ((unsigned)px & 1u) * 128u;
#pragma unroll
              for (int tq = 0; tq < 4; ++tq)
                *reinterpret_cast<sp_f32x4*>(wp + (((unsigned)(2 * tq + g)) ^ (run & 7u)) * 16u) = sp_f32x4{Y[4 * tq], Y[4 * tq + 1], Y[4 * tq + 2], Y[4 * tq + 3]};
            }
            const int yr = ytop - K;
            const bool row_ok = yr >= pr.y0 && yr < ((SP_ABL & 16) ? pr.y0 + 1 : pr.y1);
            const unsigned sY = st_spread((unsigned)max(yr, 0)) << 1;
            const sp_f32x4 bv = *reinterpret_cast<const sp_f32x4*>(sBias + 32 * ob + 4 * (lane & 7));
            // (all four read-backs in flight before the first is looked at: one LDS latency, not four -- hipcc otherwise sinks
            // each read into the branch of its store)
            sp_f32x4 yo4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const unsigned pk = 8u * k + ((unsigned)lane >> 3), run = pk >> 1;
              yo4[k] = *reinterpret_cast<const sp_f32x4*>(slot + run_base(run) + (pk & 1u) * 128u + ((((unsigned)lane & 7u)) ^ (run & 7u)) * 16u);
            }
            asm volatile("" : "+v"(yo4[0]), "+v"(yo4[1]), "+v"(yo4[2]), "+v"(yo4[3]));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const unsigned pk = 8u * k + ((unsigned)lane >> 3);
              const sp_f32x4 yo = yo4[k];
              if (row_ok && (int)pk >= pfirst && (int)pk < plast) {
                const unsigned rid = morton_add(sXs, st_spread(8u * k)) | sY;
                sp_f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = fmaxf(yo[e] + bv[e], floor_v);
#ifdef DSPH_SP_STNT
                __builtin_nontemporal_store(o, reinterpret_cast<sp_f32x4*>(ymap + (size_t)rid * yrowb + (unsigned)(32 * ob + 4 * (lane & 7)) * 4u));
#else
                *reinterpret_cast<sp_f32x4*>(ymap + (size_t)rid * yrowb + (unsigned)(32 * ob + 4 * (lane & 7)) * 4u) = o;
#endif
              }
            }
          }
          SP_STAMP(6);
          xstore(snew, xv);
          SP_STAMP(7);
          slot_top = snew;
          cs_top = cs_top + 1 == CRING ? 0 : cs_top + 1;
          ++ytop;
          step_barrier();
          SP_STAMP(8);
        };
        for (int t3 = 0; t3 < T3; ++t3) {
          step(std::integral_constant<int, 0>{});
          step(std::integral_constant<int, 1>{});
          step(std::integral_constant<int, 2>{});
        }
      }
    }
  }
#undef SP_CHAIN
#undef SP_FRLOAD
}

}  // namespace dsph
