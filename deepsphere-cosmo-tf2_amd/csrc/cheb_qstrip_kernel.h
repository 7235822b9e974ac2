// Quad-strip form of the fused Chebyshev forward (round 5): the Clenshaw recurrence of cheb_strip_kernel.h with FOUR pixels
// per lane.
//
// Same mathematics, same roles and the same streaming along y as the strip kernel of round 3 (cheb_strip_kernel.h; reference
// gnn_layers.py:131-150 evaluated as y = sum_k T_k(L~) (x W_k) by Clenshaw's recurrence on the MFMA accumulators).  What
// changed is the lane layout, and why is a measurement (tools/ubench/issue_share.hip, profiles/r5_ubench_issue_share.txt):
//   * a v_fmac_f32 with a DPP operand occupies a SIMD-32's vector pipe for FOUR cycles, a plain one for two, whatever the
//     number of waves; two thirds of the round-3 stencil (the west and east taps of a lane = pixel layout) were DPP;
//   * a matrix instruction next to vector work costs its 8 cycles of issue, not its 16 or 32 of matrix pipe: what a step
//     takes is the sum of the issue costs of both waves of the SIMD, the matrix pipe time is hidden.
// So the strip is 64 pixel columns wide and a lane owns the four neighbouring pixels 4p .. 4p+3 of a row (p = lane & 15), one
// per accumulator tile of a v_mfma_f32_16x16x32_bf16 (A = weights: 16 output channels, B = x: sixteen pixels 4p + t for tile
// t).  Of the six side taps of a pixel only the west tap of tile 0 and the east tap of tile 3 cross lanes (DPP row_shr:1 /
// row_shl:1 inside the 16-lane rows, which ARE the DPP rows): 1.5 of the 9 multiply-adds per value instead of 6.  And 56 of
// the 64 columns are output (halo 4 + 4 recomputed) instead of 24 of 32: 14 % fewer of everything per output pixel.
// The price: a wave owns 16 output channels (the tile's rows), so the B fragments of a strip row are read by four waves
// instead of two, and a lane needs the nine values of L~ of four pixels: the LDS carries 1.7 x the bytes per output pixel.
//
// Work split (one workgroup = one strip = eight waves, waves w and w + 4 share a SIMD):
//   wave = (role H | L) x (output-channel quarter oq).  H: levels 4, 3, 2; L: levels 1, 0 one step later, b2 and the dying b3
//   row crossing through LDS under a counter as in the strip kernel.  Every wave keeps the weights of its levels and quarter
//   in registers.  In the Chebyshev basis the matrix instructions of level 1 run on H as well (`H1` in the kernel: z_1 rides
//   the b3 row of the hand-over), which leaves L one chain of matrix instructions and H four (H 64, L 16 VGPRs of weights).
// Work items: the rows of all strips laid end to end form one tape per map; it is cut into P equal pieces (a piece = a few runs
//   of rows of consecutive strips, each run with its own 9 run-in steps), and w workgroups share a piece, workgroup j of them
//   taking the maps j, j + w, ... -- for a batch of N <= G maps w = N and P = G / N: every workgroup gets the same number of
//   rows whatever the strips' heights, and the N workgroups of a piece (neighbours on one XCD) walk the same rows of L~ at the
//   same time, one map each (the host picks P and w: qstrip_split).
// LDS (161,344 B): ring of 7 rows of x as bf16 hi | lo B-operand fragments (16 KiB per row: [32-channel block][hi | lo][tile]
//   1 KiB fragments), hand-over 4 x 8 KiB, ring of 6 rows of L~ (2,304 B per row: the diagonal and the eight directions, each a
//   [p][tile] vector: a lane reads the 16 bytes of its four pixels per direction, 9 reads per level-row), counters.
// x: fetched a step ahead and split into fragments at the end of the step -- by all eight waves, half a tile each (two 16-byte
//   loads per lane), when level 1's matrix work runs on H (Chebyshev basis); else by the H wave of quarter q, tile q (four loads).  y: straight from the accumulators, 16 pixels x 64 contiguous bytes per instruction.
#pragma once

#include <type_traits>

#include "cheb_struct_kernel.h"
#include "dsphere_common.h"

namespace dsph {

constexpr int QS_PX = 64;                       // pixel columns of a strip: 16 lanes x 4 tiles
constexpr int QS_D = 4;                         // halo columns on either side (K = 5)
constexpr int QS_USE = QS_PX - 2 * QS_D;        // 56 output columns
constexpr int QS_THREADS = 512;
constexpr int QS_FRAG = 1024;                   // bytes of one MFMA operand fragment (64 lanes x 16 B)

typedef float qs_f4 __attribute__((ext_vector_type(4)));
typedef __bf16 qs_bf8 __attribute__((ext_vector_type(8)));
typedef unsigned qs_u2 __attribute__((ext_vector_type(2)));

// one row of a plane as a wave holds it: tile t = pixel 4 p + t, element e = output channel 16 oq + 4 (lane >> 4) + e
struct QRow {
  qs_f4 t[4];
};

// One work item (with a map of the batch): a strip of up to 56 output columns over rows [y0, y1).
struct QStrip {
  int32_t x0, w;      // virtual x of the first output column, output columns (<= QS_USE)
  int32_t xs;         // virtual x of column 0 (x0 - D; columns are clamped to [xlo, xhi] when loaded)
  int32_t y0, y1;     // output rows [y0, y1)
  int32_t xlo, xhi;   // the rectangle and its halo
  int32_t ylo, yhi;
  int32_t tab, tws;   // the rectangle's table of tile bases (offset into QStripArgs::tab) and its row stride: pixel (x, y) of the
                      // strip's plane is row tab[(y >> 4) * tws + (x >> 4)] + morton(x & 15, y & 15)
  int32_t pad[1];
};

struct QStripArgs {
  const float* x;
  const float* bias;
  float* y;
  const unsigned char* wimg;  // qstrip_wprep_kernel: [role][oq][level of the role (3)][32-channel block][hi | lo][64 lanes][16 B]
  const float* gvals8;        // [rows][8] values of L~ by direction (kDirX / kDirY order)
  const float* gdiag;         // [rows]
  const QStrip* strips;
  const int32_t* tab;         // tile-base tables of the strips' rectangles (QStrip::tab, ::tws): row numbers of 16 x 16 Morton squares
  const int32_t* prefix;      // [nstrips + 1] rows of the strips before strip s (the "tape" of one map; prefix[nstrips] = all rows)
  int64_t x_rows, y_rows;
  int nstrips, N, Fin, Fout, ld, act;
  int pieces, wg_per_piece;   // the tape is cut into `pieces`; `wg_per_piece` workgroups share a piece, each taking every wg_per_piece-th map
  float xsc, xsc_inv;         // f16 arithmetic: x is split as x * xsc (a power of two, DSPH_OPT_F16_XEXP), the store multiplies by xsc_inv
#ifdef DSPH_QS_STAMPS
  unsigned* stamps;
#endif
};

// Diagnostic build (-DDSPH_QS_STAMPS; never the shipped library): s_memtime at the slot boundaries of four steps of one
// workgroup, kept in registers of lane 0 ... written straight to a buffer nothing else reads (the waits it adds are the
// price of the diagnosis: compare stamps with stamps, not with the plain build).
#ifdef DSPH_QS_STAMPS
#define QS_STAMP(id)                                                                                          \
  do {                                                                                                        \
    if (stamp_on) {                                                                                           \
      unsigned long long t_;                                                                                  \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                              \
      if (lane == 0) a.stamps[(wave * 4 + (stamp_step & 3)) * 10 + (id)] = (unsigned)t_;                      \
    }                                                                                                         \
  } while (0)
#define QS_STAMP_DECL                                                                                         \
  const int stamp_step = ytop - (st.y0 - D) - 60;                                                             \
  const bool stamp_on = blockIdx.x == 72 && stamp_step >= 0 && stamp_step < 4;
#else
#define QS_STAMP(id)
#define QS_STAMP_DECL
#endif

// Tuning builds only (results wrong by construction): -DDSPH_QS_ABL=bits: 2 no MFMA, 4 no stencil units, 8 no x loads,
// 16 no y stores, 64 plain multiply-adds for the DPP ones, 512 B fragments not read from LDS, 1024 values of L~ not read from
// LDS, 2048 no hand-over rows, 4096 no fragment stores of x
#ifdef DSPH_QS_ABL
#define QS_ABL DSPH_QS_ABL
#else
#define QS_ABL 0
#endif

#if QS_ABL & 64
#define QS_FD "v_fmac_f32_e32 "
#define QS_DPPL "\n\t"
#define QS_DPPR "\n\t"
#else
#define QS_FD "v_fmac_f32_dpp "
#define QS_DPPL " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define QS_DPPR " row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#endif

// one MFMA: acc (+)= wa . bb   (16 output channels x 16 pixels x 32 input channels); F16: the operands are f16 pairs (same
// registers, same rate), the three-term split then keeps 11 + 11 mantissa bits of both operands (DSPH_PREC_F16X3)
template <bool F16> __device__ __forceinline__ void qs_m(qs_f4& acc, const qs_bf8& wa, const qs_bf8& bb) {
  if (QS_ABL & 2) { asm volatile("" : "+v"(acc) : "v"(wa), "v"(bb) : "memory"); return; }
  if (F16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(wa), "v"(bb) : "memory");
  else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(wa), "v"(bb) : "memory");
}
template <bool F16> __device__ __forceinline__ void qs_m0(qs_f4& acc, const qs_bf8& wa, const qs_bf8& bb) {
  if (QS_ABL & 2) { acc = qs_f4{0.f, 0.f, 0.f, 0.f}; asm volatile("" : "+v"(acc) : "v"(wa), "v"(bb) : "memory"); return; }
  if (F16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(wa), "v"(bb) : "memory");
  else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(wa), "v"(bb) : "memory");
}
// Hazard cover (the asm statements hide their instructions from hipcc's hazard recogniser).  N = 9: the result of a chain of
// 16x16x32 MFMAs (4 passes) before its first vector reader; N = 1: vector results before a DPP or a matrix reader.
template <int N> __device__ __forceinline__ void qs_settle(QRow& r) {
  if (N == 9) asm volatile("s_nop 9" : "+v"(r.t[0]), "+v"(r.t[1]), "+v"(r.t[2]), "+v"(r.t[3]) : : "memory");
  else asm volatile("s_nop 1" : "+v"(r.t[0]), "+v"(r.t[1]), "+v"(r.t[2]), "+v"(r.t[3]) : : "memory");
}

// One unit of stencil work: element e of the four tiles of a row, one source row: 12 multiply-adds,
//   acc[t] (+)= cw[t] src[t-1] + cc[t] src[t] + ce[t] src[t+1],  src[-1] = tile 3 of the lane to the left (row_shr:1),
//   src[4] = tile 0 of the lane to the right (row_shl:1).
// INIT: the centre terms initialise the accumulators; NEG: the coefficients enter negated.
template <bool INIT, bool NEG>
__device__ __forceinline__ void qs_u12(float& a0, float& a1, float& a2, float& a3, float s0, float s1, float s2, float s3,
                                       const qs_f4& cw, const qs_f4& cc, const qs_f4& ce) {
  if ((QS_ABL & 64) && NEG) { qs_u12<INIT, false>(a0, a1, a2, a3, s0, s1, s2, s3, cw, cc, ce); return; }
  const float w0 = cw[0], w1 = cw[1], w2 = cw[2], w3 = cw[3], c0 = cc[0], c1 = cc[1], c2 = cc[2], c3 = cc[3], e0 = ce[0], e1 = ce[1],
              e2 = ce[2], e3 = ce[3];
  // operands: %0-3 acc, %4-7 src, %8-11 cw, %12-15 cc, %16-19 ce
#define QS_OPS_IN "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(e0), "v"(e1), "v"(e2), "v"(e3)
  if (!INIT && !NEG)
    asm volatile(
        "v_fmac_f32_e32 %0, %4, %12\n\tv_fmac_f32_e32 %1, %5, %13\n\tv_fmac_f32_e32 %2, %6, %14\n\tv_fmac_f32_e32 %3, %7, %15\n\t"
        "v_fmac_f32_e32 %1, %4, %9\n\tv_fmac_f32_e32 %2, %5, %10\n\tv_fmac_f32_e32 %3, %6, %11\n\t"
        "v_fmac_f32_e32 %0, %5, %16\n\tv_fmac_f32_e32 %1, %6, %17\n\tv_fmac_f32_e32 %2, %7, %18\n\t"
        QS_FD "%0, %7, %8" QS_DPPL QS_FD "%3, %4, %19" QS_DPPR
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
        : QS_OPS_IN
        : "memory");
  else if (!INIT && NEG)
    asm volatile(
        "v_fma_f32 %0, %4, -%12, %0\n\tv_fma_f32 %1, %5, -%13, %1\n\tv_fma_f32 %2, %6, -%14, %2\n\tv_fma_f32 %3, %7, -%15, %3\n\t"
        "v_fma_f32 %1, %4, -%9, %1\n\tv_fma_f32 %2, %5, -%10, %2\n\tv_fma_f32 %3, %6, -%11, %3\n\t"
        "v_fma_f32 %0, %5, -%16, %0\n\tv_fma_f32 %1, %6, -%17, %1\n\tv_fma_f32 %2, %7, -%18, %2\n\t"
        QS_FD "%0, %7, -%8" QS_DPPL QS_FD "%3, %4, -%19" QS_DPPR
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
        : QS_OPS_IN
        : "memory");
  else if (INIT && !NEG)
    asm volatile(
        "v_mul_f32_e32 %0, %4, %12\n\tv_mul_f32_e32 %1, %5, %13\n\tv_mul_f32_e32 %2, %6, %14\n\tv_mul_f32_e32 %3, %7, %15\n\t"
        "v_fmac_f32_e32 %1, %4, %9\n\tv_fmac_f32_e32 %2, %5, %10\n\tv_fmac_f32_e32 %3, %6, %11\n\t"
        "v_fmac_f32_e32 %0, %5, %16\n\tv_fmac_f32_e32 %1, %6, %17\n\tv_fmac_f32_e32 %2, %7, %18\n\t"
        QS_FD "%0, %7, %8" QS_DPPL QS_FD "%3, %4, %19" QS_DPPR
        : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3)
        : QS_OPS_IN
        : "memory");
  else
    asm volatile(
        "v_mul_f32_e64 %0, %4, -%12\n\tv_mul_f32_e64 %1, %5, -%13\n\tv_mul_f32_e64 %2, %6, -%14\n\tv_mul_f32_e64 %3, %7, -%15\n\t"
        "v_fma_f32 %1, %4, -%9, %1\n\tv_fma_f32 %2, %5, -%10, %2\n\tv_fma_f32 %3, %6, -%11, %3\n\t"
        "v_fma_f32 %0, %5, -%16, %0\n\tv_fma_f32 %1, %6, -%17, %1\n\tv_fma_f32 %2, %7, -%18, %2\n\t"
        QS_FD "%0, %7, -%8" QS_DPPL QS_FD "%3, %4, -%19" QS_DPPR
        : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3)
        : QS_OPS_IN
        : "memory");
#undef QS_OPS_IN
}
// unit e (a constant after unrolling) of acc (+)= stencil row of src
template <bool INIT, bool NEG>
__device__ __forceinline__ void qs_unit(QRow& acc, const QRow& src, int e, const qs_f4& cw, const qs_f4& cc, const qs_f4& ce) {
  if ((QS_ABL & 4) && !INIT) {
    asm volatile("" : "+v"(acc.t[0]), "+v"(acc.t[1]), "+v"(acc.t[2]), "+v"(acc.t[3]) : "v"(src.t[0]), "v"(src.t[3]), "v"(cw), "v"(cc), "v"(ce) : "memory");
    return;
  }
#define QS_UE(E)                                                                                                          \
  {                                                                                                                       \
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;                                                                         \
    if (!INIT) { a0 = acc.t[0][E]; a1 = acc.t[1][E]; a2 = acc.t[2][E]; a3 = acc.t[3][E]; }                                \
    qs_u12<INIT, NEG>(a0, a1, a2, a3, src.t[0][E], src.t[1][E], src.t[2][E], src.t[3][E], cw, cc, ce);                    \
    acc.t[0][E] = a0; acc.t[1][E] = a1; acc.t[2][E] = a2; acc.t[3][E] = a3;                                               \
  }
  if (e == 0) QS_UE(0)
  else if (e == 1) QS_UE(1)
  else if (e == 2) QS_UE(2)
  else QS_UE(3)
#undef QS_UE
}

// (Packed multiply-adds for the ten plain terms of two channel registers at once -- v_pk_fma_f32 with the by-tile coefficient
// vectors as op_sel halves, natural in this layout -- were built and timed on the probe: 3.50 us per step against 3.40.
// As in round 3 and as MI355X_MICROARCH.md says of packed f32 beside MFMAs: slower.  Not kept.)
#define QS_UPR 4  /* units per source row */
#define QS_UNIT qs_unit

// The values of L~ of the lane's four pixels in one row, as the ring holds them (per tile t: directions 0..3 = W NW N NE,
// directions 4..7 = E SE S SW; the diagonal and W once more as [tile] vectors), and the three coefficient vectors
// (west, centre, east by tile) of a source row:  y-1: SW S SE;  y: W diag E;  y+1: NW N NE.
struct QCoefLo {  // what the rows y-1 and y of a level need: six directions, each a vector by tile
  qs_f4 sw, s, se, w, dg, e;
};
struct QCoefHi {  // what the row y+1 needs
  qs_f4 nw, n, ne;
};

// multiplier of L~ in level j and the sign kept with the planes: as in the strip kernel (sp_mult / sp_wsign)
__host__ __device__ constexpr float qs_wsign(bool cheb, int j) { return cheb && ((j & 3) >= 2) ? -1.f : 1.f; }

template <bool CHEB, bool F16>
__global__ __launch_bounds__(QS_THREADS, 2) void cheb_qstrip5_kernel(QStripArgs a) {
  constexpr int K = 5, D = QS_D, RING = K + 2;
  // H1: level 1's matrix work (z_1 = x W_1) runs on the H wave, which otherwise reaches the step's barrier a fifth of a step
  // before its L partner (stamps in profiles/r5_qstrip_probe.txt): H adds z_1 to the dying b3 row it hands over, from which L
  // starts b1 anyway (b1 = z_1 + 2 L~ b2 - b3), with L's own weights of that level (same sign convention).  To make room for them
  // H asks for the next row of x behind slot s1 instead of at the top of the step.  3.36 us per step against 3.47 (probe, same
  // lease; half of z_1: 3.44).  Chebyshev basis only: the monomial recurrence hands no b3 row over (with z_1 alone in that slot of
  // the hand-over the monomial kernel gained 0.4 %: 3.32 against 3.33 -- its L wave has less to do to begin with; not kept).
#ifdef DSPH_QS_NOH1
  constexpr bool H1 = false;
#else
  constexpr bool H1 = CHEB;
#endif
  constexpr int ROWB = 2 * 2 * 4 * QS_FRAG;     // 16 KiB: one ring row of x ([32-channel block][hi | lo][tile] fragments)
  constexpr int RINGB = RING * ROWB;            // 112 KiB
  constexpr int HAND1 = 2 * 4 * QS_FRAG;        // 8 KiB per quarter: [b2 row | b3 row][tile]
  constexpr int HANDB = 4 * HAND1;              // 32 KiB
  constexpr int CROWB = 2304;                   // one ring row of L~: [9: the diagonal, directions 0..7][p][tile] floats
  constexpr int CRING = K + 1;
  constexpr int CRINGB = CRING * CROWB;         // 13,824 B
  constexpr int LDS_HAND = RINGB, LDS_C = RINGB + HANDB, LDS_FLAG = LDS_C + CRINGB;
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_FLAG + 64];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool roleL = wave >= 4;
  const int oq = wave & 3;
  const int p = lane & 15, q4 = lane >> 4;
  const unsigned lane16 = (unsigned)lane * 16u;
  const unsigned hand = (unsigned)LDS_HAND + (unsigned)oq * HAND1 + lane16;
  const unsigned flag_addr = (unsigned)LDS_FLAG + 4u * (unsigned)oq;
  auto flag_set = [&](int v) __attribute__((always_inline)) {
    asm volatile("ds_write_b32 %0, %1" : : "v"(flag_addr), "v"(v) : "memory");
  };
  auto flag_get = [&]() __attribute__((always_inline)) -> int {
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(flag_addr) : "memory");
    return v;
  };
  for (int i = tid; i < (LDS_FLAG + 64) / 16; i += QS_THREADS) reinterpret_cast<qs_f4*>(smem)[i] = qs_f4{0.f, 0.f, 0.f, 0.f};

  // this workgroup's piece of the tape and its maps: the workgroups of one XCD (blockIdx & 7) are neighbours in `ord`
  const int G = gridDim.x, ord = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);  // (the host launches a multiple of 8)
  const int piece = ord / a.wg_per_piece, map0 = ord - piece * a.wg_per_piece;
  if (piece >= a.pieces) return;  // (before the first barrier: the whole workgroup leaves)
  const int64_t tape = (int64_t)a.prefix[a.nstrips];
  const int64_t tape_begin = tape * piece / a.pieces, tape_end = tape * (piece + 1) / a.pieces;
  // the piece that starts at tape row r: strip, map, first row and length (wave-uniform arithmetic)
  auto locate = [&](int64_t r, int64_t r_end, QStrip& st) __attribute__((always_inline)) -> int {
    int lo = 0, hi = a.nstrips;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if ((int64_t)a.prefix[mid] <= r) lo = mid; else hi = mid;
    }
    {  // (the record through scalar registers: every field is wave-uniform, and the compiler should know -- rows, clamps and the
       // table look-ups' branches then run on the scalar unit)
      const QStrip g = a.strips[lo];
#define QS_U(f) st.f = __builtin_amdgcn_readfirstlane(g.f)
      QS_U(x0); QS_U(w); QS_U(xs); QS_U(y0); QS_U(y1); QS_U(xlo); QS_U(xhi); QS_U(ylo); QS_U(yhi); QS_U(tab); QS_U(tws);
#undef QS_U
    }
    const int h = st.y1 - st.y0;
    const int off = (int)(r - (int64_t)a.prefix[lo]);
    const int len = (int)(((int64_t)(h - off) < r_end - r) ? (int64_t)(h - off) : r_end - r);
    st.y0 += off;
    st.y1 = st.y0 + len;
    return len;
  };
  const unsigned xrowb = (unsigned)a.Fin * 4u, yrowb = (unsigned)a.ld * 4u;

  auto step_barrier = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

  // ---- x: every wave fetches half a tile of row ytop+1 at the top of a step and files it at the end -------------------
  // wave w: tile xt = w & 3, pixels p8 + 8 (w >> 2) of it; lane: half h = lane & 1, pixel p8 = (lane >> 1) & 7,
  // qh = (lane >> 4) & 1, kk = lane >> 5; instruction i = 0, 1: the 16 bytes at offset (2 i + kk) 64 + (2 qh + h) 16 of the
  // pixel's 256: channels 32 i + 16 kk + 8 qh + 4 h ..+3 -> fragment (block i, tile xt), lane slot (pixel, 2 kk + qh), half h.
  // Who fetches x: the four H waves a tile each (X_BY_H), or all eight waves half a tile each.  With level 1's matrix work on H
  // (H1) the H wave no longer has the time to spare: all eight is 2.4 % faster there (3.28 against 3.36 us per step on the
  // probe); without H1 (monomial basis) the H waves alone are 0.6 % faster.  (-DDSPH_QS_XALL / -DDSPH_QS_XBYH: tuning)
#if defined(DSPH_QS_XALL)
  constexpr bool X_BY_H = false;
#elif defined(DSPH_QS_XBYH)
  constexpr bool X_BY_H = true;
#else
  constexpr bool X_BY_H = !H1;
#endif
  constexpr int XN = X_BY_H ? 4 : 2;  // 16-byte loads per lane and row
  // X_BY_H: the H wave of quarter q fetches tile q: lane: half h = lane & 1, pixel (lane >> 1) & 15, qh = lane >> 5;
  //   instruction i = 0..3: the 16 bytes at offset 64 i + (2 qh + h) 16 of the pixel's 256: channels 16 i + 8 qh + 4 h ..+3
  //   -> fragment (block i >> 1, tile), lane slot (pixel, 2 (i & 1) + qh), half h.
  // all waves: wave w: tile w & 3, pixels p8 + 8 (w >> 2); lane: h = lane & 1, p8 = (lane >> 1) & 7, qh = (lane >> 4) & 1,
  //   kk = lane >> 5; instruction i = 0, 1: offset (2 i + kk) 64 + (2 qh + h) 16 -> fragment (block i, tile), slot (pixel, 2 kk + qh).
  const int xt = wave & 3, xpix = X_BY_H ? ((lane >> 1) & 15) : (((lane >> 1) & 7) + 8 * (wave >> 2));
  const unsigned x_goff = X_BY_H ? (unsigned)(2 * (lane >> 5) + (lane & 1)) * 16u
                                 : (unsigned)(lane >> 5) * 64u + (unsigned)(2 * ((lane >> 4) & 1) + (lane & 1)) * 16u;
  const unsigned x_loff = (unsigned)xt * QS_FRAG + (unsigned)(lane & 1) * 8u +
                          (X_BY_H ? (unsigned)(xpix + 16 * (lane >> 5)) * 16u : (unsigned)(xpix + 16 * (2 * (lane >> 5) + ((lane >> 4) & 1))) * 16u);
  // Where a pixel of the strip's plane lives: row = tab[(y >> 4) tws + (x >> 4)] + morton(x & 15, y & 15) -- the rectangle's table
  // of tile bases (cheb_fused.hip, build_qtstrips: a rectangle may cross base-pixel borders that continue the pixel grid by a
  // translation; inside a base pixel the table is the Morton plane itself).  A strip's 64 columns lie in at most five tile
  // columns, the row y is wave-uniform: the five bases of a tile row come by SCALAR loads (they share no counter with the
  // vector memory: a vector load here would wait for the previous step's y stores to drain -- measured, 12 % of the forward),
  // a lane keeps the one of its column (ci = its tile column - the strip's first) -- looked up when a row enters a new tile
  // row, every sixteenth step.
  typedef int qs_i4 __attribute__((ext_vector_type(4)));
  auto tab_lane = [&](const QStrip& st, unsigned ci, int yrow) __attribute__((always_inline)) -> unsigned {
    const int yc = min(max(yrow, st.ylo), st.yhi);
    const int32_t* trow = a.tab + __builtin_amdgcn_readfirstlane(st.tab + (yc >> 4) * st.tws + (max(st.xs, st.xlo) >> 4));
    qs_i4 b;
    int b4;
    asm volatile("s_load_dwordx4 %0, %2, 0x0\n\ts_load_dword %1, %2, 0x10\n\ts_waitcnt lgkmcnt(0)" : "=&s"(b), "=&s"(b4) : "s"(trow) : "memory");
    return (unsigned)(ci == 0 ? b[0] : ci == 1 ? b[1] : ci == 2 ? b[2] : ci == 3 ? b[3] : b4);
  };
  // A row's Morton bits inside its tile, spread(y & 15) << 1, are kept as wave-uniform state and stepped with the row:
  // (m | 0x55) + 1 & 0xaa counts in the odd bits and wraps to zero where a row enters a new tile row -- the moment for the
  // look-up (three scalar instructions per row and step; clamping y, spreading its bits and three range tests per row took
  // fifty, and at ~ 4 cycles of a wave's issue each that was 2 % of the forward).  Rows need no clamp to the strip's halo:
  // the table has a ring of one tile around the rectangle, the steps of a run stay inside it, and what lies beyond the halo
  // feeds no output.
  auto my_of = [&](int yrow) __attribute__((always_inline)) -> unsigned { return st_spread((unsigned)yrow & 15u) << 1; };
  auto my_next = [&](unsigned m) __attribute__((always_inline)) -> unsigned { return ((m | 0x55u) + 1u) & 0xaau; };
  auto row_in = [&](const QStrip& st, unsigned base, unsigned mX, int yrow) __attribute__((always_inline)) -> unsigned {
    const int yc = min(max(yrow, st.ylo), st.yhi);
    return base + (mX | (st_spread((unsigned)yc & 15u) << 1));
  };
  auto xfetch = [&](const char* xmap, unsigned xrow, qs_f4 (&xv)[XN]) __attribute__((always_inline)) {
    if (QS_ABL & 8) {
#pragma unroll
      for (int i = 0; i < XN; ++i) xv[i] = qs_f4{0.f, 0.f, 0.f, 0.f};
      return;
    }
    // (plain loads: the compiler knows the data is in flight, keeps the registers out of other use and counts vmcnt itself;
    // the asm statements around them keep the requests where they are written -- see cheb_qwgrad_kernel.h for what loads
    // inside asm statements did there)
    const char* src = xmap + (size_t)(xrow * xrowb + x_goff);
#pragma unroll
    for (int i = 0; i < XN; ++i) xv[i] = *reinterpret_cast<const qs_f4*>(src + (X_BY_H ? 64 : 128) * i);
  };
  // the point where the row is needed: the compiler waits for it here, not earlier (everything below depends on this statement)
  auto xw_wait = [&](qs_f4 (&xv)[XN]) __attribute__((always_inline)) {
    if (X_BY_H) asm volatile("" : "+v"(xv[0]), "+v"(xv[1]), "+v"(xv[XN - 2]), "+v"(xv[XN - 1]) : : "memory");
    else asm volatile("" : "+v"(xv[0]), "+v"(xv[1]) : : "memory");
  };
  auto xstore = [&](int slot, const qs_f4 (&xv)[XN]) __attribute__((always_inline)) {
    if (QS_ABL & 4096) {
#pragma unroll
      for (int i = 0; i < XN; ++i) asm volatile("" : : "v"(xv[i]));
      return;
    }
    unsigned char* q = smem + (unsigned)slot * ROWB + x_loff;
#pragma unroll
    for (int i = 0; i < XN; ++i) {
      // fragment of instruction i: block and, with X_BY_H, the upper / lower half of the block's lane slots
      const unsigned fo = X_BY_H ? (unsigned)(i >> 1) * (2 * 4 * QS_FRAG) + (unsigned)(i & 1) * (32u * 16u) : (unsigned)i * (2 * 4 * QS_FRAG);
      typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
      typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      qs_u2 hi, lo;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        // (f16: times the caller's power of two -- an f16 pair keeps 22 bits only where hi AND lo are normal numbers, i.e. for
        // |x xsc| from 2^-3 up; the store divides it out again, all exact)
        const float a0 = F16 ? xv[i][2 * j] * a.xsc : xv[i][2 * j], a1 = F16 ? xv[i][2 * j + 1] * a.xsc : xv[i][2 * j + 1];
        if (F16) {  // (a value beyond the f16 range becomes an infinity here and a NaN row in y: loud, not wrong)
          const f16x2 h = __builtin_convertvector(f32x2{a0, a1}, f16x2);
          const f32x2 hf = __builtin_convertvector(h, f32x2);
          const f16x2 l = __builtin_convertvector(f32x2{a0 - hf[0], a1 - hf[1]}, f16x2);
          hi[j] = __builtin_bit_cast(unsigned, h);
          lo[j] = __builtin_bit_cast(unsigned, l);
        } else {
          const bf16x2 h = __builtin_convertvector(f32x2{a0, a1}, bf16x2);
          const unsigned hu = __builtin_bit_cast(unsigned, h);
          const float h0 = __builtin_bit_cast(float, hu << 16), h1 = __builtin_bit_cast(float, hu & 0xffff0000u);
          const bf16x2 l = __builtin_convertvector(f32x2{a0 - h0, a1 - h1}, bf16x2);
          hi[j] = hu;
          lo[j] = __builtin_bit_cast(unsigned, l);
        }
      }
      *reinterpret_cast<qs_u2*>(q + fo) = hi;
      *reinterpret_cast<qs_u2*>(q + fo + 4 * QS_FRAG) = lo;
    }
  };
  // ---- L~: the H wave of quarter oq fetches the row's values of the pixels 4 p + oq (lanes q4 = 0: directions 0..3,
  // q4 = 1: directions 4..7, every lane the diagonal) and files them, doubled (Chebyshev), in the ring -----------------
  auto cfetch = [&](unsigned rid, qs_f4& cv, float& cd) __attribute__((always_inline)) {
    const char* pv = reinterpret_cast<const char*>(a.gvals8) + (size_t)rid * 32u + (unsigned)(q4 & 1) * 16u;
    const char* pd = reinterpret_cast<const char*>(a.gdiag) + (size_t)rid * 4u;
    cv = *reinterpret_cast<const qs_f4*>(pv);
    cd = *reinterpret_cast<const float*>(pd);
  };
  auto cw_wait = [&](qs_f4& cv, float& cd) __attribute__((always_inline)) {
    asm volatile("" : "+v"(cv), "+v"(cd) : : "memory");
  };
  // ring row: vector v = 0 the diagonal, v = 1 + d direction d (W NW N NE E SE S SW), each [p][tile] -- what a lane reads is
  // the 16 bytes of its four pixels of one direction
  auto cstore = [&](int slot, qs_f4 cv, float cd) __attribute__((always_inline)) {
    if (CHEB) { cv = cv + cv; cd = cd + cd; }
    unsigned char* q = smem + LDS_C + (unsigned)slot * CROWB + (unsigned)p * 16u + (unsigned)oq * 4u;
    if (q4 < 2) {
#pragma unroll
      for (int d = 0; d < 4; ++d) *reinterpret_cast<float*>(q + (unsigned)(1 + 4 * q4 + d) * 256u) = cv[d];
    }
    if (q4 == 2) *reinterpret_cast<float*>(q) = cd;
  };
  auto cvec = [&](const unsigned char* q, int v) __attribute__((always_inline)) -> qs_f4 {
    if (QS_ABL & 1024) { qs_f4 c = qs_f4{0.1f, 0.1f, 0.1f, 0.1f}; asm volatile("" : "+v"(c)); return c; }
    return *reinterpret_cast<const qs_f4*>(q + (unsigned)v * 256u);
  };
  auto clo_read = [&](int slot) __attribute__((always_inline)) -> QCoefLo {
    const unsigned char* q = smem + LDS_C + (unsigned)slot * CROWB + (unsigned)p * 16u;
    QCoefLo c;
    c.sw = cvec(q, 8); c.s = cvec(q, 7); c.se = cvec(q, 6);
    c.w = cvec(q, 1); c.dg = cvec(q, 0); c.e = cvec(q, 5);
    return c;
  };
  auto chi_read = [&](int slot) __attribute__((always_inline)) -> QCoefHi {
    const unsigned char* q = smem + LDS_C + (unsigned)slot * CROWB + (unsigned)p * 16u;
    QCoefHi c;
    c.nw = cvec(q, 2); c.n = cvec(q, 3); c.ne = cvec(q, 4);
    return c;
  };
// coefficient vectors of the three source rows
#define QS_LO0(c) (c).sw, (c).s, (c).se
#define QS_LO1(c) (c).w, (c).dg, (c).e
#define QS_HI(c) (c).nw, (c).n, (c).ne

  // The MFMA chain of one level (24 MFMAs: 2 channel blocks x 2 pairs of tiles x 3 terms x 2 tiles) beside NU stencil units:
  // MFMA m is followed by the units that fall to it.  Consecutive MFMAs go to different tiles; the fragments of the next
  // (block, pair) are requested when the current one starts.  Order of the three terms: W_hi.x_lo, W_lo.x_hi, W_hi.x_hi.
#if QS_ABL & 512
#define QS_FRLOAD(P) (wr[0][0][0])
#else
#define QS_FRLOAD(P) (*reinterpret_cast<const qs_bf8*>(P))
#endif
#define QS_ORD_A(j) ((j) == 1 ? 1 : 0)
#define QS_ORD_B(j) ((j) == 0 ? 1 : 0)
#define QS_FR0(FADDR)                                                                                                     \
  _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                                         \
    fr[0][u][0] = QS_FRLOAD(smem + (FADDR) + (unsigned)u * QS_FRAG);                                                      \
    fr[0][u][1] = QS_FRLOAD(smem + (FADDR) + (unsigned)(4 + u) * QS_FRAG);                                                \
  }
// (fr[0] holds set 0 on entry -- QS_FR0, issued in the tail of the chain before or at the top of the step; the units fall to
// MFMAs 0 .. 17; TAIL runs behind MFMA 17, when the units' coefficients and fr[0] are dead: the next slot's requests go out
// under the last six MFMAs)
#define QS_SETS(S0, S1, ROW, ZERO, WLEV, FADDR, NU, ...)                                                                  \
  _Pragma("unroll") for (int s = (S0); s < (S1); ++s) { /* set s = (block kb = s >> 1, pair tp = s & 1) */                \
    if (s + 1 < 4) {                                                                                                      \
      _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                                     \
        const unsigned fo = (unsigned)(((s + 1) >> 1) * 8 + ((s + 1) & 1) * 2 + u) * QS_FRAG;                             \
        fr[(s + 1) & 1][u][0] = QS_FRLOAD(smem + (FADDR) + fo);                                                           \
        fr[(s + 1) & 1][u][1] = QS_FRLOAD(smem + (FADDR) + fo + 4 * QS_FRAG);                                             \
      }                                                                                                                   \
    }                                                                                                                     \
    _Pragma("unroll") for (int j = 0; j < 3; ++j) {                                                                       \
      _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                                     \
        const int m = (s * 3 + j) * 2 + u;                                                                                \
        const int tt = (s & 1) * 2 + u;                                                                                   \
        const qs_bf8& wa_ = wr[WLEV][s >> 1][QS_ORD_A(j)];                                                                \
        const qs_bf8& bb_ = fr[s & 1][u][QS_ORD_B(j)];                                                                    \
        if ((ZERO) && s < 2 && j == 0) qs_m0<F16>((ROW).t[tt], wa_, bb_);                                                 \
        else qs_m<F16>((ROW).t[tt], wa_, bb_);                                                                            \
        _Pragma("unroll") for (int qq = (m * (NU)) / 18; qq < ((m + 1) * (NU)) / 18 && m < 18; ++qq) { __VA_ARGS__; }     \
      }                                                                                                                   \
    }                                                                                                                     \
  }
#define QS_CHAIN(ROW, ZERO, WLEV, FADDR, NU, TAIL, ...)                                                                   \
  {                                                                                                                       \
    QS_SETS(0, 3, ROW, ZERO, WLEV, FADDR, NU, __VA_ARGS__)                                                                \
    TAIL;                                                                                                                 \
    QS_SETS(3, 4, ROW, ZERO, WLEV, FADDR, NU, __VA_ARGS__)                                                                \
  }

  __syncthreads();

  if (!roleL) {
    // =================================================================================================================
    // H: levels 4, 3, 2.  R[0] = b4 rows, R[1] = b3 rows; logical row s of a plane in phase PH = R[.][(s + PH) % 3].
    // =================================================================================================================
    qs_bf8 wr[4][2][2];  // (wr[3]: level 1, from L's part of the image -- H1)
    if (H1) {
      const unsigned char* wp1 = a.wimg + ((size_t)(1 * 4 + oq) * 3) * (2 * 2 * QS_FRAG) + lane16;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int h = 0; h < 2; ++h) wr[3][kb][h] = *reinterpret_cast<const qs_bf8*>(wp1 + (size_t)(kb * 2 + h) * QS_FRAG);
    }
    {
      const unsigned char* wp = a.wimg + ((size_t)(0 * 4 + oq) * 3) * (2 * 2 * QS_FRAG) + lane16;
#pragma unroll
      for (int l = 0; l < 3; ++l)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int h = 0; h < 2; ++h) wr[l][kb][h] = *reinterpret_cast<const qs_bf8*>(wp + ((size_t)(l * 2 + kb) * 2 + h) * QS_FRAG);
    }
    int handed = 0;
    for (int64_t tr = tape_begin; tr < tape_end;) {
      QStrip st;
      tr += locate(tr, tape_end, st);
      for (int nq = map0; nq < a.N; nq += a.wg_per_piece) {
      const int Xc = min(max(st.xs + 4 * p + oq, st.xlo), st.xhi);         // L~: pixel 4 p + oq
      const int Xf = min(max(st.xs + 4 * xpix + xt, st.xlo), st.xhi);      // x: pixel 4 xpix + xt
      const int tc0 = max(st.xs, st.xlo) >> 4;  // the strip's first tile column
      const unsigned ciC = (unsigned)((Xc >> 4) - tc0), mXc = st_spread((unsigned)Xc & 15u);
      const unsigned ciF = (unsigned)((Xf >> 4) - tc0), mXf = st_spread((unsigned)Xf & 15u);
      const int T3 = ((st.y1 - st.y0) + 2 * D + 1 + 3) / 3;
      const char* __restrict__ xmap = reinterpret_cast<const char*>(a.x) + (size_t)nq * a.x_rows * xrowb;
      QRow R[2][3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
          for (int t = 0; t < 4; ++t) R[i][s].t[t] = qs_f4{0.f, 0.f, 0.f, 0.f};
      int ytop = st.y0 - D, slot_top = 0, cs_top = 0;
      step_barrier();  // (the previous item's last reads of the rings)
      {
        qs_f4 cv, xv[XN];
        float cd;
        cfetch(row_in(st, tab_lane(st, ciC, ytop - 1), mXc, ytop - 1), cv, cd);
        xfetch(xmap, row_in(st, tab_lane(st, ciF, ytop), mXf, ytop), xv);
        cw_wait(cv, cd);
        cstore(CRING - 1, cv, cd);
        xw_wait(xv);
        xstore(0, xv);
      }
      // the lane's tile bases of the rows the steps ask for: x of row ytop + 1, L~ of row ytop
      unsigned bXf = tab_lane(st, ciF, ytop + 1), bXc = tab_lane(st, ciC, ytop);
      unsigned myX = my_of(ytop + 1), myC = my_of(ytop);
      step_barrier();
      auto step = [&](auto ph_c) __attribute__((always_inline)) {
        constexpr int PH = decltype(ph_c)::value;
        constexpr int L0 = PH % 3, L1 = (PH + 1) % 3, L2 = (PH + 2) % 3;
        int snew = slot_top + 1;
        snew = snew == RING ? 0 : snew;
        auto slot_ix = [&](int back) __attribute__((always_inline)) -> int {
          int s = slot_top - back;
          s += s < 0 ? RING : 0;
          return s;
        };
        auto cslot_ix = [&](int back) __attribute__((always_inline)) -> int {
          int s = cs_top - back;
          s += s < 0 ? CRING : 0;
          return s;
        };
        QS_STAMP_DECL
        QS_STAMP(0);
        qs_f4 xv[XN];
        constexpr bool XLATE = H1;  // (the row's registers are H1's weights' in the slots s0, s1: requested behind s1 instead)
        if (myX == 0) bXf = tab_lane(st, ciF, ytop + 1);
        if (myC == 0) bXc = tab_lane(st, ciC, ytop);
        if (!XLATE) xfetch(xmap, bXf + (mXf | myX), xv);
        const unsigned f0 = (unsigned)slot_top * ROWB + lane16, f1 = (unsigned)slot_ix(1) * ROWB + lane16, f2 = (unsigned)slot_ix(2) * ROWB + lane16;
        const unsigned f3 = (unsigned)slot_ix(3) * ROWB + lane16;
        constexpr bool N3 = CHEB;  // level 3 enters with -2 L~ (Chebyshev), level 2 with +2 L~
        qs_bf8 fr[2][2][2];  // [buffer][tile of the pair][hi | lo]: the B fragments of the MFMA chains, two sets in flight
        QCoefLo c3 = clo_read(cslot_ix(1));  // row ytop-1: level 3
        QS_FR0(f0)
        QCoefLo c2;
        QCoefHi c3h, c2h;
        qs_f4 cv;
        float cd;
        // s0: z_4 -> b4[new] | b3[new] = -+ (b4[-2], b4[-1])
        QS_CHAIN(R[0][L2], true, 0, f0, 2 * QS_UPR, { c2 = clo_read(cslot_ix(2)); QS_FR0(f1) },
                 { if (qq < QS_UPR) QS_UNIT<true, N3>(R[1][L2], R[0][L0], qq, QS_LO0(c3)); else QS_UNIT<false, N3>(R[1][L2], R[0][L1], qq - QS_UPR, QS_LO1(c3)); })
        qs_settle<9>(R[0][L2]);
        qs_settle<1>(R[1][L2]);
        QS_STAMP(1);
        // s1: z_3 -> b3[new] | b2[new] = b4[-2] + (b3[-2], b3[-1]), in place in R[0][L0]   (c2: row ytop-2, level 2)
        QS_CHAIN(R[1][L2], false, 1, f1, 2 * QS_UPR, { c3h = chi_read(cslot_ix(1)); QS_FR0(f2) },
                 { if (qq < QS_UPR) QS_UNIT<!CHEB, false>(R[0][L0], R[1][L0], qq, QS_LO0(c2)); else QS_UNIT<false, false>(R[0][L0], R[1][L1], qq - QS_UPR, QS_LO1(c2)); })
        qs_settle<9>(R[1][L2]);
        qs_settle<1>(R[0][L0]);
        if (XLATE) xfetch(xmap, bXf + (mXf | myX), xv);  // (two thirds of a step ahead of its use: still more than the memory's latency)
        QS_STAMP(2);
        // s2: z_2 -> b2[new] | b3[new] += b4[new]
        // (the row ytop of L~ is requested in the tail as well: its latency is H's to wait out, H reaches the barrier before L)
        QS_CHAIN(R[0][L0], false, 2, f2, QS_UPR, { c2h = chi_read(cslot_ix(2)); cfetch(bXc + (mXc | myC), cv, cd); if (H1) { QS_FR0(f3) } },
                 { QS_UNIT<false, N3>(R[1][L2], R[0][L2], qq, QS_HI(c3h)); })
        qs_settle<9>(R[0][L0]);
        qs_settle<1>(R[1][L2]);
        QS_STAMP(3);
        // s3: b2[new] += b3[new] | H1: z_1 onto the dying b3 row
        if (H1) {
          QS_CHAIN(R[1][L0], false, 3, f3, QS_UPR, {}, { QS_UNIT<false, false>(R[0][L0], R[1][L2], qq, QS_HI(c2h)); })
          qs_settle<9>(R[1][L0]);
        } else {
#pragma unroll
          for (int qq = 0; qq < QS_UPR; ++qq) QS_UNIT<false, false>(R[0][L0], R[1][L2], qq, QS_HI(c2h));
        }
        QS_STAMP(4);
        // hand-over: b2[new] and the dying row of b3 -- once L has taken the previous pair
        for (int spin = 0; flag_get() <= handed && spin < (1 << 22); ++spin) {}  // (bounded: a lost partner must not hang the device)
        if (!(QS_ABL & 2048)) {
          unsigned char* hp = smem + hand;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            *reinterpret_cast<qs_f4*>(hp + t * QS_FRAG) = R[0][L0].t[t];
            if (CHEB) *reinterpret_cast<qs_f4*>(hp + (4 + t) * QS_FRAG) = R[1][L0].t[t];
          }
        }
        ++handed;
        QS_STAMP(5);
        xw_wait(xv);  // (waits for the row of L~ as well: requests complete in order)
        cw_wait(cv, cd);
        QS_STAMP(6);
        cstore(cs_top, cv, cd);
        xstore(snew, xv);
        QS_STAMP(7);
        slot_top = snew;
        cs_top = cs_top + 1 == CRING ? 0 : cs_top + 1;
        ++ytop;
        myX = my_next(myX);
        myC = my_next(myC);
        step_barrier();
        QS_STAMP(8);
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
    // L: levels 1, 0.  R[0] = b2 rows (from H), R[1] = b1 rows (the new one starts as the b3 row from H), Y = 2 y
    // (Chebyshev: level 0 runs doubled -- its weights carry the 2, the b2 row is doubled into Y, the stencil uses the 2 L~
    // of the ring as it stands -- and the store halves it; monomial: Y = y).
    // =================================================================================================================
    qs_bf8 wr[2][2][2];
    {
      const unsigned char* wp = a.wimg + ((size_t)(1 * 4 + oq) * 3) * (2 * 2 * QS_FRAG) + lane16;
#pragma unroll
      for (int l = 0; l < 2; ++l)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int h = 0; h < 2; ++h) wr[l][kb][h] = *reinterpret_cast<const qs_bf8*>(wp + ((size_t)(l * 2 + kb) * 2 + h) * QS_FRAG);
    }
    const float floor_v = a.act == DSPH_ACT_RELU ? 0.f : -__builtin_huge_valf();
    // (the f16 image carries its weights times a power of two, qstrip_wprep_kernel: the store takes it out again)
    const float ysc = (CHEB ? 0.5f : 1.f) * (F16 ? *reinterpret_cast<const float*>(a.wimg + 2 * 4 * 3 * 2 * 2 * QS_FRAG) * a.xsc_inv : 1.f);
#ifdef DSPH_QS_LPRIO  // (tuning: the L waves are the younger ones of their SIMDs and lose the issue arbitration to their H partner)
#define QS_STR2(x) #x
#define QS_STR(x) QS_STR2(x)
    asm volatile("s_setprio " QS_STR(DSPH_QS_LPRIO));
#endif
    qs_f4 bv = qs_f4{0.f, 0.f, 0.f, 0.f};
    if (a.bias != nullptr) bv = *reinterpret_cast<const qs_f4*>(a.bias + 16 * oq + 4 * q4);
    int taken = 0;
    for (int64_t tr = tape_begin; tr < tape_end;) {
      QStrip st;
      tr += locate(tr, tape_end, st);
      for (int nq = map0; nq < a.N; nq += a.wg_per_piece) {
      const int Xf = min(max(st.xs + 4 * xpix + xt, st.xlo), st.xhi);
      const int tc0 = max(st.xs, st.xlo) >> 4;  // the strip's first tile column
      const unsigned ciF = (unsigned)((Xf >> 4) - tc0), mXf = st_spread((unsigned)Xf & 15u);
      // the lane's four output pixels 4 p + t are one aligned group of four inside one tile (xs is a multiple of four): their
      // Morton bits are those of the group's first pixel plus 0, 1, 4, 5 (only pixels of [x0, x0 + w) are stored: never clamped)
      const int Xg = st.xs + 4 * p;
      const unsigned ciY = (unsigned)((min(max(Xg, st.xlo), st.xhi) >> 4) - tc0), mXg = st_spread((unsigned)Xg & 12u);
      const int cfirst = st.x0 - st.xs, clast = cfirst + st.w;  // output columns of the strip: [cfirst, clast)
      const int T3 = ((st.y1 - st.y0) + 2 * D + 1 + 3) / 3;
      const char* __restrict__ xmap = reinterpret_cast<const char*>(a.x) + (size_t)nq * a.x_rows * xrowb;
      char* __restrict__ ymap = reinterpret_cast<char*>(a.y) + (size_t)nq * a.y_rows * yrowb;
      QRow R[2][3], Y;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
          for (int t = 0; t < 4; ++t) R[i][s].t[t] = qs_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 4; ++t) Y.t[t] = qs_f4{0.f, 0.f, 0.f, 0.f};
      int ytop = st.y0 - D, slot_top = 0, cs_top = 0;
      step_barrier();  // (the previous item's last reads of the rings)
      if (!X_BY_H) {
        qs_f4 xv[XN];
        xfetch(xmap, row_in(st, tab_lane(st, ciF, ytop), mXf, ytop), xv);
        xw_wait(xv);
        xstore(0, xv);
      }
      // the lane's tile bases: x of row ytop + 1 (when all eight waves fetch x), y of row ytop - K
      unsigned bXf = X_BY_H ? 0u : tab_lane(st, ciF, ytop + 1), bY = tab_lane(st, ciY, st.y0);
      unsigned myX = my_of(ytop + 1), myY = my_of(ytop - K);
      step_barrier();
      auto step = [&](auto ph_c) __attribute__((always_inline)) {
        constexpr int PH = decltype(ph_c)::value;
        constexpr int L0 = PH % 3, L1 = (PH + 1) % 3, L2 = (PH + 2) % 3;
        int snew = slot_top + 1;
        snew = snew == RING ? 0 : snew;
        auto slot_ix = [&](int back) __attribute__((always_inline)) -> int {
          int s = slot_top - back;
          s += s < 0 ? RING : 0;
          return s;
        };
        auto cslot_ix = [&](int back) __attribute__((always_inline)) -> int {
          int s = cs_top - back;
          s += s < 0 ? CRING : 0;
          return s;
        };
        QS_STAMP_DECL
        QS_STAMP(0);
        qs_f4 xv[XN];
        if (!X_BY_H && myX == 0) bXf = tab_lane(st, ciF, ytop + 1);
        // (a row under the run's first one looks its tile row up too: the base is the right one by the time a row is stored --
        // the look-up of the last multiple of sixteen up to y0, or the one made before the steps)
        if (myY == 0) bY = tab_lane(st, ciY, ytop - K);
        if (!X_BY_H) xfetch(xmap, bXf + (mXf | myX), xv);
#ifdef DSPH_QS_LEARLY
        constexpr bool LE = CHEB && H1;  // (tuning: level 0's lower source rows first -- they need nothing H hands over)
#else
        constexpr bool LE = false;
#endif
        QCoefLo c0;
        QCoefHi c0h;
        if (LE) c0 = clo_read(cslot_ix(5));  // (requested before the hand-over rows: the first units then wait for these alone)
        // the rows H left at the end of the previous step: b2[new] -> R[0][L2] (the set that died then), b3 -> R[1][L2]
        if (!(QS_ABL & 2048)) {
          const unsigned char* hp = smem + hand;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            R[0][L2].t[t] = *reinterpret_cast<const qs_f4*>(hp + t * QS_FRAG);
            if (CHEB) R[1][L2].t[t] = *reinterpret_cast<const qs_f4*>(hp + (4 + t) * QS_FRAG);
          }
        }
        ++taken;
        flag_set(taken);  // (LDS operations of a wave complete in order: the reads above are done first)
        const unsigned f1 = (unsigned)slot_ix(4) * ROWB + lane16, f0 = (unsigned)slot_ix(5) * ROWB + lane16;
        constexpr bool N1 = CHEB;  // level 1 enters with -2 L~, level 0 with +2 L~ into the doubled Y
        QS_STAMP(1);
        qs_bf8 fr[2][2][2];
        if (LE) {
#pragma unroll
          for (int t = 0; t < 4; ++t) Y.t[t] = R[0][L0].t[t] + R[0][L0].t[t];
          qs_settle<1>(Y);
#pragma unroll
          for (int qq = 0; qq < 2 * QS_UPR; ++qq) {
            if (qq < QS_UPR) QS_UNIT<false, false>(Y, R[1][L0], qq, QS_LO0(c0)); else QS_UNIT<false, false>(Y, R[1][L1], qq - QS_UPR, QS_LO1(c0));
          }
          qs_settle<1>(Y);
        }
        const QCoefLo c1 = clo_read(cslot_ix(4));  // row ytop-4: level 1
        const QCoefHi c1h = chi_read(cslot_ix(4));
        QS_FR0(f0)
#ifdef DSPH_QS_LBARE
        // (tuning build: L's stencil units first and bare, its one chain of matrix instructions last -- away from H's first two chains)
        if (CHEB && H1) {
          c0 = clo_read(cslot_ix(5));
          c0h = chi_read(cslot_ix(5));
#pragma unroll
          for (int qq = 0; qq < 3 * QS_UPR; ++qq) {
            if (qq < QS_UPR) QS_UNIT<false, N1>(R[1][L2], R[0][L0], qq, QS_LO0(c1));
            else if (qq < 2 * QS_UPR) QS_UNIT<false, N1>(R[1][L2], R[0][L1], qq - QS_UPR, QS_LO1(c1));
            else QS_UNIT<false, N1>(R[1][L2], R[0][L2], qq - 2 * QS_UPR, QS_HI(c1h));
          }
#pragma unroll
          for (int t = 0; t < 4; ++t) Y.t[t] = R[0][L0].t[t] + R[0][L0].t[t];
          qs_settle<1>(Y);
#pragma unroll
          for (int qq = 0; qq < 2 * QS_UPR; ++qq) {
            if (qq < QS_UPR) QS_UNIT<false, false>(Y, R[1][L0], qq, QS_LO0(c0)); else QS_UNIT<false, false>(Y, R[1][L1], qq - QS_UPR, QS_LO1(c0));
          }
          qs_settle<1>(R[1][L2]);
#pragma unroll
          for (int qq = 0; qq < QS_UPR; ++qq) QS_UNIT<false, false>(Y, R[1][L2], qq, QS_HI(c0h));
          qs_settle<1>(Y);
          QS_STAMP(2);
          QS_CHAIN(Y, false, 1, f0, 0, {}, {})
          qs_settle<9>(Y);
          QS_STAMP(3);
        } else
#endif
        {
        if (CHEB && !LE) {
#pragma unroll
          for (int t = 0; t < 4; ++t) Y.t[t] = R[0][L0].t[t] + R[0][L0].t[t];
          qs_settle<1>(Y);
        }
        // s0: 2 z_0 (+ 2 b2[-1]) -> Y | b1[new] (= b3 row from H) -+= (b2[-1], b2[0], b2[+1])
        QS_CHAIN(Y, !CHEB, 1, f0, 3 * QS_UPR, { if (!LE) c0 = clo_read(cslot_ix(5)); else c0h = chi_read(cslot_ix(5)); if (!H1) { QS_FR0(f1) } },
                 { if (qq < QS_UPR) QS_UNIT<!CHEB, N1>(R[1][L2], R[0][L0], qq, QS_LO0(c1));
                   else if (qq < 2 * QS_UPR) QS_UNIT<false, N1>(R[1][L2], R[0][L1], qq - QS_UPR, QS_LO1(c1));
                   else QS_UNIT<false, N1>(R[1][L2], R[0][L2], qq - 2 * QS_UPR, QS_HI(c1h)); })
        qs_settle<9>(Y);
        qs_settle<1>(R[1][L2]);
        QS_STAMP(2);
        // s1: z_1 -> b1[new] | Y += (b1[-2], b1[-1])   (c0: row ytop-5, level 0)
        if (H1 && LE) {  // (the lower rows went first)
        } else if (H1) {  // (z_1 is in the row H handed over)
          c0h = chi_read(cslot_ix(5));
#pragma unroll
          for (int qq = 0; qq < 2 * QS_UPR; ++qq) {
            if (qq < QS_UPR) QS_UNIT<false, false>(Y, R[1][L0], qq, QS_LO0(c0)); else QS_UNIT<false, false>(Y, R[1][L1], qq - QS_UPR, QS_LO1(c0));
          }
        } else {
          QS_CHAIN(R[1][L2], false, 0, f1, 2 * QS_UPR, { c0h = chi_read(cslot_ix(5)); },
                   { if (qq < QS_UPR) QS_UNIT<false, false>(Y, R[1][L0], qq, QS_LO0(c0)); else QS_UNIT<false, false>(Y, R[1][L1], qq - QS_UPR, QS_LO1(c0)); })
        }
        qs_settle<9>(R[1][L2]);
        QS_STAMP(3);
        // s2: Y += b1[new]: y of row ytop - K
#pragma unroll
        for (int qq = 0; qq < QS_UPR; ++qq) QS_UNIT<false, false>(Y, R[1][L2], qq, QS_HI(c0h));
        }
        QS_STAMP(4);
        if (!X_BY_H) xw_wait(xv);  // (here, in front of this step's y stores: the wait is for everything in flight)
        QS_STAMP(5);
        {
          const int yr = ytop - K;
          const bool row_ok = yr >= st.y0 && yr < ((QS_ABL & 16) ? st.y0 + 1 : st.y1);
          const unsigned rowg = bY + (mXg | myY);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int c = 4 * p + t;
            if (row_ok && c >= cfirst && c < clast) {
              qs_f4 o;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float v = fmaf(Y.t[t][e], ysc, bv[e]);
                // (f16: an input beyond the f16 range has become a NaN by now -- it must reach y, not be floored away by the ReLU)
                o[e] = F16 ? (v < floor_v ? floor_v : v) : fmaxf(v, floor_v);
              }
              *reinterpret_cast<qs_f4*>(ymap + (size_t)(rowg + (unsigned)((t & 1) + 4 * (t >> 1))) * yrowb + (unsigned)(16 * oq + 4 * q4) * 4u) = o;
            }
          }
        }
        QS_STAMP(6);
        if (!X_BY_H) xstore(snew, xv);
        QS_STAMP(7);
        slot_top = snew;
        cs_top = cs_top + 1 == CRING ? 0 : cs_top + 1;
        ++ytop;
        myX = my_next(myX);
        myY = my_next(myY);
        step_barrier();
        QS_STAMP(8);
      };
      for (int t3 = 0; t3 < T3; ++t3) {
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{});
      }
      }
    }
  }
#undef QS_CHAIN
#undef QS_SETS
#undef QS_FR0
#undef QS_FRLOAD
}

}  // namespace dsph
