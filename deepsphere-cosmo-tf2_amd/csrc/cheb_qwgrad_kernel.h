// Quad-strip form of the weight gradient (round 5): dW[f K + k, o] = sum over maps and pixels of T_k(L~)x [., f] * dy[., o]
// (the gradient TF's autodiff derives from gnn_layers.py:131-150) for K = 5, 64 -> 64, on the strips of cheb_qstrip_kernel.h.
//
// The contraction runs over PIXELS, and a matrix instruction's inner index lives in a lane's registers while the stencil wants
// the pixels across the lanes (DPP) -- so every plane row has to cross the LDS once, from "lane = pixel group" to "lane =
// channel, registers = eight pixels".  What makes that fit the CU's 160 KiB is the product rule of the polynomials,
//     T_3 = 2 T_2 T_1 - T_1,   T_4 = 2 T_2 T_2 - T_0      (monomial basis: L^3 = L^2 L, L^4 = L^2 L^2)
// with a symmetric L~ (checked by the host; T_k(L~) is then symmetric too):
//     <T_3 x, dy> = 2 <T_2 x, T_1 dy> - <x, T_1 dy>,   <T_4 x, dy> = 2 <T_2 x, T_2 dy> - <x, dy>,   <T_1 x, dy> = <x, T_1 dy>
// i.e. the recurrence runs only to order TWO, on BOTH operands, and the five products
//     G00 = <P0,Q0>  G01 = <P0,Q1>  G20 = <P2,Q0>  G21 = <P2,Q1>  G22 = <P2,Q2>,     P_j = T_j x,  Q_j = T_j dy
// give dW_0 = G00, dW_1 = G01, dW_2 = G20, dW_3 = 2 G21 - G01, dW_4 = 2 G22 - G00 (monomial: the five as they are).  All the
// (the sums running over ALL pixels: for the share of a set of output pixels Omega -- a strip segment -- dy is zeroed outside
// Omega when it is loaded and the products run over Omega grown by two rings, which the strip's four halo columns and the
// run-in rows cover: sum_{Omega} (T_k x) dy = <T_k x, 1_Omega dy> exactly, whatever the neighbouring work items do).  All the
// planes of a row are ready in the same step (no ring of dy rows waiting for level 4: that ring was what did not fit), the
// vector work is the forward's (four stencil applications of 64 channels per row), and the two operands' recurrences are the
// SAME code: waves 0..3 run it on x, waves 4..7 on dy, a wave per 16 channels, no hand-over between roles.
//
// One workgroup = eight waves = one strip row per step (lane layout, strips, tape and work split: cheb_qstrip_kernel.h).
// Step with top row t:  S0[t] arrives (global -> registers, requested a step ahead), S1[t-1] = L~ S0, S2[t-2] = 2 L~ S1 - S0
//   (12 + 12 stencil units), while the matrix pipe contracts the row staged by the previous step; barrier; the planes of row
//   t-2 are split hi + lo into bf16 and staged; barrier.
// Staging (5 plane rows x 16,640 B): [plane][hi | lo][16-channel group][32-pixel block][k-group kg][chunk][8 pixels x 2 B],
//   the 16-byte chunk of channel m in k-group kg at chunk index m ^ 2 kg (256-byte k-groups, 1,040-byte blocks): the writers'
//   ds_write2_b64 (lane = (pixel group p, channel group q4)) and the readers' ds_read_b128 (lane = (channel m, k-group): an A or
//   B operand of v_mfma_f32_16x16x32_bf16 as it stands) are both free of bank conflicts -- see `rd_off` / `wr_a` below.  (The
//   first layout padded the k-groups to 272 bytes instead: a third of the LDS cycles were conflicts, profiles/r5_qwgrad_pmc.json.)
//   The inner index of a 32-pixel block is k = 4 (p & 7) + tile, the same on both operands.
// Matrix work: wave (og = w & 3, fh = w >> 2) owns the output tiles (pair, f-group 2 fh + {0, 1}, o-group og): ten 16 x 16
//   accumulators (40 registers) that live for the whole kernel; three terms per product (hi.hi + hi.lo + lo.hi).
// (Tried: P0, Q0, Q1 of a row exist when the step begins; staged among the step's matrix instructions into a second set of
//   buffers (154 KiB) they leave one plane per side behind the barrier -- 3.90 us per step against 3.81: the stores and the
//   fragment reads share the LDS, and the LDS is what the step waits for.  Not kept.)
// Output: one slab [5][64][64] per workgroup (of every second workgroup: the slab of -dy, see `sgn`); qwgrad_reduce_kernel adds
//   the slabs in a fixed order and applies the rule above.
#pragma once

#include "cheb_qstrip_kernel.h"

namespace dsph {

#ifdef DSPH_QW_PAD  // (tuning: the padded layout this kernel started with: 272-byte k-groups, 1,216-byte blocks, no swizzle)
constexpr bool QW_SWZ = false;
constexpr int QW_SK = 272, QW_SB = 1216;
#else
constexpr bool QW_SWZ = true;
constexpr int QW_SK = 256;               // bytes of one k-group: 16 channels x 16 B
constexpr int QW_SB = 1040;              // one 32-pixel block of a 16-channel group: 4 k-groups, + 16
#endif
constexpr int QW_CG = 2 * QW_SB;         // a 16-channel group: two blocks
constexpr int QW_HL = 4 * QW_CG;         // hi or lo of a plane row
constexpr int QW_PLANE = 2 * QW_HL;      // 19,456 B
constexpr int QW_P0 = 0, QW_P2 = 1, QW_Q0 = 2, QW_Q1 = 3, QW_Q2 = 4, QW_NPLANES = 5;
constexpr int QW_PAIRS = 5;              // G00 G01 G20 G21 G22
constexpr int QW_SLAB = QW_PAIRS * 64 * 64;  // floats per workgroup

struct QWgradArgs {
  const float* x;
  const float* dy;
  float* slabs;               // [gridDim.x][QW_SLAB]
  const float* gvals8;
  const float* gdiag;
  const QStrip* strips;
  const int32_t* tab;         // the rectangles' tables of tile bases (cheb_qstrip_kernel.h: QStrip::tab, ::tws)
  const int32_t* prefix;
  int64_t x_rows, dy_rows;    // rows per map
  int nstrips, N, lddy;       // lddy: row stride of dy in floats (x: 64)
  int pieces, wg_per_piece;
};

template <bool CHEB>
__global__ __launch_bounds__(QS_THREADS, 2) void cheb_qwgrad5_kernel(QWgradArgs a) {
  constexpr int D = QS_D;                 // rows of run-in on either side: two rings for the products' domain, two more for P2 on it
  constexpr int CROWB = 2304, CRING = 3;  // rows of L~ as in cheb_qstrip_kernel.h: [9][p][tile] floats
  constexpr int LDS_C = QW_NPLANES * QW_PLANE;
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_C + CRING * CROWB];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool sideQ = wave >= 4;  // waves 4..7: the recurrence on dy
  const int cq = wave & 3;       // the side's 16-channel quarter this wave runs the recurrence on
  const int og = wave & 3, fh = wave >> 2;  // matrix work: o-group, f-half
  const int p = lane & 15, q4 = lane >> 4;
  for (int i = tid; i < (int)sizeof(smem) / 16; i += QS_THREADS) reinterpret_cast<qs_f4*>(smem)[i] = qs_f4{0.f, 0.f, 0.f, 0.f};

  const int G = gridDim.x, ord = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int piece = ord / a.wg_per_piece, map0 = ord - piece * a.wg_per_piece;
  float* __restrict__ slab = a.slabs + (size_t)blockIdx.x * QW_SLAB;
  // Every second workgroup runs on -dy and hands in -G (qwgrad_reduce_kernel subtracts its slab).  Why: the matrix pipe adds
  // the 32 products of an instruction to an accumulator hundreds of times their size, and what it drops when it aligns them
  // is dropped towards minus infinity -- measured at the headline size (tests/diag_dw_by_order.py): every element of dW low by the
  // same 2e-5 of max |dW| whatever its sign, growing with the length of the sum (the same in the BFS-tile kernel's bf16 mode).
  // A bias that does not depend on the sign of the data cancels between a workgroup and its mirror.
  const float sgn = (sideQ && (ord & 1)) ? -1.f : 1.f;
  qs_f4 acc[QW_PAIRS][2];
#pragma unroll
  for (int i = 0; i < QW_PAIRS; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = qs_f4{0.f, 0.f, 0.f, 0.f};

  const int64_t tape = (int64_t)a.prefix[a.nstrips];
  const int64_t tape_begin = piece < a.pieces ? tape * piece / a.pieces : 0, tape_end = piece < a.pieces ? tape * (piece + 1) / a.pieces : 0;
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
  auto step_barrier = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

  const unsigned srowb = sideQ ? (unsigned)a.lddy * 4u : 256u;  // bytes of a pixel's row of this wave's operand
  const int64_t srows = sideQ ? a.dy_rows : a.x_rows;
  const char* sbase = reinterpret_cast<const char*>(sideQ ? a.dy : a.x);
  // staging: where this lane writes (per plane, hi | lo, channel cc: + cc 16) and reads (per plane, hi | lo, group, block)
  // Swizzle: the 16-byte chunk of channel m in k-group kg sits at chunk index m ^ 2 kg.  A reader's lane groups (ds_read_b128:
  // {0-3, 12-15, 20-27}, ...) then still cover sixteen different chunks of 256 bytes, and the four k-groups a store's lane group
  // touches (ds_write2_b64: eight lanes = four k-groups x two halves, two chunks each) fall on four different 32-byte blocks of
  // the 128 bytes a store's banks span: no conflicts on either side with contiguous 1 KiB fragments (counters of the padded
  // layout: a third of the LDS cycles were conflicts, profiles/r5_qwgrad_pmc.json).
  const int kgw = (p >> 1) & 3;  // k-group this lane writes
  const unsigned wr_com = (unsigned)cq * QW_CG + (unsigned)(p >> 3) * QW_SB + (unsigned)kgw * QW_SK + (unsigned)(p & 1) * 8u +
                          (unsigned)(4 * (QW_SWZ ? (q4 ^ (kgw >> 1)) : q4)) * 16u;
  // channels 4 q4 + {0, 1} go to wr_a (+ 0, + 16), channels 4 q4 + {2, 3} to wr_b: swapped in the odd k-groups (cc ^ 2 (kg & 1))
  const unsigned wr_a = wr_com + ((QW_SWZ && (kgw & 1)) ? 32u : 0u), wr_b = wr_com + ((QW_SWZ && (kgw & 1)) ? 0u : 32u);
  const unsigned rd_off = (unsigned)(lane >> 4) * QW_SK + (unsigned)((lane & 15) ^ (QW_SWZ ? 2 * (lane >> 4) : 0)) * 16u;

  // ---- L~ (waves 0..3): the row's values of the pixels 4 p + cq, filed as cheb_qstrip_kernel.h files them ---------------
  auto cfetch = [&](const char* gv, const char* gd, unsigned offv, unsigned offd, qs_f4& cv, float& cd) __attribute__((always_inline)) {
    cv = *reinterpret_cast<const qs_f4*>(gv + offv);
    cd = *reinterpret_cast<const float*>(gd + offd);
  };
  auto cstore = [&](int slot, qs_f4 cv, float cd) __attribute__((always_inline)) {
    unsigned char* q = smem + LDS_C + (unsigned)slot * CROWB + (unsigned)p * 16u + (unsigned)cq * 4u;
    if (q4 < 2) {
#pragma unroll
      for (int d = 0; d < 4; ++d) *reinterpret_cast<float*>(q + (unsigned)(1 + 4 * q4 + d) * 256u) = cv[d];
    }
    if (q4 == 2) *reinterpret_cast<float*>(q) = cd;
  };
  auto cvec = [&](int slot, int v) __attribute__((always_inline)) -> qs_f4 {
    return *reinterpret_cast<const qs_f4*>(smem + LDS_C + (unsigned)slot * CROWB + (unsigned)p * 16u + (unsigned)v * 256u);
  };
  struct C3 { qs_f4 w, c, e; };
  // coefficient vectors (west, centre, east by tile) of the source row y-1 (which = 0), y (1), y+1 (2)
  auto crow = [&](int slot, int which) __attribute__((always_inline)) -> C3 {
    C3 r;
    if (which == 0) { r.w = cvec(slot, 8); r.c = cvec(slot, 7); r.e = cvec(slot, 6); }
    else if (which == 1) { r.w = cvec(slot, 1); r.c = cvec(slot, 0); r.e = cvec(slot, 5); }
    else { r.w = cvec(slot, 2); r.c = cvec(slot, 3); r.e = cvec(slot, 4); }
    return r;
  };

  // one plane row of this wave, split hi + lo into bf16 and staged
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  auto stage = [&](int plane, const QRow& R) __attribute__((always_inline)) {
    if (QS_ABL & 8192) { asm volatile("" : : "v"(R.t[0]), "v"(R.t[1]), "v"(R.t[2]), "v"(R.t[3])); return; }  // (tuning builds: QS_ABL)
    unsigned char* qa = smem + (unsigned)plane * QW_PLANE + wr_a;
    unsigned char* qb = smem + (unsigned)plane * QW_PLANE + wr_b;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      unsigned char* q = (cc < 2 ? qa : qb) + (cc & 1) * 16;
      qs_u2 hi, lo;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float a0 = R.t[2 * j][cc], a1 = R.t[2 * j + 1][cc];
        const bf16x2 h = __builtin_convertvector(f32x2{a0, a1}, bf16x2);
        const unsigned hu = __builtin_bit_cast(unsigned, h);
        const float h0 = __builtin_bit_cast(float, hu << 16), h1 = __builtin_bit_cast(float, hu & 0xffff0000u);
        const bf16x2 l = __builtin_convertvector(f32x2{a0 - h0, a1 - h1}, bf16x2);
        hi[j] = hu;
        lo[j] = __builtin_bit_cast(unsigned, l);
      }
      *reinterpret_cast<qs_u2*>(q) = hi;
      *reinterpret_cast<qs_u2*>(q + QW_HL) = lo;
    }
  };
  auto frag = [&](int plane, int hl, int cg, int blk) __attribute__((always_inline)) -> qs_bf8 {
    if (QS_ABL & 16384) { qs_bf8 z; asm volatile("v_mov_b32 %0, 0" : "=v"(z)); return z; }
    return *reinterpret_cast<const qs_bf8*>(smem + (unsigned)plane * QW_PLANE + (unsigned)hl * QW_HL + (unsigned)cg * QW_CG + (unsigned)blk * QW_SB + rd_off);
  };

  __syncthreads();

  for (int64_t tr = tape_begin; tr < tape_end;) {
    QStrip st;
    tr += locate(tr, tape_end, st);
    for (int nq = map0; nq < a.N; nq += a.wg_per_piece) {
      // byte offsets of this lane's four pixels in a row of the map (+ its channels), of its pixel of L~
      unsigned voff[4];
      bool colk[4];
      const int cfirst = st.x0 - st.xs, clast = cfirst + st.w;
      // (rows through the rectangle's table of tile bases, as the forward kernel: row = tab[(y >> 4) tws + (x >> 4)] + morton(x & 15,
      // y & 15); the lane's four pixels are one aligned group of four inside one tile, or -- clamped -- one and the same pixel)
      const int xclo = sideQ ? st.x0 : st.xlo, xchi = sideQ ? st.x0 + st.w - 1 : st.xhi;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        // (dy is read on the output pixels only -- elsewhere it is zeroed, and its halo may not exist: the load stays inside)
        voff[t] = st_spread((unsigned)min(max(st.xs + 4 * p + t, xclo), xchi) & 15u) * srowb + (unsigned)(16 * cq + 4 * q4) * 4u;
        colk[t] = !sideQ || (4 * p + t >= cfirst && 4 * p + t < clast);
      }
      // (one register for the lane's tile columns, counted from the strip's first: the operand's pixels [0, 4), the pixel of L~
      // [4, 8), its Morton bits [8, 15); the five tile bases of a tile row come by scalar loads when a row enters a new tile row,
      // as in cheb_qstrip_kernel.h)
      const int Xc = min(max(st.xs + 4 * p + cq, st.xlo), st.xhi);
      const int tc0 = max(st.xs, st.xlo) >> 4;
      const unsigned pk = (unsigned)((min(max(st.xs + 4 * p, xclo), xchi) >> 4) - tc0) | ((unsigned)((Xc >> 4) - tc0) << 4) | (st_spread((unsigned)Xc & 15u) << 8);
      const char* smap = sbase + (size_t)nq * srows * srowb;
      typedef int qs_i4 __attribute__((ext_vector_type(4)));
      auto tab_lane = [&](unsigned ci, int yc) __attribute__((always_inline)) -> unsigned {
        const int32_t* trow = a.tab + __builtin_amdgcn_readfirstlane(st.tab + (yc >> 4) * st.tws + tc0);
        qs_i4 b;
        int b4;
        asm volatile("s_load_dwordx4 %0, %2, 0x0\n\ts_load_dword %1, %2, 0x10\n\ts_waitcnt lgkmcnt(0)" : "=&s"(b), "=&s"(b4) : "s"(trow) : "memory");
        return (unsigned)(ci == 0 ? b[0] : ci == 1 ? b[1] : ci == 2 ? b[2] : ci == 3 ? b[3] : b4);
      };
      const int rlo = sideQ ? st.y0 : st.ylo, rhi = sideQ ? st.y1 - 1 : st.yhi;
      // the lane's tile bases of the rows in flight: the operand's (row clamped to [rlo, rhi]) and L~'s (to [ylo, yhi])
      unsigned bV = tab_lane(pk & 15u, min(max(st.y0 - D, rlo), rhi)), bC = sideQ ? 0u : tab_lane((pk >> 4) & 15u, min(max(st.y0 - D, st.ylo), st.yhi));
      // Plain loads, not asm: the compiler then knows the data is in flight -- it keeps the registers out of other use until
      // the first reader and counts vmcnt itself.  (With the loads in asm statements it took their results for present and
      // moved them through registers it reused meanwhile: wrong rows.)  The asm statements around keep the requests in place.
      auto row_fetch = [&](int yrow, QRow& R) __attribute__((always_inline)) {
        // (32-bit offsets from the map's scalar base: a map is under 4 GiB -- strips_apply -- and the loads keep their scalar-base form)
        const int yc = min(max(yrow, rlo), rhi);
        if (yrow > rlo && yrow <= rhi && (yrow & 15) == 0) bV = tab_lane(pk & 15u, yc);
        const unsigned rb = (bV + (st_spread((unsigned)yc & 15u) << 1)) * srowb;
#pragma unroll
        for (int t = 0; t < 4; ++t) R.t[t] = *reinterpret_cast<const qs_f4*>(smap + (rb + voff[t]));
      };
      const int T3 = ((st.y1 - st.y0) + 2 * D + 1 + 2) / 3;
      QRow S0[3], S1[3];
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) { S0[s].t[t] = qs_f4{0.f, 0.f, 0.f, 0.f}; S1[s].t[t] = qs_f4{0.f, 0.f, 0.f, 0.f}; }
      int ytop = st.y0 - D, cs = 0;  // cs: ring slot of the row ytop of L~
      QRow nx;  // the row in flight
      row_fetch(ytop, nx);
      // the row of L~ travels like the row of the operand: requested at the end of a step, filed at the end of the next
      qs_f4 cv = qs_f4{0.f, 0.f, 0.f, 0.f};
      float cd = 0.f;
      auto coef_fetch = [&](int yrow) __attribute__((always_inline)) {
        const int yc = min(max(yrow, st.ylo), st.yhi);
        if (yrow > st.ylo && yrow <= st.yhi && (yrow & 15) == 0) bC = tab_lane((pk >> 4) & 15u, yc);
        const size_t rid = bC + ((pk >> 8) | (st_spread((unsigned)yc & 15u) << 1));
        cfetch(reinterpret_cast<const char*>(a.gvals8) + rid * 32u, reinterpret_cast<const char*>(a.gdiag) + rid * 4u, (unsigned)(q4 & 1) * 16u, 0u, cv, cd);
      };
      if (!sideQ) coef_fetch(ytop);

      auto step = [&](auto ph_c) __attribute__((always_inline)) {
        constexpr int PH = decltype(ph_c)::value;
        constexpr int L0 = PH % 3, L1 = (PH + 1) % 3, L2 = (PH + 2) % 3;
        const int cs1 = cs == 0 ? 2 : cs - 1, cs2 = cs1 == 0 ? 2 : cs1 - 1;  // slots of the rows ytop-1, ytop-2
        QRow A2;  // L~ S1 at row ytop-2
        // -------- the stencil units, in the order their sources become available ------------------------------------------
        C3 ca = crow(cs2, 0), cb = crow(cs2, 1);
        auto unit = [&](auto u_c) __attribute__((always_inline)) {
          constexpr int u = decltype(u_c)::value;
          if (u < 4) QS_UNIT<true, false>(A2, S1[L0], u, ca.w, ca.c, ca.e);
          else if (u < 8) QS_UNIT<false, false>(A2, S1[L1], u - 4, cb.w, cb.c, cb.e);
          else if (u < 12) QS_UNIT<true, false>(S1[L2], S0[L0], u - 8, ca.w, ca.c, ca.e);
          else if (u < 16) QS_UNIT<false, false>(S1[L2], S0[L1], u - 12, cb.w, cb.c, cb.e);
          else if (u < 20) QS_UNIT<false, false>(S1[L2], S0[L2], u - 16, ca.w, ca.c, ca.e);
          else QS_UNIT<false, false>(A2, S1[L2], u - 20, cb.w, cb.c, cb.e);
          // behind the last unit of a group: the coefficients of the group after the next
          if (u == 3) ca = crow(cs1, 0);
          if (u == 7) cb = crow(cs1, 1);
          if (u == 11) ca = crow(cs1, 2);
          if (u == 15) {
            cb = crow(cs2, 2);
            // The row ytop of the operand, requested by the previous step (the first: before the loop), is needed from here on.
            // It arrives in `nx` and enters the rotation through this product: dy counts on the output pixels of this work
            // item only (times the workgroup's sign), x everywhere.  (A row loaded straight into its place in the rotation
            // was copied about by the compiler at the top of the step, each copy a wait for the row.)
            const bool rowk = !sideQ || (ytop >= st.y0 && ytop < st.y1);
            float one = sgn;
            asm volatile("" : "+v"(one));  // (pins the products below to this place: they wait for the row)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const float mk = (rowk && colk[t]) ? one : 0.f;  // (what is multiplied by 0 is dy inside the item too: the loads are clamped to it)
#pragma unroll
              for (int e = 0; e < 4; ++e) S0[L2].t[t][e] = nx.t[t][e] * mk;
            }
          }
          if (u == 19) qs_settle<1>(S1[L2]);
        };
        // -------- the matrix work on the row staged by the previous step, ten groups of six, the 24 units spread over them ---
        qs_bf8 fa[2][2], fb[2][2];  // [f-group][hi | lo] of one P plane, [buffer][hi | lo] of two Q planes
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int h = 0; h < 2; ++h) fa[g][h] = frag(QW_P0, h, 2 * fh + g, 0);
#pragma unroll
        for (int h = 0; h < 2; ++h) fb[0][h] = frag(QW_Q0, h, og, 0);
        auto group = [&](auto gi_c) __attribute__((always_inline)) {
          constexpr int gi = decltype(gi_c)::value, blk = gi / 5, g = gi % 5;
          constexpr int pair = g == 0 ? 0 : g == 1 ? 1 : g == 2 ? 3 : g == 3 ? 2 : 4;
          // b buffers by group: Q0 -> 0, Q1 -> 1, (Q1 stays), Q0 -> 0, Q2 -> 1
          constexpr int bb = (g == 0 || g == 3) ? 0 : 1;
          // requests for what comes next (the P plane has one set of registers: its successor is requested when the last
          // instruction that reads it has been issued -- the data takes longer to arrive than that instruction to read)
          if (g == 0) { fb[1][0] = frag(QW_Q1, 0, og, blk); fb[1][1] = frag(QW_Q1, 1, og, blk); }
          if (g == 1) { fb[0][0] = frag(QW_Q0, 0, og, blk); fb[0][1] = frag(QW_Q0, 1, og, blk); }
          if (g == 2) {
#pragma unroll
            for (int gg = 0; gg < 2; ++gg)
#pragma unroll
              for (int h = 0; h < 2; ++h) fa[gg][h] = frag(QW_P2, h, 2 * fh + gg, blk);
          }
          if (g == 3) { fb[1][0] = frag(QW_Q2, 0, og, blk); fb[1][1] = frag(QW_Q2, 1, og, blk); }
          if (g == 0 && blk == 1) {
#pragma unroll
            for (int gg = 0; gg < 2; ++gg)
#pragma unroll
              for (int h = 0; h < 2; ++h) fa[gg][h] = frag(QW_P0, h, 2 * fh + gg, 1);
          }
          if (g == 4 && blk == 0) { fb[0][0] = frag(QW_Q0, 0, og, 1); fb[0][1] = frag(QW_Q0, 1, og, 1); }
          // hi.lo, lo.hi, hi.hi; consecutive instructions go to different accumulators; two stencil units per five of them
          auto mf = [&](auto i_c) __attribute__((always_inline)) {
            constexpr int i = decltype(i_c)::value, j = i >> 1, fg = i & 1, m = gi * 6 + i;  // m = 0 .. 59
            qs_m<false>(acc[pair][fg], fa[fg][j == 1 ? 1 : 0], fb[bb][j == 0 ? 1 : 0]);
            // (all 24 behind the first 48 instructions, the last 12 bare: 3.85 us per step against 3.80)
            if constexpr (m % 5 == 1 || m % 5 == 3) unit(std::integral_constant<int, 2 * (m / 5) + (m % 5 == 3 ? 1 : 0)>{});
          };
          mf(std::integral_constant<int, 0>{}); mf(std::integral_constant<int, 1>{}); mf(std::integral_constant<int, 2>{});
          mf(std::integral_constant<int, 3>{}); mf(std::integral_constant<int, 4>{}); mf(std::integral_constant<int, 5>{});
        };
        group(std::integral_constant<int, 0>{}); group(std::integral_constant<int, 1>{}); group(std::integral_constant<int, 2>{});
        group(std::integral_constant<int, 3>{}); group(std::integral_constant<int, 4>{}); group(std::integral_constant<int, 5>{});
        group(std::integral_constant<int, 6>{}); group(std::integral_constant<int, 7>{}); group(std::integral_constant<int, 8>{});
        group(std::integral_constant<int, 9>{});
        // S2 of row ytop-2
        QRow S2;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) S2.t[t][e] = CHEB ? fmaf(2.f, A2.t[t][e], -S0[L0].t[t][e]) : A2.t[t][e];
        step_barrier();  // every wave has read the staged row
        // (the operands of the last group stay allocated until here: the compiler does not know the asm statements above are
        // matrix instructions, and a vector instruction that reuses an operand register right behind one corrupts it -- seen)
        // (fb[0] died with the last-but-one group: listed as well -- a register set that is reloaded from LDS is safe, the data
        // takes longer to arrive than the instruction to read; one that is simply free is not)
        asm volatile("" : : "v"(fa[0][0]), "v"(fa[0][1]), "v"(fa[1][0]), "v"(fa[1][1]), "v"(fb[1][0]), "v"(fb[1][1]), "v"(fb[0][0]), "v"(fb[0][1]));
        if (!sideQ) {
          stage(QW_P0, S0[L0]);
          stage(QW_P2, S2);
        } else {
          stage(QW_Q0, S0[L0]);
          stage(QW_Q1, S1[L1]);
          stage(QW_Q2, S2);
        }
        if (!sideQ) cstore(cs, cv, cd);  // (the row ytop of L~, requested a step ago)
        row_fetch(ytop + 1, nx);
        if (!sideQ) coef_fetch(ytop + 1);
        cs = cs == 2 ? 0 : cs + 1;
        ++ytop;
        step_barrier();
      };
      for (int t3 = 0; t3 < T3; ++t3) {
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{});
      }
    }
  }
  // the accumulators: lane (n = lane & 15, mg = lane >> 4), element i <-> (f = 16 (2 fh + fg) + 4 mg + i, o = 16 og + n)
#pragma unroll
  for (int i = 0; i < QW_PAIRS; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) asm volatile("s_nop 9\n\ts_nop 9" : "+v"(acc[i][j]) : : "memory");
#pragma unroll
  for (int pr = 0; pr < QW_PAIRS; ++pr)
#pragma unroll
    for (int fg = 0; fg < 2; ++fg)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        slab[((size_t)pr * 64 + (size_t)(16 * (2 * fh + fg) + 4 * (lane >> 4) + i)) * 64 + (size_t)(16 * og + (lane & 15))] = acc[pr][fg][i];
}

}  // namespace dsph
