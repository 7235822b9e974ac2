// Quad-strip form of the fused forward for K = 8, 32 -> 32 channels (round 6; BASELINE configs[3]: nside 2048, K 8, 32 -> 32).
//
// The reference's loop is order-agnostic (gnn_layers.py:140-143); until this round every K > 5 forward ran on the breadth-first
// tile kernel with a 7-ring halo (16 x 16 tiles: 3.5 x the stencil work, LDS-bound: 21 ms at configs[3], four rounds flat).
// This is cheb_qstrip_kernel.h's form -- Clenshaw's recurrence on the MFMA accumulators, y = sum_k T_k(L~)(x W_k), 64-column
// strips streamed along y, four pixels per lane, rows through the rectangle's table of tile bases -- with the seven stencil
// levels dealt to THREE roles instead of two:
//
//   top    levels 7, 6, 5   the H role of the K = 5 kernel as it stands (three levels, nothing comes in);
//   middle levels 4, 3, 2   new: three levels WITH an incoming pair of rows -- B5[new] and the dying B6 row from `top` -- and an
//                           outgoing one, B2[new] and the dying B3 row; three 3-row windows = 144 registers, which fit because 32
//                           input channels need 24 registers of weights where the K = 5 kernel's 64 need 64;
//   bottom levels 1, 0      the L role of the K = 5 kernel (without its H1 variant): y leaves from here.
//
// Why three roles and not four pairs: every hand-over costs the consumer one step of delay, i.e. one more row of x and of L~ in
// the rings, and 16 KiB of hand-over rows; four roles need 171 KiB of LDS (priced in round 5), three need 143.  With 32 output
// channels a role is two waves (16 channels each, the MFMA tile's rows): six waves.  `top` and `bottom` share SIMDs 0 and 1,
// `middle` -- 36 of the step's 84 stencil units -- has SIMDs 2 and 3 to itself.
//
// Timeline of a step with top row ytop (row r of level k is B_k[r]; x row ytop + 1 is fetched during the step):
//   top     B7[ytop], B6[ytop-1], B5[ytop-2]           -> hands B5[ytop-2], B6[ytop-3]
//   middle  B4[ytop-4], B3[ytop-5], B2[ytop-6]         (from B5[.. ytop-3], B6[ytop-4]: what top handed over a step ago)
//   bottom  B1[ytop-8], y[ytop-9]                      (from B2[.. ytop-7], B3[ytop-8])
// so the ring of x holds rows ytop+1 .. ytop-9 (11 rows of 8 KiB: 32 channels as bf16 hi | lo fragments), the ring of L~ ten.
// LDS: 88 + 32 (two hand-overs x two quarters x 8 KiB) + 22.5 KiB + flags = 145,984 B.
// Signs and doubling as in the K = 5 kernel: planes are kept as s_j b_j (qs_wsign), so "- b_{j+2}" is an addition and level j
// enters with -2 L~ when j is odd; the ring holds 2 L~; level 0 runs doubled and the store halves it.
#pragma once

#include <type_traits>

#include "cheb_qstrip_kernel.h"

namespace dsph {

constexpr int Q8_K = 8;
constexpr int Q8_D = 7;                        // halo columns / rows on either side
constexpr int Q8_USE = 48;                     // output columns of a strip's 64: 8 columns of lead-in (a multiple of four), 48, 8 behind
constexpr int Q8_THREADS = 384;                // six waves
constexpr int Q8_RUNIN = 2 * Q8_D + 2;         // steps before the first output row of a run of rows (y leaves at lag K + 1)
constexpr int Q8_WIMG = 3 * 2 * 3 * 2 * QS_FRAG;  // [role][quarter][level of the role][hi | lo] 1 KiB A fragments: 36 KiB (+ 256 B: the f16 image's factor)

struct Q8Args {
  const float* x;
  const float* bias;
  float* y;
  const unsigned char* wimg;
  const float* gvals8;
  const float* gdiag;
  const QStrip* strips;
  const int32_t* tab;
  const int32_t* prefix;
  int64_t x_rows, y_rows;
  int nstrips, N, ld, act;
  int pieces, wg_per_piece;
  float xsc, xsc_inv;  // f16 arithmetic: x is split as x * xsc (a power of two), the store multiplies by xsc_inv
};

// acc = wa . bb + c  (the first product of a chain that starts from another row's registers)
__device__ __forceinline__ void q8_mc(qs_f4& acc, const qs_bf8& wa, const qs_bf8& bb, const qs_f4& c) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(acc) : "v"(wa), "v"(bb), "v"(c) : "memory");
}

// VARIANT 0: six waves, the two `top` waves fetch x and L~; 1: eight waves -- two helper waves on SIMDs 2 and 3 (beside `middle`) do
// the fetching, splitting and filing, which takes ~80 vector instructions a step off the SIMDs that carry two roles
// F16: the three-term split on f16 pairs (DSPH_PREC_F16X3: 11 + 11 mantissa bits of both operands, fp32-equivalent) -- x times the
// caller's power of two (DSPH_OPT_F16_XEXP), the weights times the image's, both taken out again in the store; as cheb_qstrip_kernel.h
template <int VARIANT, bool F16>
__global__ __launch_bounds__(VARIANT == 1 ? 512 : Q8_THREADS, 1) void cheb_qstrip8_kernel(Q8Args a) {
  constexpr bool HELP = VARIANT == 1;
  constexpr int NTHREADS = HELP ? 512 : Q8_THREADS;
  constexpr int K = Q8_K, D = Q8_D, RING = 11, CRING = 10;
  constexpr int ROWB = 2 * 4 * QS_FRAG;          // 8 KiB: one ring row of x ([hi | lo][tile] fragments of the 32 channels)
  constexpr int RINGB = RING * ROWB;             // 88 KiB
  constexpr int HAND1 = 2 * 4 * QS_FRAG;         // 8 KiB per (boundary, quarter): [new row | dying row][tile]
  constexpr int HANDB = 2 * 2 * HAND1;           // 32 KiB
  constexpr int CROWB = 2304;                    // one ring row of L~: [9: the diagonal, directions 0..7][p][tile] floats
  constexpr int CRINGB = CRING * CROWB;
  constexpr int LDS_HAND = RINGB, LDS_C = RINGB + HANDB, LDS_FLAG = LDS_C + CRINGB;
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_FLAG + 64];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // waves 0, 1: top (quarters 0, 1); 2, 3: middle; 4, 5: bottom; (6, 7: helpers) -- wave w runs on SIMD w & 3
  const int role = wave >> 1, oq = wave & 1;
  const int p = lane & 15, q4 = lane >> 4;
  const unsigned lane16 = (unsigned)lane * 16u;
  // hand-over b (0: top -> middle, 1: middle -> bottom) of this quarter, and its counter
  auto hand_at = [&](int b) __attribute__((always_inline)) -> unsigned { return (unsigned)LDS_HAND + (unsigned)(b * 2 + oq) * HAND1 + lane16; };
  auto flag_at = [&](int b) __attribute__((always_inline)) -> unsigned { return (unsigned)LDS_FLAG + 4u * (unsigned)(b * 2 + oq); };
  auto flag_set = [&](unsigned addr, int v) __attribute__((always_inline)) { asm volatile("ds_write_b32 %0, %1" : : "v"(addr), "v"(v) : "memory"); };
  auto flag_get = [&](unsigned addr) __attribute__((always_inline)) -> int {
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
  };
  for (int i = tid; i < (LDS_FLAG + 64) / 16; i += NTHREADS) reinterpret_cast<qs_f4*>(smem)[i] = qs_f4{0.f, 0.f, 0.f, 0.f};

  const int G = gridDim.x, ord = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int piece = ord / a.wg_per_piece, map0 = ord - piece * a.wg_per_piece;
  if (piece >= a.pieces) return;
  const int64_t tape = (int64_t)a.prefix[a.nstrips];
  const int64_t tape_begin = tape * piece / a.pieces, tape_end = tape * (piece + 1) / a.pieces;
  auto locate = [&](int64_t r, int64_t r_end, QStrip& st) __attribute__((always_inline)) -> int {
    int lo = 0, hi = a.nstrips;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if ((int64_t)a.prefix[mid] <= r) lo = mid; else hi = mid;
    }
    {
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
  const unsigned xrowb = 32u * 4u, yrowb = (unsigned)a.ld * 4u;
  auto step_barrier = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

  // ---- rows through the rectangle's table of tile bases (cheb_qstrip_kernel.h) ------------------------------------------------
  typedef int qs_i4 __attribute__((ext_vector_type(4)));
  auto tab_lane = [&](const QStrip& st, unsigned ci, int yrow) __attribute__((always_inline)) -> unsigned {
    const int yc = min(max(yrow, st.ylo), st.yhi);
    const int32_t* trow = a.tab + __builtin_amdgcn_readfirstlane(st.tab + (yc >> 4) * st.tws + (max(st.xs, st.xlo) >> 4));
    qs_i4 b;
    int b4;
    asm volatile("s_load_dwordx4 %0, %2, 0x0\n\ts_load_dword %1, %2, 0x10\n\ts_waitcnt lgkmcnt(0)" : "=&s"(b), "=&s"(b4) : "s"(trow) : "memory");
    return (unsigned)(ci == 0 ? b[0] : ci == 1 ? b[1] : ci == 2 ? b[2] : ci == 3 ? b[3] : b4);
  };
  auto tab_new_row = [&](const QStrip& st, int yrow) __attribute__((always_inline)) -> bool {
    return yrow > st.ylo && yrow <= st.yhi && (yrow & 15) == 0;
  };
  auto row_in = [&](const QStrip& st, unsigned base, unsigned mX, int yrow) __attribute__((always_inline)) -> unsigned {
    const int yc = min(max(yrow, st.ylo), st.yhi);
    return base + (mX | (st_spread((unsigned)yc & 15u) << 1));
  };

  // ---- x (the two `top` waves; `middle` has no register to spare and `bottom` has the stores): wave q fetches tiles 2 q and
  // 2 q + 1 of row ytop + 1 -- lane: half h = lane & 1, pixel p8 = (lane >> 1) & 7 (+ 8 for the second load of a tile), qh = (lane
  // >> 4) & 1, kk = lane >> 5: the 16 bytes at offset kk 64 + (2 qh + h) 16 of the pixel's 128: channels 16 kk + 8 qh + 4 h ..+3 ->
  // fragment of the tile, lane slot (pixel, 2 kk + qh), half h; split hi | lo at the end of the step -----------------------------
  const unsigned x_goff = (unsigned)(lane >> 5) * 64u + (unsigned)(2 * ((lane >> 4) & 1) + (lane & 1)) * 16u;
  auto xw_wait = [&](qs_f4 (&xv)[4]) __attribute__((always_inline)) { asm volatile("" : "+v"(xv[0]), "+v"(xv[1]), "+v"(xv[2]), "+v"(xv[3]) : : "memory"); };
  auto xstore = [&](int slot, const qs_f4 (&xv)[4]) __attribute__((always_inline)) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // load i: tile 2 oq + (i >> 1), pixels p8 + 8 (i & 1)
      const int pix = ((lane >> 1) & 7) + 8 * (i & 1);
      unsigned char* q = smem + (unsigned)slot * ROWB + (unsigned)(2 * oq + (i >> 1)) * QS_FRAG + (unsigned)(lane & 1) * 8u +
                         (unsigned)(pix + 16 * (2 * (lane >> 5) + ((lane >> 4) & 1))) * 16u;
      qs_u2 hi, lo;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float a0 = F16 ? xv[i][2 * j] * a.xsc : xv[i][2 * j], a1 = F16 ? xv[i][2 * j + 1] * a.xsc : xv[i][2 * j + 1];
        if (F16) {  // (a value beyond the f16 range becomes an infinity here and a NaN row in y: loud, not wrong)
          typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
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
      *reinterpret_cast<qs_u2*>(q) = hi;
      *reinterpret_cast<qs_u2*>(q + 4 * QS_FRAG) = lo;
    }
  };
  // ---- L~ (the two `top` waves): wave q fetches the row's values of the pixels 4 p + q and 4 p + q + 2 and files them, doubled,
  // in the ring ------------------------------------------------------------------------------------------------------------
  auto cfetch = [&](unsigned rid, qs_f4& cv, float& cd) __attribute__((always_inline)) {
    cv = *reinterpret_cast<const qs_f4*>(reinterpret_cast<const char*>(a.gvals8) + (size_t)rid * 32u + (unsigned)(q4 & 1) * 16u);
    cd = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.gdiag) + (size_t)rid * 4u);
  };
  auto cstore = [&](int slot, int res, qs_f4 cv, float cd) __attribute__((always_inline)) {
    cv = cv + cv;
    cd = cd + cd;
    unsigned char* q = smem + LDS_C + (unsigned)slot * CROWB + (unsigned)p * 16u + (unsigned)res * 4u;
    if (q4 < 2) {
#pragma unroll
      for (int d = 0; d < 4; ++d) *reinterpret_cast<float*>(q + (unsigned)(1 + 4 * q4 + d) * 256u) = cv[d];
    }
    if (q4 == 2) *reinterpret_cast<float*>(q) = cd;
  };
  auto cvec = [&](const unsigned char* q, int v) __attribute__((always_inline)) -> qs_f4 { return *reinterpret_cast<const qs_f4*>(q + (unsigned)v * 256u); };
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
#define Q8_LO0(c) (c).sw, (c).s, (c).se
#define Q8_LO1(c) (c).w, (c).dg, (c).e
#define Q8_HI(c) (c).nw, (c).n, (c).ne

  // The MFMA chain of one level: 12 instructions (2 pairs of tiles x 3 terms x 2 tiles; consecutive ones go to different tiles),
  // each followed by the stencil units that fall to it (NU units over the first nine); the fragments of the second pair are
  // requested when the first starts.  FIRST: how the first product of a tile starts -- 0 accumulate onto ROW, 1 from zero,
  // 2 from the row CROW (another plane's dying row).  Order of the three terms: W_hi.x_lo, W_lo.x_hi, W_hi.x_hi.
#define Q8_FR(BUF, FADDR, PAIR)                                                                                           \
  _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                                         \
    fr[BUF][u][0] = *reinterpret_cast<const qs_bf8*>(smem + (FADDR) + (unsigned)((PAIR) * 2 + u) * QS_FRAG);              \
    fr[BUF][u][1] = *reinterpret_cast<const qs_bf8*>(smem + (FADDR) + (unsigned)(4 + (PAIR) * 2 + u) * QS_FRAG);          \
  }
#define Q8_CHAIN_B(BUFS, ROW, FIRST, CROW, WLEV, FADDR, NU, ...)                                                         \
  {                                                                                                                       \
    Q8_FR(0, FADDR, 0)                                                                                                    \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                                       \
      if ((BUFS) == 2 && s == 0) { Q8_FR((BUFS) == 2 ? 1 : 0, FADDR, 1) }                                                                   \
      if ((BUFS) == 1 && s == 1) { Q8_FR(0, FADDR, 1) } /* (reloaded from LDS behind their last reader: safe, see cheb_qwgrad_kernel.h) */ \
      _Pragma("unroll") for (int j = 0; j < 3; ++j) {                                                                     \
        _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                                   \
          const int m = (s * 3 + j) * 2 + u;                                                                              \
          const int tt = s * 2 + u;                                                                                       \
          const qs_bf8& wa_ = wr[WLEV][j == 1 ? 1 : 0];                                                                   \
          const qs_bf8& bb_ = fr[(BUFS) == 2 ? s : 0][u][j == 0 ? 1 : 0];                                                 \
          if ((FIRST) == 1 && j == 0) qs_m0<F16>((ROW).t[tt], wa_, bb_);                                                \
          else if ((FIRST) == 2 && j == 0) q8_mc((ROW).t[tt], wa_, bb_, (CROW).t[tt]);                                    \
          else qs_m<F16>((ROW).t[tt], wa_, bb_);                                                                        \
          _Pragma("unroll") for (int qq = (m * (NU)) / 9; qq < ((m + 1) * (NU)) / 9 && m < 9; ++qq) { __VA_ARGS__; }      \
        }                                                                                                                 \
      }                                                                                                                   \
    }                                                                                                                     \
  }
#define Q8_CHAIN(ROW, FIRST, CROW, WLEV, FADDR, NU, ...) Q8_CHAIN_B(2, ROW, FIRST, CROW, WLEV, FADDR, NU, __VA_ARGS__)
#define Q8_CHAIN1(ROW, FIRST, CROW, WLEV, FADDR, NU, ...) Q8_CHAIN_B(1, ROW, FIRST, CROW, WLEV, FADDR, NU, __VA_ARGS__)
#define Q8_CHAINM(ROW, FIRST, CROW, WLEV, FADDR, NU, ...) Q8_CHAIN_B((HELP ? 2 : 1), ROW, FIRST, CROW, WLEV, FADDR, NU, __VA_ARGS__)

  __syncthreads();

  if (role == 0) {
    // =================================================================================================================
    // top: levels 7, 6, 5.  R[0] = B7 rows, R[1] = B6 rows; B5[new] is built in place in the dying B7 row.
    // =================================================================================================================
    qs_bf8 wr[3][2];
    {
      const unsigned char* wp = a.wimg + ((size_t)(0 * 2 + oq) * 3) * (2 * QS_FRAG) + lane16;
#pragma unroll
      for (int l = 0; l < 3; ++l)
#pragma unroll
        for (int h = 0; h < 2; ++h) wr[l][h] = *reinterpret_cast<const qs_bf8*>(wp + (size_t)(l * 2 + h) * QS_FRAG);
    }
    int handed = 0;
    for (int64_t tr = tape_begin; tr < tape_end;) {
      QStrip st;
      tr += locate(tr, tape_end, st);
      for (int nq = map0; nq < a.N; nq += a.wg_per_piece) {
        const int tc0 = max(st.xs, st.xlo) >> 4;
        // what this wave fetches: L~ of the pixels 4 p + oq + 2 i; x of the pixels 4 (p8 + 8 (i & 1)) + 2 oq + (i >> 1), i = 0..3
        unsigned pkC[2], pkF[4];  // (tile column - the strip's first) << 8 | Morton bits of the column inside its tile
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int X = min(max(st.xs + 4 * p + oq + 2 * i, st.xlo), st.xhi);
          pkC[i] = ((unsigned)((X >> 4) - tc0) << 8) | st_spread((unsigned)X & 15u);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int X = min(max(st.xs + 4 * (((lane >> 1) & 7) + 8 * (i & 1)) + 2 * oq + (i >> 1), st.xlo), st.xhi);
          pkF[i] = ((unsigned)((X >> 4) - tc0) << 8) | st_spread((unsigned)X & 15u);
        }
        const int T3 = ((st.y1 - st.y0) + Q8_RUNIN + 2) / 3;
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
        unsigned bF[4], bC[2];  // the lane's tile bases of the rows in flight
        auto bases_x = [&](int yrow) __attribute__((always_inline)) {
#pragma unroll
          for (int i = 0; i < 4; ++i) bF[i] = tab_lane(st, pkF[i] >> 8, yrow);
        };
        auto bases_c = [&](int yrow) __attribute__((always_inline)) {
#pragma unroll
          for (int i = 0; i < 2; ++i) bC[i] = tab_lane(st, pkC[i] >> 8, yrow);
        };
        auto fetch_x = [&](int yrow, qs_f4 (&xv)[4]) __attribute__((always_inline)) {
#pragma unroll
          for (int i = 0; i < 4; ++i) xv[i] = *reinterpret_cast<const qs_f4*>(xmap + ((size_t)row_in(st, bF[i], pkF[i] & 255u, yrow) * xrowb + x_goff));  // (64-bit: a map may exceed 4 GiB)
        };
        auto fetch_c = [&](int yrow, qs_f4 (&cv)[2], float (&cd)[2]) __attribute__((always_inline)) {
#pragma unroll
          for (int i = 0; i < 2; ++i) cfetch(row_in(st, bC[i], pkC[i] & 255u, yrow), cv[i], cd[i]);
        };
        auto cw_wait = [&](qs_f4 (&cv)[2], float (&cd)[2]) __attribute__((always_inline)) {
          asm volatile("" : "+v"(cv[0]), "+v"(cv[1]), "+v"(cd[0]), "+v"(cd[1]) : : "memory");
        };
        if (!HELP) {
          bases_x(ytop);
          bases_c(ytop - 1);
          qs_f4 cv[2], xv[4];
          float cd[2];
          fetch_c(ytop - 1, cv, cd);
          fetch_x(ytop, xv);
          cw_wait(cv, cd);
          cstore(CRING - 1, oq, cv[0], cd[0]);
          cstore(CRING - 1, oq + 2, cv[1], cd[1]);
          xw_wait(xv);
          xstore(0, xv);
          bases_c(ytop);
          bases_x(ytop + 1);
        }
        step_barrier();
        auto step = [&](auto ph_c) __attribute__((always_inline)) {
          constexpr int PH = decltype(ph_c)::value;
          constexpr int L0 = PH % 3, L1 = (PH + 1) % 3, L2 = (PH + 2) % 3;
          int snew = slot_top + 1;
          snew = snew == RING ? 0 : snew;
          auto slot_ix = [&](int back) __attribute__((always_inline)) -> int { int s = slot_top - back; s += s < 0 ? RING : 0; return s; };
          auto cslot_ix = [&](int back) __attribute__((always_inline)) -> int { int s = cs_top - back; s += s < 0 ? CRING : 0; return s; };
          qs_f4 xv[4], cv[2];
          float cd[2];
          if (!HELP) {
            if (tab_new_row(st, ytop + 1)) bases_x(ytop + 1);
            if (tab_new_row(st, ytop)) bases_c(ytop);
            fetch_x(ytop + 1, xv);
            fetch_c(ytop, cv, cd);
          }
          const unsigned f0 = (unsigned)slot_top * ROWB + lane16, f1 = (unsigned)slot_ix(1) * ROWB + lane16, f2 = (unsigned)slot_ix(2) * ROWB + lane16;
          qs_bf8 fr[2][2][2];
          const QCoefLo c6 = clo_read(cslot_ix(1));  // row ytop-1: level 6
          // s0: z_7 -> B7[new] | B6[new] = + (B7[-2], B7[-1])                     (level 6 enters with +2 L~)
          Q8_CHAIN(R[0][L2], 1, R[0][L2], 0, f0, 2 * QS_UPR,
                   { if (qq < QS_UPR) QS_UNIT<true, false>(R[1][L2], R[0][L0], qq, Q8_LO0(c6)); else QS_UNIT<false, false>(R[1][L2], R[0][L1], qq - QS_UPR, Q8_LO1(c6)); })
          qs_settle<9>(R[0][L2]);
          qs_settle<1>(R[1][L2]);
          const QCoefLo c5 = clo_read(cslot_ix(2));  // row ytop-2: level 5
          const QCoefHi c6h = chi_read(cslot_ix(1));
          // s1: z_6 -> B6[new] | B5[new] = B7[-2] - (B6[-2], B6[-1]), in place in R[0][L0]      (level 5 enters with -2 L~)
          Q8_CHAIN(R[1][L2], 0, R[1][L2], 1, f1, 2 * QS_UPR,
                   { if (qq < QS_UPR) QS_UNIT<false, true>(R[0][L0], R[1][L0], qq, Q8_LO0(c5)); else QS_UNIT<false, true>(R[0][L0], R[1][L1], qq - QS_UPR, Q8_LO1(c5)); })
          qs_settle<9>(R[1][L2]);
          qs_settle<1>(R[0][L0]);
          const QCoefHi c5h = chi_read(cslot_ix(2));
          // s2: z_5 -> B5[new] | B6[new] += B7[new]
          Q8_CHAIN(R[0][L0], 0, R[0][L0], 2, f2, QS_UPR, { QS_UNIT<false, false>(R[1][L2], R[0][L2], qq, Q8_HI(c6h)); })
          qs_settle<9>(R[0][L0]);
          qs_settle<1>(R[1][L2]);
          // s3: B5[new] -= B6[new]
#pragma unroll
          for (int qq = 0; qq < QS_UPR; ++qq) QS_UNIT<false, true>(R[0][L0], R[1][L2], qq, Q8_HI(c5h));
          // hand-over: B5[new] and the dying row of B6 -- once `middle` has taken the previous pair
          for (int spin = 0; flag_get(flag_at(0)) <= handed && spin < (1 << 22); ++spin) {}
          {
            unsigned char* hp = smem + hand_at(0);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              *reinterpret_cast<qs_f4*>(hp + t * QS_FRAG) = R[0][L0].t[t];
              *reinterpret_cast<qs_f4*>(hp + (4 + t) * QS_FRAG) = R[1][L0].t[t];
            }
          }
          ++handed;
          if (!HELP) {
            xw_wait(xv);
            cw_wait(cv, cd);
            cstore(cs_top, oq, cv[0], cd[0]);
            cstore(cs_top, oq + 2, cv[1], cd[1]);
            xstore(snew, xv);
          }
          slot_top = snew;
          cs_top = cs_top + 1 == CRING ? 0 : cs_top + 1;
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
  } else if (role == 1) {
    // =================================================================================================================
    // middle: levels 4, 3, 2.  R[0] = B5 rows (from top), R[1] = B4 rows (the new one starts as the dying B6 row from top),
    // R[2] = B3 rows (the new one starts as a copy of the dying B5 row); B2[new] is built in place in the dying B4 row.
    // =================================================================================================================
    qs_bf8 wr[3][2];
    {
      const unsigned char* wp = a.wimg + ((size_t)(1 * 2 + oq) * 3) * (2 * QS_FRAG) + lane16;
#pragma unroll
      for (int l = 0; l < 3; ++l)
#pragma unroll
        for (int h = 0; h < 2; ++h) wr[l][h] = *reinterpret_cast<const qs_bf8*>(wp + (size_t)(l * 2 + h) * QS_FRAG);
    }
    int handed = 0, taken = 0;
    for (int64_t tr = tape_begin; tr < tape_end;) {
      QStrip st;
      tr += locate(tr, tape_end, st);
      for (int nq = map0; nq < a.N; nq += a.wg_per_piece) {
        const int T3 = ((st.y1 - st.y0) + Q8_RUNIN + 2) / 3;
        QRow R[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t) R[i][s].t[t] = qs_f4{0.f, 0.f, 0.f, 0.f};
        int ytop = st.y0 - D, slot_top = 0, cs_top = 0;
        step_barrier();
        step_barrier();
        auto step = [&](auto ph_c) __attribute__((always_inline)) {
          constexpr int PH = decltype(ph_c)::value;
          constexpr int L0 = PH % 3, L1 = (PH + 1) % 3, L2 = (PH + 2) % 3;
          int snew = slot_top + 1;
          snew = snew == RING ? 0 : snew;
          auto slot_ix = [&](int back) __attribute__((always_inline)) -> int { int s = slot_top - back; s += s < 0 ? RING : 0; return s; };
          auto cslot_ix = [&](int back) __attribute__((always_inline)) -> int { int s = cs_top - back; s += s < 0 ? CRING : 0; return s; };
          // the rows `top` left at the end of the previous step: B5[new] -> R[0][L2], the dying B6 row -> R[1][L2] (= B4[new] so far)
          {
            const unsigned char* hp = smem + hand_at(0);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              R[0][L2].t[t] = *reinterpret_cast<const qs_f4*>(hp + t * QS_FRAG);
              R[1][L2].t[t] = *reinterpret_cast<const qs_f4*>(hp + (4 + t) * QS_FRAG);
            }
          }
          ++taken;
          flag_set(flag_at(0), taken);  // (LDS operations of a wave complete in order: the reads above are done first)
          const unsigned f4 = (unsigned)slot_ix(4) * ROWB + lane16, f5 = (unsigned)slot_ix(5) * ROWB + lane16, f6 = (unsigned)slot_ix(6) * ROWB + lane16;
          qs_bf8 fr[HELP ? 2 : 1][2][2];  // (six waves: one set of B fragments -- 16 registers that the three windows need; eight: two)
          // Rows of this step: B5 window = rows ytop-5, -4, -3 (R[0][L0..L2]); B4[new] = row ytop-4; B3[new] = row ytop-5;
          // B2[new] = row ytop-6.  A row takes its stencil units and its chain of matrix instructions in DIFFERENT slots
          // (the asm statements hide both from the hazard recogniser), and a slot's units use one row's coefficients (registers).
          // B3[new] starts as the dying B5 row (which stays a source of level 4 below)
#pragma unroll
          for (int t = 0; t < 4; ++t) R[2][L2].t[t] = R[0][L0].t[t];
          asm volatile("" : "+v"(R[2][L2].t[0]), "+v"(R[2][L2].t[1]), "+v"(R[2][L2].t[2]), "+v"(R[2][L2].t[3]) : : "memory");
          {
            const QCoefLo c3 = clo_read(cslot_ix(5));  // row ytop-5: level 3
            // m0: z_4 -> B4[new] (onto the B6 row) | B3[new] -= (B4[-2], B4[-1])                  (level 3 enters with -2 L~)
            Q8_CHAINM(R[1][L2], 0, R[1][L2], 0, f4, 2 * QS_UPR,
                     { if (qq < QS_UPR) QS_UNIT<false, true>(R[2][L2], R[1][L0], qq, Q8_LO0(c3)); else QS_UNIT<false, true>(R[2][L2], R[1][L1], qq - QS_UPR, Q8_LO1(c3)); })
          }
          qs_settle<9>(R[1][L2]);
          qs_settle<1>(R[2][L2]);
          {
            const QCoefLo c4 = clo_read(cslot_ix(4));  // row ytop-4: level 4
            const QCoefHi c4h = chi_read(cslot_ix(4));
            // m1: z_3 -> B3[new] | B4[new] += (B5[-1], B5[0], B5[+1])                             (level 4 enters with +2 L~)
            Q8_CHAINM(R[2][L2], 0, R[2][L2], 1, f5, 3 * QS_UPR,
                     { if (qq < QS_UPR) QS_UNIT<false, false>(R[1][L2], R[0][L0], qq, Q8_LO0(c4));
                       else if (qq < 2 * QS_UPR) QS_UNIT<false, false>(R[1][L2], R[0][L1], qq - QS_UPR, Q8_LO1(c4));
                       else QS_UNIT<false, false>(R[1][L2], R[0][L2], qq - 2 * QS_UPR, Q8_HI(c4h)); })
          }
          qs_settle<9>(R[2][L2]);
          qs_settle<1>(R[1][L2]);
          {
            const QCoefHi c3h = chi_read(cslot_ix(5));
            // m2: z_2 -> B2[new], in place on the dying B4 row | B3[new] -= B4[new] (its row y+1)
            Q8_CHAINM(R[1][L0], 0, R[1][L0], 2, f6, QS_UPR, { QS_UNIT<false, true>(R[2][L2], R[1][L2], qq, Q8_HI(c3h)); })
          }
          qs_settle<9>(R[1][L0]);
          qs_settle<1>(R[2][L2]);
          {
            const QCoefLo c2 = clo_read(cslot_ix(6));  // row ytop-6: level 2
            const QCoefHi c2h = chi_read(cslot_ix(6));
            // m3: B2[new] += (B3[-1], B3[0], B3[+1])                                               (level 2 enters with +2 L~)
#pragma unroll
            for (int qq = 0; qq < 3 * QS_UPR; ++qq) {
              if (qq < QS_UPR) QS_UNIT<false, false>(R[1][L0], R[2][L0], qq, Q8_LO0(c2));
              else if (qq < 2 * QS_UPR) QS_UNIT<false, false>(R[1][L0], R[2][L1], qq - QS_UPR, Q8_LO1(c2));
              else QS_UNIT<false, false>(R[1][L0], R[2][L2], qq - 2 * QS_UPR, Q8_HI(c2h));
            }
          }
          // hand-over: B2[new] and the dying row of B3 -- once `bottom` has taken the previous pair
          for (int spin = 0; flag_get(flag_at(1)) <= handed && spin < (1 << 22); ++spin) {}
          {
            unsigned char* hp = smem + hand_at(1);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              *reinterpret_cast<qs_f4*>(hp + t * QS_FRAG) = R[1][L0].t[t];
              *reinterpret_cast<qs_f4*>(hp + (4 + t) * QS_FRAG) = R[2][L0].t[t];
            }
          }
          ++handed;
          (void)ytop;
          slot_top = snew;
          cs_top = cs_top + 1 == CRING ? 0 : cs_top + 1;
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
  } else if (role == 3) {
    // =================================================================================================================
    // helpers (VARIANT 1): fetch, split and file the rows of x and L~ that `top` otherwise fetches -- nothing else.
    // =================================================================================================================
    for (int64_t tr = tape_begin; tr < tape_end;) {
      QStrip st;
      tr += locate(tr, tape_end, st);
      for (int nq = map0; nq < a.N; nq += a.wg_per_piece) {
        const int tc0 = max(st.xs, st.xlo) >> 4;
        unsigned pkC[2], pkF[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int X = min(max(st.xs + 4 * p + oq + 2 * i, st.xlo), st.xhi);
          pkC[i] = ((unsigned)((X >> 4) - tc0) << 8) | st_spread((unsigned)X & 15u);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int X = min(max(st.xs + 4 * (((lane >> 1) & 7) + 8 * (i & 1)) + 2 * oq + (i >> 1), st.xlo), st.xhi);
          pkF[i] = ((unsigned)((X >> 4) - tc0) << 8) | st_spread((unsigned)X & 15u);
        }
        const int T3 = ((st.y1 - st.y0) + Q8_RUNIN + 2) / 3;
        const char* __restrict__ xmap = reinterpret_cast<const char*>(a.x) + (size_t)nq * a.x_rows * xrowb;
        int ytop = st.y0 - D, slot_top = 0, cs_top = 0;
        step_barrier();
        unsigned bF[4], bC[2];
        auto bases_x = [&](int yrow) __attribute__((always_inline)) {
#pragma unroll
          for (int i = 0; i < 4; ++i) bF[i] = tab_lane(st, pkF[i] >> 8, yrow);
        };
        auto bases_c = [&](int yrow) __attribute__((always_inline)) {
#pragma unroll
          for (int i = 0; i < 2; ++i) bC[i] = tab_lane(st, pkC[i] >> 8, yrow);
        };
        auto fetch_x = [&](int yrow, qs_f4 (&xv)[4]) __attribute__((always_inline)) {
#pragma unroll
          for (int i = 0; i < 4; ++i) xv[i] = *reinterpret_cast<const qs_f4*>(xmap + ((size_t)row_in(st, bF[i], pkF[i] & 255u, yrow) * xrowb + x_goff));
        };
        auto fetch_c = [&](int yrow, qs_f4 (&cv)[2], float (&cd)[2]) __attribute__((always_inline)) {
#pragma unroll
          for (int i = 0; i < 2; ++i) cfetch(row_in(st, bC[i], pkC[i] & 255u, yrow), cv[i], cd[i]);
        };
        bases_x(ytop);
        bases_c(ytop - 1);
        {
          qs_f4 cv[2], xv[4];
          float cd[2];
          fetch_c(ytop - 1, cv, cd);
          fetch_x(ytop, xv);
          cstore(CRING - 1, oq, cv[0], cd[0]);
          cstore(CRING - 1, oq + 2, cv[1], cd[1]);
          xstore(0, xv);
        }
        bases_c(ytop);
        bases_x(ytop + 1);
        step_barrier();
        for (int t = 0; t < 3 * T3; ++t) {
          int snew = slot_top + 1;
          snew = snew == RING ? 0 : snew;
          qs_f4 xv[4], cv[2];
          float cd[2];
          if (tab_new_row(st, ytop + 1)) bases_x(ytop + 1);
          if (tab_new_row(st, ytop)) bases_c(ytop);
          fetch_x(ytop + 1, xv);
          fetch_c(ytop, cv, cd);
          cstore(cs_top, oq, cv[0], cd[0]);
          cstore(cs_top, oq + 2, cv[1], cd[1]);
          xstore(snew, xv);
          slot_top = snew;
          cs_top = cs_top + 1 == CRING ? 0 : cs_top + 1;
          ++ytop;
          step_barrier();
        }
      }
    }
  } else {
    // =================================================================================================================
    // bottom: levels 1, 0.  R[0] = B2 rows (from middle), R[1] = B1 rows (the new one starts as the dying B3 row), Y = 2 y.
    // =================================================================================================================
    qs_bf8 wr[2][2];
    {
      const unsigned char* wp = a.wimg + ((size_t)(2 * 2 + oq) * 3) * (2 * QS_FRAG) + lane16;
#pragma unroll
      for (int l = 0; l < 2; ++l)
#pragma unroll
        for (int h = 0; h < 2; ++h) wr[l][h] = *reinterpret_cast<const qs_bf8*>(wp + (size_t)(l * 2 + h) * QS_FRAG);
    }
    const float floor_v = a.act == DSPH_ACT_RELU ? 0.f : -__builtin_huge_valf();
    const float ysc = 0.5f * (F16 ? *reinterpret_cast<const float*>(a.wimg + Q8_WIMG) * a.xsc_inv : 1.f);
    qs_f4 bv = qs_f4{0.f, 0.f, 0.f, 0.f};
    if (a.bias != nullptr) bv = *reinterpret_cast<const qs_f4*>(a.bias + 16 * oq + 4 * q4);
    int taken = 0;
    for (int64_t tr = tape_begin; tr < tape_end;) {
      QStrip st;
      tr += locate(tr, tape_end, st);
      for (int nq = map0; nq < a.N; nq += a.wg_per_piece) {
        const int tc0 = max(st.xs, st.xlo) >> 4;
        const int Xg = st.xs + 4 * p;
        const unsigned ciY = (unsigned)((min(max(Xg, st.xlo), st.xhi) >> 4) - tc0), mXg = st_spread((unsigned)Xg & 12u);
        const int cfirst = st.x0 - st.xs, clast = cfirst + st.w;
        const int T3 = ((st.y1 - st.y0) + Q8_RUNIN + 2) / 3;
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
        step_barrier();
        unsigned bY = tab_lane(st, ciY, st.y0);
        // (the stored row's Morton bits inside its tile, stepped with the row: cheb_qstrip_kernel.h, my_next)
        unsigned myY = st_spread((unsigned)(ytop - (K + 1)) & 15u) << 1;
        step_barrier();
        auto step = [&](auto ph_c) __attribute__((always_inline)) {
          constexpr int PH = decltype(ph_c)::value;
          constexpr int L0 = PH % 3, L1 = (PH + 1) % 3, L2 = (PH + 2) % 3;
          int snew = slot_top + 1;
          snew = snew == RING ? 0 : snew;
          auto slot_ix = [&](int back) __attribute__((always_inline)) -> int { int s = slot_top - back; s += s < 0 ? RING : 0; return s; };
          auto cslot_ix = [&](int back) __attribute__((always_inline)) -> int { int s = cs_top - back; s += s < 0 ? CRING : 0; return s; };
          const int yr = ytop - (K + 1);
          if (myY == 0) bY = tab_lane(st, ciY, yr);
          // the rows `middle` left at the end of the previous step: B2[new] -> R[0][L2], the dying B3 row -> R[1][L2] (= B1[new] so far)
          {
            const unsigned char* hp = smem + hand_at(1);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              R[0][L2].t[t] = *reinterpret_cast<const qs_f4*>(hp + t * QS_FRAG);
              R[1][L2].t[t] = *reinterpret_cast<const qs_f4*>(hp + (4 + t) * QS_FRAG);
            }
          }
          ++taken;
          flag_set(flag_at(1), taken);
          const unsigned f1 = (unsigned)slot_ix(8) * ROWB + lane16, f0 = (unsigned)slot_ix(9) * ROWB + lane16;
          qs_bf8 fr[2][2][2];
          const QCoefLo c1 = clo_read(cslot_ix(8));  // row ytop-8: level 1
          const QCoefHi c1h = chi_read(cslot_ix(8));
#pragma unroll
          for (int t = 0; t < 4; ++t) Y.t[t] = R[0][L0].t[t] + R[0][L0].t[t];
          qs_settle<1>(Y);
          // b0: 2 z_0 + 2 B2[-1] -> Y | B1[new] (= B3 row) -= (B2[-1], B2[0], B2[+1])              (level 1 enters with -2 L~)
          Q8_CHAIN(Y, 0, Y, 1, f0, 3 * QS_UPR,
                   { if (qq < QS_UPR) QS_UNIT<false, true>(R[1][L2], R[0][L0], qq, Q8_LO0(c1));
                     else if (qq < 2 * QS_UPR) QS_UNIT<false, true>(R[1][L2], R[0][L1], qq - QS_UPR, Q8_LO1(c1));
                     else QS_UNIT<false, true>(R[1][L2], R[0][L2], qq - 2 * QS_UPR, Q8_HI(c1h)); })
          qs_settle<9>(Y);
          qs_settle<1>(R[1][L2]);
          const QCoefLo c0 = clo_read(cslot_ix(9));  // row ytop-9: level 0
          const QCoefHi c0h = chi_read(cslot_ix(9));
          // b1: z_1 -> B1[new] | Y += (B1[-2], B1[-1])                                              (level 0 enters with +2 L~ into the doubled Y)
          Q8_CHAIN(R[1][L2], 0, R[1][L2], 0, f1, 2 * QS_UPR,
                   { if (qq < QS_UPR) QS_UNIT<false, false>(Y, R[1][L0], qq, Q8_LO0(c0)); else QS_UNIT<false, false>(Y, R[1][L1], qq - QS_UPR, Q8_LO1(c0)); })
          qs_settle<9>(R[1][L2]);
          // b2: Y += B1[new]: y of row ytop - 9
#pragma unroll
          for (int qq = 0; qq < QS_UPR; ++qq) QS_UNIT<false, false>(Y, R[1][L2], qq, Q8_HI(c0h));
          {
            const bool row_ok = yr >= st.y0 && yr < st.y1;
            const unsigned rowg = bY + (mXg | myY);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const int c = 4 * p + t;
              if (row_ok && c >= cfirst && c < clast) {
                qs_f4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const float v = fmaf(Y.t[t][e], ysc, bv[e]);
                  o[e] = F16 ? (v < floor_v ? floor_v : v) : fmaxf(v, floor_v);  // (f16: a NaN must reach y, not be floored away)
                }
                *reinterpret_cast<qs_f4*>(ymap + (size_t)(rowg + (unsigned)((t & 1) + 4 * (t >> 1))) * yrowb + (unsigned)(16 * oq + 4 * q4) * 4u) = o;
              }
            }
          }
          slot_top = snew;
          cs_top = cs_top + 1 == CRING ? 0 : cs_top + 1;
          ++ytop;
          myY = ((myY | 0x55u) + 1u) & 0xaau;
          step_barrier();
        };
        for (int t3 = 0; t3 < T3; ++t3) {
          step(std::integral_constant<int, 0>{});
          step(std::integral_constant<int, 1>{});
          step(std::integral_constant<int, 2>{});
        }
      }
    }
  }
#undef Q8_CHAIN
#undef Q8_CHAIN1
#undef Q8_CHAINM
#undef Q8_CHAIN_B
#undef Q8_FR
#undef Q8_LO0
#undef Q8_LO1
#undef Q8_HI
}

}  // namespace dsph
