// Input-side strip form of the fused Chebyshev forward (round 4): layers with few input channels.
//
// The strip kernel of round 3 (cheb_strip_kernel.h) runs Clenshaw's recurrence on OUTPUT-channel planes; that is the right
// side when the layer has as many input as output channels.  The layers a DeepSphere stack starts with have few inputs
// and more outputs (1 -> 16 -> 32 -> 64, reference tests/test_healpy_networks.py:96-107; BASELINE configs[0] and [1]):
// there the recurrence belongs on the INPUT side, exactly as the reference writes it (gnn_layers.py:134-143),
//     T_0 = x,  T_1 = L~ x,  T_k = 2 L~ T_{k-1} - T_{k-2},      y = sum_k T_k W_k            (gnn_layers.py:144-150)
// and with at most 16 input channels a whole strip of it fits ONE wave:
//   * lane = (pixel column px = lane & 31 of a 32-pixel strip row, half g = lane >> 5), registers = CH of the row's input
//     channels (CH = 8: channels 8 g .. 8 g + 7 of 16; CH = 4: channels 4 g .. 4 g + 3 of 8) -- which is, register for
//     register, the B operand of v_mfma_f32_32x32x16_bf16 (column = pixel, inner index 8 g + j): a plane row goes from
//     the recurrence into the matrix pipe with a bf16 split and nothing else, no LDS, no re-layout;
//   * the strip is streamed along y, level k lagging k rows behind x: the y-neighbours of a pixel are the previous / next
//     rows in the same lane's registers (three rows per plane, rotated through a 3-phase unrolled step), the
//     x-neighbours the adjacent lanes, read through the DPP operand of the multiply-add (as in cheb_strip_kernel.h);
//   * y[r] collects W_0 T_0[r] + W_1 T_1[r] + W_2 T_2[r] in the step that makes T_2[r] (T_0[r] and T_1[r] are still in
//     their rings then), W_3 T_3[r] one step later, W_4 T_4[r] after that: three accumulator rows in flight; the weights
//     of all levels sit in LDS as MFMA A-operand fragments (10 KiB, read per contraction: fifteen 1 KiB reads per step);
//   * a wave needs nothing from any other wave: no barrier, no hand-over, no roles.  A workgroup is eight independent
//     workers; the work items are (strip segment, map).  LDS holds, per wave, a ring of the last rows of L~ (nine values per
//     pixel) and the 4 KiB block through which a finished y row is turned so that eight lanes store 128 contiguous bytes.
// x is read once per strip row (1.33 x in all: 24 of 32 columns are output) straight into the registers, y written once.
// Shapes: K = 2 .. 5, at most 16 input channels (a multiple of four; other counts arrive zero-padded), 32 output columns
// per launch (wider layers: one launch per block; the layer's width a multiple of four: 16-byte stores), all three
// contraction arithmetics, both bases.
// Which pixels: the strip rectangles of cheb_fused.hip (class-R tiles), cut into single strips and short segments.
#pragma once

#include <type_traits>

#include "cheb_strip_kernel.h"

namespace dsph {

constexpr int IS_THREADS = 512;
constexpr int IS_WAVES = 8;
constexpr int IS_CRING = 6;                    // rows of L~ held per wave: ytop - 4 .. ytop + 1
constexpr int IS_CROWB = 32 * 32 + 32 * 4;     // one ring row: [directions 0-3][px] | [directions 4-7][px] | [px] diagonal
constexpr int IS_YSTB = 4096;                  // y staging block: 32 pixels x 32 channels
constexpr int IS_WAVEB = IS_CRING * IS_CROWB + 2 * IS_YSTB;  // (two y blocks: the pooled epilogue keeps an even and an odd row)

struct IStripArgs {
  const float* x;
  const float* bias;          // of this launch's 32-column block, or NULL
  float* y;                   // column 0 of this launch's block
  const unsigned char* wimg;  // istrip_wprep_kernel: [level][term][64 lanes][16 B]
  const float* gvals8;
  const float* gdiag;
  const StripPair* pairs;     // every pair is two single strips here
  int64_t x_rows, y_rows;
  int npairs, N, Fin, Fout, ld, act;  // Fout: columns of this block (<= 32)
  int nseg;                           // every strip is cut into nseg row segments (chosen per call: istrip_segments)
  float* ypool;                       // cheb_istrip1_kernel, pool != 0: the 2 x 2 NEST-pooled output (column 0 of this launch's block), ypool_rows rows per map
  int64_t ypool_rows;
  int pool;                           //   0 none (y is written), 1 max, 2 mean of the four children (y is NOT written by this kernel)
  int pair;                           // cheb_istrip1_kernel: two maps per wave (one input channel, at most 16 output columns)
  int cheb;                           // T_k = 2 L~ T_{k-1} - T_{k-2} (1, Chebyshev) or L~ T_{k-1} (0, monomial), k >= 2
};

typedef float is_f32x8 __attribute__((ext_vector_type(8)));

// number of A-operand fragments per level: hi | lo (three-term split), hi | mid | lo (six-term), 8 fp32 steps (exact)
__host__ __device__ constexpr int is_terms(int prec) { return prec == DSPH_PREC_BF16X3 ? 2 : (prec == DSPH_PREC_BF16X6 ? 3 : 8); }
__host__ __device__ constexpr int is_term_bytes(int prec) { return prec == DSPH_PREC_FP32 ? 256 : 1024; }

// acc += cw * src[x-1] + cc * src[x] + ce * src[x+1] for the lane's CH channels (see sp_row)
template <int CH>
__device__ __forceinline__ void is_row(float (&acc)[CH], const float (&src)[CH], float cw, float cc, float ce) {
#pragma unroll
  for (int c = 0; c < CH; ++c) acc[c] = fmaf(cc, src[c], acc[c]);
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    float t = acc[c];
    sp_fmac_left(t, src[c], cw);
    sp_fmac_right(t, src[c], ce);
    acc[c] = t;
  }
}
// (a VALU result read through DPP, or as an MFMA operand, needs two wait states: the row as an operand of an s_nop)
template <int CH> __device__ __forceinline__ void is_fence(float (&v)[CH]) {
  if (CH == 8)
    asm volatile("s_nop 1" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4 % CH]), "+v"(v[5 % CH]), "+v"(v[6 % CH]), "+v"(v[7 % CH]));
  else
    asm volatile("s_nop 1" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
}

// the B operand(s) of a plane row: 8 inner-index slots per lane, slot j <- channel j of the lane's CH (zero beyond CH)
struct IsFrag {
  sp_bf16x8 t[3];  // hi, lo (three-term) / hi, mid, lo (six-term)
};
template <int CH, int PREC>
__device__ __forceinline__ IsFrag is_split(const float (&v)[CH]) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 w[3] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
#pragma unroll
  for (int j = 0; j < CH / 2; ++j) {
    const float a0 = v[2 * j], a1 = v[2 * j + 1];
    if (PREC == DSPH_PREC_BF16X3) {
      // per pair: one packed convert for the two hi halves, a shift and a mask to get them back as floats, two subtractions,
      // one packed convert for the lo halves (cheb_strip_kernel.h, xstore)
      const unsigned hu = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a0, a1}, bf16x2));
      const float h0 = __builtin_bit_cast(float, hu << 16), h1 = __builtin_bit_cast(float, hu & 0xffff0000u);
      w[0][j] = hu;
      w[1][j] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a0 - h0, a1 - h1}, bf16x2));
    } else {
      // exact split into 8 + 8 + 8 mantissa bits by truncation (cheb_struct_kernel.h, st_contract): a = h + m + l
      const unsigned u0 = __builtin_bit_cast(unsigned, a0), u1 = __builtin_bit_cast(unsigned, a1);
      const float r0 = a0 - __builtin_bit_cast(float, u0 & 0xffff0000u), r1 = a1 - __builtin_bit_cast(float, u1 & 0xffff0000u);
      const unsigned q0 = __builtin_bit_cast(unsigned, r0), q1 = __builtin_bit_cast(unsigned, r1);
      const float l0 = r0 - __builtin_bit_cast(float, q0 & 0xffff0000u), l1 = r1 - __builtin_bit_cast(float, q1 & 0xffff0000u);
      w[0][j] = (u0 >> 16) | (u1 & 0xffff0000u);
      w[1][j] = (q0 >> 16) | (q1 & 0xffff0000u);
      w[2][j] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{l0, l1}, bf16x2));  // (at most 8 significant bits left: exact)
    }
  }
  IsFrag f;
#pragma unroll
  for (int t = 0; t < 3; ++t) f.t[t] = __builtin_bit_cast(sp_bf16x8, w[t]);
  return f;
}

// acc (+)= W_level . row: the contraction of one plane row with the weights of its level
template <int CH, int PREC, bool INIT>
__device__ __forceinline__ void is_contract(sp_f32x16& acc, const float (&row)[CH], const unsigned char* __restrict__ wlev, int lane) {
  if (INIT) {
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.f;
  }
  if (PREC == DSPH_PREC_FP32) {
    // v_mfma_f32_32x32x2_f32: inner index = 2 s + (lane >> 5); the lane's operand of step s is slot j = s of the OTHER layout,
    // so the eight steps walk the lane's eight slots: step s contracts channels {slot s of half 0, slot s of half 1}
#pragma unroll
    for (int s = 0; s < CH; ++s) {
      const float b = row[s];
      const float wa = *reinterpret_cast<const float*>(wlev + s * 256 + lane * 4);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa, b, acc, 0, 0, 0);
    }
    return;
  }
  const IsFrag f = is_split<CH, PREC>(row);
  const sp_bf16x8 w0 = *reinterpret_cast<const sp_bf16x8*>(wlev + lane * 16);
  const sp_bf16x8 w1 = *reinterpret_cast<const sp_bf16x8*>(wlev + 1024 + lane * 16);
  if (PREC == DSPH_PREC_BF16X3) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, f.t[1], acc, 0, 0, 0);  // small terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, f.t[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, f.t[0], acc, 0, 0, 0);
  } else {
    const sp_bf16x8 w2 = *reinterpret_cast<const sp_bf16x8*>(wlev + 2048 + lane * 16);
    // the six products down to 2^-16: hl, lh, mm, hm, mh, hh (w index first)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, f.t[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, f.t[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, f.t[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, f.t[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, f.t[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, f.t[0], acc, 0, 0, 0);
  }
}

template <int K, int CH, int PREC>
__global__ __launch_bounds__(IS_THREADS, 2) void cheb_istrip_kernel(IStripArgs a) {
  constexpr int D = K - 1;
  constexpr int G = K - 1 < 2 ? K - 1 : 2;          // y[r] is opened in the step that makes T_G[r]
  constexpr int NP = K - 1;                          // planes with a ring: T_0 .. T_{K-2}
  constexpr int WLEVB = is_terms(PREC) * is_term_bytes(PREC);
  __shared__ __attribute__((aligned(16))) unsigned char smem[IS_WAVES * IS_WAVEB + K * WLEVB + 128];
  unsigned char* const sW = smem + IS_WAVES * IS_WAVEB;   // the weights of all levels (read per contraction: 2 - 3 ds_read_b128)
  float* const sBias = reinterpret_cast<float*>(sW + K * WLEVB);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = lane & 31, g = lane >> 5;
  unsigned char* const cring = smem + wave * IS_WAVEB;
  unsigned char* const yst = cring + IS_CRING * IS_CROWB;

  for (int i = tid; i < K * WLEVB / 16; i += IS_THREADS) reinterpret_cast<sp_f32x4*>(sW)[i] = reinterpret_cast<const sp_f32x4*>(a.wimg)[i];
  if (tid < 32) sBias[tid] = (a.bias != nullptr && tid < a.Fout) ? a.bias[tid] : 0.f;
  __syncthreads();  // (the only barrier of the kernel)

  const int G_ = gridDim.x, xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3;
  const int nslots = (G_ + 7 - xcd) / 8;
  // work items: (pair, strip of the pair, row segment, map); a contiguous eighth of them per XCD, dealt to the XCD's waves in turn
  // Two maps per wave (a.pair; CH = 4 only: at most four input channels and at most 16 output columns, e.g. the 4 -> 8 layers behind
  // a pseudo-convolution, reference tests/test_healpy_networks.py:96-107): half g of the wave carries map 2 n + g instead of
  // channels 4 .. 7 that do not exist; rows 0 .. 15 of every level's weight image hold W in the inner slots of half 0, rows
  // 16 .. 31 in those of half 1, so an accumulator row is [map 2 n's 16 columns | map 2 n + 1's] (see cheb_istrip1_kernel below).
  const bool pairm = CH == 4 && __builtin_amdgcn_readfirstlane(a.pair) != 0;
  const int NI = pairm ? (a.N + 1) / 2 : a.N;
  const int64_t n_items = (int64_t)a.npairs * 2 * a.nseg * NI;
  const int64_t q_begin = n_items * xcd / 8, q_end = n_items * (xcd + 1) / 8;
  const unsigned xrowb = (unsigned)a.Fin * 4u, yrowb = (unsigned)a.ld * 4u;
  const float floor_v = a.act == DSPH_ACT_RELU ? 0.f : -__builtin_huge_valf();
  const int nch = a.Fin;  // real channels (multiple of four); this lane's are CH * g .. CH * g + CH - 1
  const bool cheb = __builtin_amdgcn_readfirstlane(a.cheb) != 0;
  const bool all_ch = nch == 2 * CH || pairm;             // every lane's CH channels exist: no masking of the loads
  const int pool = __builtin_amdgcn_readfirstlane(a.pool);  // pooled epilogue: see cheb_istrip1_kernel below

  for (int64_t q = q_begin + slot0 * IS_WAVES + wave; q < q_end; q += (int64_t)nslots * IS_WAVES) {
    const int n = (int)(q % NI);
    const int sg = (int)((q / NI) % a.nseg);
    const int e = (int)((q / ((int64_t)NI * a.nseg)) & 1);
    const int p = (int)(q / (2 * (int64_t)NI * a.nseg));
    StripPair pr = a.pairs[p];
    {
      const int H = pr.y1 - pr.y0, em = pool ? ~1 : ~0;  // (pooled: cuts at even rows)
      const int ya = pr.y0 + ((int)((int64_t)H * sg / a.nseg) & em), yb = pr.y0 + ((int)((int64_t)H * (sg + 1) / a.nseg) & em);
      pr.y0 = ya;
      pr.y1 = yb;
      if (yb <= ya) continue;
    }
    const int x0 = e ? pr.x0[1] : pr.x0[0], wuse = e ? pr.w[1] : pr.w[0], xs = e ? pr.xs[1] : pr.xs[0];
    if (wuse <= 0) continue;
    const unsigned sX = st_spread((unsigned)min(max(xs + px, pr.xlo), pr.xhi));
    const unsigned sXs = st_spread((unsigned)(xs + (lane >> 3)));  // the pixel whose y chunk this lane stores (instruction 0)
    const int pfirst = x0 - xs, plast = x0 - xs + wuse;
    // (pairs: this half's own map -- the last map again where an odd batch has no partner, its stores masked; chunks 0 .. 3 of a
    // pixel's y are map 2 n's, 4 .. 7 map 2 n + 1's)
    const int n_x = pairm ? min(2 * n + g, a.N - 1) : n;
    const int n_y = pairm ? 2 * n + ((lane >> 2) & 1) : n;
    const bool y_live = n_y < a.N;
    const char* __restrict__ xmap = reinterpret_cast<const char*>(a.x) + (size_t)n_x * a.x_rows * xrowb;
    char* __restrict__ ymap = pool ? reinterpret_cast<char*>(a.ypool) + (size_t)(y_live ? n_y : 0) * a.ypool_rows * yrowb
                                   : reinterpret_cast<char*>(a.y) + (size_t)(y_live ? n_y : 0) * a.y_rows * yrowb;
    auto spread_y = [&](int yrow) __attribute__((always_inline)) -> unsigned {
      return st_spread((unsigned)min(max(yrow, pr.ylo), pr.yhi)) << 1;
    };

    float P[NP > 0 ? NP : 1][3][CH];
    sp_f32x16 Y[3];
#pragma unroll
    for (int k = 0; k < NP; ++k)
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int c = 0; c < CH; ++c) P[k][s][c] = 0.f;
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int c = 0; c < 16; ++c) Y[s][c] = 0.f;

    // the row of x being fetched (this lane's CH channels) and the row of L~ being fetched
    float xin[CH];
    sp_f32x4 cv;
    float cd;
    auto xfetch = [&](int yrow) __attribute__((always_inline)) {
      const char* row = xmap + (size_t)(sX | spread_y(yrow)) * xrowb;
#pragma unroll
      for (int c4 = 0; c4 < CH / 4; ++c4) {
        // (branch-free: a lane whose channels the layer does not have reads the row's first quad and drops it)
        const int ch = (pairm ? 0 : CH * g) + 4 * c4;
        const bool have = all_ch || ch < nch;
        const sp_f32x4 v = *reinterpret_cast<const sp_f32x4*>(row + (have ? ch : 0) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) xin[4 * c4 + j] = have ? v[j] : 0.f;
      }
    };
    auto cfetch = [&](int yrow) __attribute__((always_inline)) {
      const unsigned rid = sX | spread_y(yrow);
      cv = *reinterpret_cast<const sp_f32x4*>(reinterpret_cast<const char*>(a.gvals8) + (size_t)rid * 32u + (unsigned)g * 16u);
      cd = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.gdiag) + (size_t)rid * 4u);
    };
    auto cstore = [&](int slot) __attribute__((always_inline)) {
      unsigned char* pr_ = cring + (unsigned)slot * IS_CROWB;
      *reinterpret_cast<sp_f32x4*>(pr_ + (unsigned)g * 512u + (unsigned)px * 16u) = cv;  // (16-byte stride: no bank conflicts)
      if (g == 0) *reinterpret_cast<float*>(pr_ + 1024 + (unsigned)px * 4u) = cd;
    };
    auto c9 = [&](int slot) __attribute__((always_inline)) -> SpCoef {
      const unsigned char* pr_ = cring + (unsigned)slot * IS_CROWB;
      SpCoef c;
      c.a = *reinterpret_cast<const sp_f32x4*>(pr_ + (unsigned)px * 16u);
      c.b = *reinterpret_cast<const sp_f32x4*>(pr_ + 512u + (unsigned)px * 16u);
      c.d = *reinterpret_cast<const float*>(pr_ + 1024 + (unsigned)px * 4u);
      return c;
    };

    // steps t = 0 .. : row ytop = y0 - D + t of x arrives; ring slot of the row with step index u is u mod 3 (planes and
    // accumulators), u mod IS_CRING (rows of L~).  Prologue: the rows of x and L~ of step 0.
    const int ybase = pr.y0 - D;
    const int T3 = ((pr.y1 - pr.y0) + 2 * D + 2) / 3;  // steps, in threes: output row yr leaves at step yr + D - ybase = yr - y0 + 2 D
    xfetch(ybase);
    cfetch(ybase);
    int cs = 0;  // L~ ring slot of row ybase + t
    auto step = [&](auto ph_c, int t) __attribute__((always_inline)) {
      constexpr int PH = decltype(ph_c)::value;  // t mod 3
      const int ytop = ybase + t;
      // this step's rows land: x -> T_0[new], the row of L~ -> its ring slot; the next ones go out
#pragma unroll
      for (int c = 0; c < CH; ++c) P[0][PH][c] = xin[c];
      cstore(cs);
      xfetch(ytop + 1);
      cfetch(ytop + 1);
      __builtin_amdgcn_wave_barrier();  // (the ring row just written is read below by other lanes of this wave)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

      // levels 1 .. K-1: T_k[ytop - k] from the rows ytop-k-1, ytop-k, ytop-k+1 of T_{k-1}
      float top[CH];  // T_{K-1}[ytop - (K-1)]: never stored
#pragma unroll
      for (int k = 1; k <= K - 1; ++k) {
        constexpr int dummy = 0; (void)dummy;
        // ring slots: row with step index u sits in slot u mod 3; T_{k-1}'s newest row has index t - (k-1)
        const int s_new = ((PH - (k - 1)) % 3 + 3) % 3, s_mid = (s_new + 2) % 3, s_old = (s_new + 1) % 3;
        int cslot = cs - k;
        cslot += cslot < 0 ? IS_CRING : 0;
        const SpCoef cf = c9(cslot);
        float acc[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = 0.f;
        is_row<CH>(acc, P[k - 1][s_old], cf.b[3], cf.b[2], cf.b[1]);  // y-1: directions 7, 6, 5
        is_row<CH>(acc, P[k - 1][s_mid], cf.a[0], cf.d, cf.b[0]);     // y  : 0, diagonal, 4
        is_row<CH>(acc, P[k - 1][s_new], cf.a[1], cf.a[2], cf.a[3]);  // y+1: 1, 2, 3
        if (k >= 2 && cheb) {  // (wave-uniform)
          // T_k = 2 (L~ T_{k-1}) - T_{k-2}[ytop - k]: the oldest row of T_{k-2}'s ring (index t - k = (t - (k-2)) - 2)
          const int s2 = (((PH - (k - 2)) % 3 + 3) % 3 + 1) % 3;
#pragma unroll
          for (int c = 0; c < CH; ++c) acc[c] = fmaf(2.f, acc[c], -P[k - 2][s2][c]);
        }
        is_fence<CH>(acc);
        if (k <= K - 2) {
          const int s_k = ((PH - k) % 3 + 3) % 3;  // T_k's newest row, index t - k
#pragma unroll
          for (int c = 0; c < CH; ++c) P[k][s_k][c] = acc[c];
        } else {
#pragma unroll
          for (int c = 0; c < CH; ++c) top[c] = acc[c];
        }
      }
      // contraction.  Row r = ytop - G is opened with levels 0 .. G; every level k > G adds to row ytop - k; row ytop - (K-1) is
      // complete.  Accumulator slot of the row with step index u: u mod 3.
      {
        const int sy = ((PH - G) % 3 + 3) % 3;
#pragma unroll
        for (int k = 0; k <= G; ++k) {
          // T_k[ytop - G]: index t - G in T_k's ring (k <= K-2), or the transient top row (k == K-1 == G)
          if (k <= K - 2) {
            const int s = ((PH - G) % 3 + 3) % 3;
            if (k == 0) is_contract<CH, PREC, true>(Y[sy], P[k][s], sW + k * WLEVB, lane);
            else is_contract<CH, PREC, false>(Y[sy], P[k][s], sW + k * WLEVB, lane);
          } else {
            is_contract<CH, PREC, false>(Y[sy], top, sW + k * WLEVB, lane);
          }
        }
#pragma unroll
        for (int k = G + 1; k <= K - 1; ++k) {
          const int syk = ((PH - k) % 3 + 3) % 3;
          if (k <= K - 2) is_contract<CH, PREC, false>(Y[syk], P[k][((PH - k) % 3 + 3) % 3], sW + k * WLEVB, lane);
          else is_contract<CH, PREC, false>(Y[syk], top, sW + k * WLEVB, lane);
        }
      }
      // y of row ytop - (K-1): through the staging block so that eight lanes store 128 contiguous bytes (cheb_strip_kernel.h)
      {
        const int yr = ytop - D;
        const sp_f32x16& Yd = Y[((PH - D) % 3 + 3) % 3];
        auto run_base = [](unsigned run) -> unsigned { return run * 256u; };
        {
          const unsigned run = (unsigned)px >> 1;
          unsigned char* wp = yst + (pool && (yr & 1) ? IS_YSTB : 0) + run_base(run) + ((unsigned)px & 1u) * 128u;
#pragma unroll
          for (int tq = 0; tq < 4; ++tq)
            *reinterpret_cast<sp_f32x4*>(wp + (((unsigned)(2 * tq + g)) ^ (run & 7u)) * 16u) =
                sp_f32x4{Yd[4 * tq], Yd[4 * tq + 1], Yd[4 * tq + 2], Yd[4 * tq + 3]};
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int och = 4 * (pairm ? lane & 3 : lane & 7);
        if (pool == 0 && yr >= pr.y0 && yr < pr.y1) {  // (wave-uniform)
          const unsigned sY = st_spread((unsigned)yr) << 1;
          const sp_f32x4 bv = *reinterpret_cast<const sp_f32x4*>(sBias + och);
          sp_f32x4 yo4[4];
#pragma unroll
          for (int k4 = 0; k4 < 4; ++k4) {
            const unsigned pk = 8u * k4 + ((unsigned)lane >> 3), run = pk >> 1;
            yo4[k4] = *reinterpret_cast<const sp_f32x4*>(yst + run_base(run) + (pk & 1u) * 128u + ((((unsigned)lane & 7u)) ^ (run & 7u)) * 16u);
          }
#pragma unroll
          for (int k4 = 0; k4 < 4; ++k4) {
            const int pk = 8 * k4 + (lane >> 3);
            const unsigned rid = (((sXs | 0xAAAAAAAAu) + st_spread(8u * k4)) & 0x55555555u) | sY;
            float* dst = reinterpret_cast<float*>(ymap + (size_t)rid * yrowb) + och;
            sp_f32x4 o;
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) o[e4] = fmaxf(yo4[k4][e4] + bv[e4], floor_v);
            // (the launch guarantees 16-byte stores: the block's width and the row stride of y are multiples of four)
            if (pk >= pfirst && pk < plast && och < a.Fout && y_live) *reinterpret_cast<sp_f32x4*>(dst) = o;
          }
        } else if (pool != 0 && (yr & 1) && yr >= pr.y0 && yr < pr.y1) {
          // pooled epilogue (cheb_istrip1_kernel below): rows yr - 1 (block 0) and yr (block 1), the pixel pairs with an even x
          const sp_f32x4 bv = *reinterpret_cast<const sp_f32x4*>(sBias + och);
          const unsigned sYp = st_spread((unsigned)yr >> 1) << 1, cpl = (unsigned)lane & 7u, pl = (unsigned)lane >> 3;
#pragma unroll
          for (int k2 = 0; k2 < 2; ++k2) {
            const unsigned pp = 8u * k2 + pl, pk0u = 2u * pp + ((unsigned)xs & 1u), pk1u = min(pk0u + 1u, 31u);
            const unsigned off0 = run_base(pk0u >> 1) + (pk0u & 1u) * 128u + ((cpl ^ ((pk0u >> 1) & 7u))) * 16u;
            const unsigned off1 = run_base(pk1u >> 1) + (pk1u & 1u) * 128u + ((cpl ^ ((pk1u >> 1) & 7u))) * 16u;
            sp_f32x4 c[4];
            c[0] = *reinterpret_cast<const sp_f32x4*>(yst + off0);
            c[1] = *reinterpret_cast<const sp_f32x4*>(yst + off1);
            c[2] = *reinterpret_cast<const sp_f32x4*>(yst + IS_YSTB + off0);
            c[3] = *reinterpret_cast<const sp_f32x4*>(yst + IS_YSTB + off1);
            sp_f32x4 o;
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
              float r = pool == 1 ? -__builtin_huge_valf() : 0.f;
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const float v = fmaxf(c[q][e4] + bv[e4], floor_v);
                r = pool == 1 ? fmaxf(r, v) : r + v;
              }
              o[e4] = pool == 1 ? r : r * 0.25f;
            }
            const unsigned rid = st_spread(((unsigned)xs + pk0u) >> 1) | sYp;
            float* dst = reinterpret_cast<float*>(ymap + (size_t)rid * yrowb) + och;
            if ((int)pk0u >= pfirst && (int)pk0u + 1 < plast && och < a.Fout && y_live) *reinterpret_cast<sp_f32x4*>(dst) = o;
          }
        }
        __builtin_amdgcn_wave_barrier();  // (the block is rewritten in the next step)
      }
      cs = cs + 1 == IS_CRING ? 0 : cs + 1;
    };
    for (int t3 = 0; t3 < T3; ++t3) {
      step(std::integral_constant<int, 0>{}, 3 * t3);
      step(std::integral_constant<int, 1>{}, 3 * t3 + 1);
      step(std::integral_constant<int, 2>{}, 3 * t3 + 2);
    }
  }
}

// ---- one or two input channels: the levels go into the MFMA's inner index ---------------------------------------------------
// The first layer of every reference model has ONE input channel (tests/test_healpy_networks.py:96; BASELINE configs[0]), and
// zero-padded to the four-channel form above it spends three quarters of its recurrence and four of its five contractions on
// zeros.  Here a lane is (pixel column, channel g = lane >> 5) with ONE value per plane row, and the contraction's inner index
// carries the LEVELS: slot j of half g <- T_j[r] of channel g, so  y[r] = sum_k T_k[r] W_k  is one MFMA group per row instead
// of one per level.  That needs T_0[r] .. T_{K-1}[r] at the same time: level k keeps its last max(3, K - k) rows (one register
// each; shifted, not rotated -- no phase unrolling), T_k[r] made in step r + k and used in step r + K - 1.  One accumulator
// row, 16 registers.  With so little plane state the rows of L~ live in registers too (K rows of nine values, shifted like
// the planes; each half of the wave fetches four of a pixel's eight directions and v_permlane32_swap hands both halves both
// quads), and so does the ONE weight image: the LDS holds the y block only, which a step fills and the NEXT step turns and
// stores (no LDS round trip inside a step).  Strips, segments and item dealing are the kernel above's.  Two steps are in
// flight on the load side (rows ytop + 1 and ytop + 2), two phases unrolled for that.  Workgroups of four waves, three per
// CU (140 registers).
constexpr int IS1_WAVES = 4;
constexpr int IS1_THREADS = 64 * IS1_WAVES;
constexpr int IS1_WG_PER_CU = 3;  // workgroups of four waves per CU: three waves per SIMD at up to 168 registers

template <int K, int PREC>
__global__ __launch_bounds__(IS1_THREADS, 3) void cheb_istrip1_kernel(IStripArgs a) {
  constexpr int D = K - 1;
  constexpr int NP = K - 1;            // planes with rows kept: T_0 .. T_{K-2}
  constexpr int WAVEB = 2 * IS_YSTB;   // LDS per wave: two y blocks (even and odd rows: the pooled epilogue needs both; the rows of L~ live in registers)
  __shared__ __attribute__((aligned(16))) unsigned char smem[IS1_WAVES * WAVEB + 128];
  float* const sBias = reinterpret_cast<float*>(smem + IS1_WAVES * WAVEB);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = lane & 31, g = lane >> 5;
  unsigned char* const yst = smem + wave * WAVEB;

  // the ONE weight image stays in registers: this lane's A-operand fragments (12 registers; K under the exact arithmetic)
  sp_bf16x8 wf[3];
  float wx[K];
  if (PREC == DSPH_PREC_FP32) {
#pragma unroll
    for (int s_ = 0; s_ < K; ++s_) wx[s_] = *reinterpret_cast<const float*>(a.wimg + s_ * 256 + lane * 4);
  } else {
#pragma unroll
    for (int t_ = 0; t_ < is_terms(PREC); ++t_) wf[t_] = *reinterpret_cast<const sp_bf16x8*>(a.wimg + t_ * 1024 + lane * 16);
  }
  if (tid < 32) sBias[tid] = (a.bias != nullptr && tid < a.Fout) ? a.bias[tid] : 0.f;
  __syncthreads();

  const int G_ = gridDim.x, xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3;
  const int nslots = (G_ + 7 - xcd) / 8;
  // Two maps per wave (a.pair: ONE input channel and at most 16 output columns, the first layer of a network): half g of the
  // wave runs the recurrence of map 2 n + g -- its lanes would carry a channel that does not exist --, the rows of L~ fetched once
  // serve both, and ONE MFMA group contracts both: rows 0 .. 15 of the weight image hold W in the inner slots of half 0, rows
  // 16 .. 31 hold it in the slots of half 1, so accumulator rows 0 .. 15 are map 2 n's 16 columns and rows 16 .. 31 map 2 n + 1's.
  // The y block then holds [map 2 n | map 2 n + 1] per pixel and the eight lanes of a pixel store 64 bytes into each.
  const bool pairm = __builtin_amdgcn_readfirstlane(a.pair) != 0;
  const int NI = pairm ? (a.N + 1) / 2 : a.N;  // maps, or pairs of maps, per strip segment
  const int64_t n_items = (int64_t)a.npairs * 2 * a.nseg * NI;
  const int64_t q_begin = n_items * xcd / 8, q_end = n_items * (xcd + 1) / 8;
  const unsigned xrowb = (unsigned)a.Fin * 4u, yrowb = (unsigned)a.ld * 4u;
  const float floor_v = a.act == DSPH_ACT_RELU ? 0.f : -__builtin_huge_valf();
  const bool cheb = __builtin_amdgcn_readfirstlane(a.cheb) != 0;
  // at most 16 output columns (a first layer: 1 -> 16): a y row is 64 bytes, FOUR lanes store it, and two store instructions with
  // every lane at work cover the strip row -- instead of four with half of their lanes masked
  // Pooled epilogue (a.pool: the layer is followed by HealpyPool(p = 1), reference healpy_layers.py:20-63): the four children of a
  // coarse pixel are the pixels (2 i, 2 j), (2 i + 1, 2 j), (2 i, 2 j + 1), (2 i + 1, 2 j + 1) -- two neighbouring pixels of two
  // consecutive strip rows, i.e. the two 128-byte halves of one run of the y block, of the even row's block and of the odd row's.
  // The odd row's store step reads the four, applies bias and activation to each, reduces them in child order and stores ONE row of
  // the pooled map; the full-resolution y of these pixels never exists.  (Segments are then cut at even rows.)
  const int pool = __builtin_amdgcn_readfirstlane(a.pool);
  const bool nout = __builtin_amdgcn_readfirstlane(a.Fout) <= 16 && !pairm && pool == 0;

  for (int64_t q = q_begin + slot0 * IS1_WAVES + wave; q < q_end; q += (int64_t)nslots * IS1_WAVES) {
    const int n = (int)(q % NI);
    const int sg = (int)((q / NI) % a.nseg);
    const int e = (int)((q / ((int64_t)NI * a.nseg)) & 1);
    const int p = (int)(q / (2 * (int64_t)NI * a.nseg));
    StripPair pr = a.pairs[p];
    {
      const int H = pr.y1 - pr.y0, em = pool ? ~1 : ~0;  // (pooled: cuts at even rows)
      const int ya = pr.y0 + ((int)((int64_t)H * sg / a.nseg) & em), yb = pr.y0 + ((int)((int64_t)H * (sg + 1) / a.nseg) & em);
      pr.y0 = ya;
      pr.y1 = yb;
      if (yb <= ya) continue;
    }
    const int x0 = e ? pr.x0[1] : pr.x0[0], wuse = e ? pr.w[1] : pr.w[0], xs = e ? pr.xs[1] : pr.xs[0];
    if (wuse <= 0) continue;
    const unsigned sX = st_spread((unsigned)min(max(xs + px, pr.xlo), pr.xhi));
    const unsigned sXs = st_spread((unsigned)(xs + (nout ? lane >> 2 : lane >> 3)));
    const int pfirst = x0 - xs, plast = x0 - xs + wuse;
    // (x arrives zero-padded to four channels: channel g of a one-channel layer reads the padding)
    // (pairs: this half's own map, the last map again where an odd batch has no partner -- its stores are masked)
    const int n_x = pairm ? min(2 * n + g, a.N - 1) : n;
    const char* __restrict__ xmap = reinterpret_cast<const char*>(a.x) + (size_t)n_x * a.x_rows * xrowb + (pairm ? 0u : (unsigned)g * 4u);
    const int n_y = pairm ? 2 * n + ((lane >> 2) & 1) : n;  // (pairs: chunks 0 .. 3 of a pixel are map 2 n's, 4 .. 7 map 2 n + 1's)
    const bool y_live = n_y < a.N;
    char* __restrict__ ymap = pool ? reinterpret_cast<char*>(a.ypool) + (size_t)(y_live ? n_y : 0) * a.ypool_rows * yrowb
                                   : reinterpret_cast<char*>(a.y) + (size_t)(y_live ? n_y : 0) * a.y_rows * yrowb;
    auto spread_y = [&](int yrow) __attribute__((always_inline)) -> unsigned {
      return st_spread((unsigned)min(max(yrow, pr.ylo), pr.yhi)) << 1;
    };

    // R[k][d]: row (newest - d) of T_k; level k keeps max(3, K - k) rows
    float R[NP][K < 3 ? 3 : K];
#pragma unroll
    for (int k = 0; k < NP; ++k)
#pragma unroll
      for (int d = 0; d < (K < 3 ? 3 : K); ++d) R[k][d] = 0.f;

    float xin[2];
    sp_f32x4 cv[2];
    float cd[2];
    auto fetch = [&](auto b_c, int yrow) __attribute__((always_inline)) {
      constexpr int B = decltype(b_c)::value;
      const unsigned rid = sX | spread_y(yrow);
      xin[B] = *reinterpret_cast<const float*>(xmap + (size_t)rid * xrowb);
      cv[B] = *reinterpret_cast<const sp_f32x4*>(reinterpret_cast<const char*>(a.gvals8) + (size_t)rid * 32u + (unsigned)g * 16u);
      cd[B] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.gdiag) + (size_t)rid * 4u);
    };
    // CF[j]: the nine values of L~ of this lane's pixel in row ytop - j (level k uses CF[k]); shifted like the planes' rows
    SpCoef CF[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      CF[j].a = sp_f32x4{0.f, 0.f, 0.f, 0.f};
      CF[j].b = sp_f32x4{0.f, 0.f, 0.f, 0.f};
      CF[j].d = 0.f;
    }

    const int ybase = pr.y0 - D;
    const int T2 = ((pr.y1 - pr.y0) + 2 * D + 2) / 2;  // steps, in twos: output row yr is made in step yr - y0 + 2 D, stored in the next
    fetch(std::integral_constant<int, 0>{}, ybase);
    fetch(std::integral_constant<int, 1>{}, ybase + 1);
    auto step = [&](auto ph_c, int t) __attribute__((always_inline)) {
      constexpr int PH = decltype(ph_c)::value;  // t mod 2: which fetch buffer holds this step's rows
      const int ytop = ybase + t;
      // every plane's rows age by one; this step's rows land; the rows of step t + 2 go out
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        constexpr int dummy = 0; (void)dummy;
        const int dep = (K - k) < 3 ? 3 : (K - k);
#pragma unroll
        for (int d = (K < 3 ? 3 : K) - 1; d >= 1; --d)
          if (d < dep) R[k][d] = R[k][d - 1];
      }
      R[0][0] = xin[PH];
      // (the rows are read through DPP below, by instructions the compiler's hazard recogniser cannot see: two wait states
      // behind the moves above)
#pragma unroll
      for (int k = 0; k < NP; ++k) asm volatile("s_nop 1" : "+v"(R[k][0]), "+v"(R[k][1]), "+v"(R[k][2]));
#pragma unroll
      for (int j = K - 1; j >= 1; --j) CF[j] = CF[j - 1];
      // the row of L~ that arrived: half g of the wave fetched directions 4 g .. 4 g + 3 of its pixel; v_permlane32_swap gives both
      // halves both quads (the lower half's register to every lane, and the upper half's)
#pragma unroll
      for (int e4 = 0; e4 < 4; ++e4) {
        // (as asm: the compiler folded the four builtin calls of this loop into one; the no-ops are the wait states a VALU result
        // needs before the swap reads it and the swap's results before a VALU reads them)
        float lo = cv[PH][e4], hi = cv[PH][e4];
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(hi));
        CF[0].a[e4] = lo;
        CF[0].b[e4] = hi;
      }
      CF[0].d = cd[PH];
      fetch(ph_c, ytop + 2);
      __builtin_amdgcn_wave_barrier();  // (the LDS executes a wave's instructions in order: the previous step's block is complete)
      // the y row the PREVIOUS step left in the block
      auto run_base = [](unsigned run) -> unsigned { return run * 256u; };
      // (pixels per store instruction PPI = 8 with eight 16-byte chunks each, or 16 with four; wave-uniform)
      const unsigned ppi = nout ? 16u : 8u, cpl = nout ? (unsigned)lane & 3u : (unsigned)lane & 7u, pl = nout ? (unsigned)lane >> 2 : (unsigned)lane >> 3;
      const int och = 4 * (int)(pairm ? cpl & 3u : cpl);
      const int yr = ytop - D - 1;  // finished by the previous step
      if (pool == 0) {
        const unsigned char* yb_ = yst;
        sp_f32x4 yo4[4];
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
          if (nout && k4 >= 2) break;
          const unsigned pk = ppi * k4 + pl, run = pk >> 1;
          yo4[k4] = *reinterpret_cast<const sp_f32x4*>(yb_ + run_base(run) + (pk & 1u) * 128u + ((cpl ^ (run & 7u))) * 16u);
        }
        __builtin_amdgcn_wave_barrier();
        if (yr >= pr.y0 && yr < pr.y1) {  // (wave-uniform)
          const unsigned sY = st_spread((unsigned)yr) << 1;
          const sp_f32x4 bv = *reinterpret_cast<const sp_f32x4*>(sBias + och);
#pragma unroll
          for (int k4 = 0; k4 < 4; ++k4) {
            if (nout && k4 >= 2) break;
            const int pk = (int)(ppi * k4 + pl);
            const unsigned rid = (((sXs | 0xAAAAAAAAu) + st_spread(ppi * k4)) & 0x55555555u) | sY;
            float* dst = reinterpret_cast<float*>(ymap + (size_t)rid * yrowb) + och;
            sp_f32x4 o;
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) o[e4] = fmaxf(yo4[k4][e4] + bv[e4], floor_v);
            if (pk >= pfirst && pk < plast && och < a.Fout && y_live) *reinterpret_cast<sp_f32x4*>(dst) = o;
          }
        }
      } else if ((yr & 1) && yr >= pr.y0 && yr < pr.y1) {  // (wave-uniform) rows yr - 1 (block 0) and yr (block 1) are both there
        const sp_f32x4 bv = *reinterpret_cast<const sp_f32x4*>(sBias + och);
        const unsigned sYp = st_spread((unsigned)yr >> 1) << 1;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          // pooled pixel pp of the strip: the pixels pk0, pk0 + 1 with an EVEN absolute x (the strip starts K - 1 columns left of
          // its first output column: at an odd x when K is even)
          const unsigned pp = 8u * k2 + pl, pk0u = 2u * pp + ((unsigned)xs & 1u), pk1u = min(pk0u + 1u, 31u);
          const unsigned off0 = run_base(pk0u >> 1) + (pk0u & 1u) * 128u + ((cpl ^ ((pk0u >> 1) & 7u))) * 16u;
          const unsigned off1 = run_base(pk1u >> 1) + (pk1u & 1u) * 128u + ((cpl ^ ((pk1u >> 1) & 7u))) * 16u;
          sp_f32x4 c[4];
          c[0] = *reinterpret_cast<const sp_f32x4*>(yst + off0);
          c[1] = *reinterpret_cast<const sp_f32x4*>(yst + off1);
          c[2] = *reinterpret_cast<const sp_f32x4*>(yst + IS_YSTB + off0);
          c[3] = *reinterpret_cast<const sp_f32x4*>(yst + IS_YSTB + off1);
          sp_f32x4 o;
#pragma unroll
          for (int e4 = 0; e4 < 4; ++e4) {
            float r = pool == 1 ? -__builtin_huge_valf() : 0.f;  // (healpix_pool_kernel's order: the children in row order)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float v = fmaxf(c[q][e4] + bv[e4], floor_v);
              r = pool == 1 ? fmaxf(r, v) : r + v;
            }
            o[e4] = pool == 1 ? r : r * 0.25f;
          }
          const unsigned rid = st_spread(((unsigned)xs + pk0u) >> 1) | sYp;
          float* dst = reinterpret_cast<float*>(ymap + (size_t)rid * yrowb) + och;
          const int pk0 = (int)pk0u;
          if (pk0 >= pfirst && pk0 + 1 < plast && och < a.Fout && y_live) *reinterpret_cast<sp_f32x4*>(dst) = o;
        }
      }
      __builtin_amdgcn_wave_barrier();

      // levels 1 .. K-1: T_k[ytop - k] from T_{k-1}'s three newest rows (ytop-k+1, ytop-k, ytop-k-1) and T_{k-2}[ytop - k]
      float row[8];  // the contraction's operand: slot j <- T_j[ytop - (K-1)]
#pragma unroll
      for (int j = 0; j < 8; ++j) row[j] = 0.f;
#pragma unroll
      for (int k = 1; k <= K - 1; ++k) {
        float acc[1] = {0.f};
        const float s_old[1] = {R[k - 1][2]}, s_mid[1] = {R[k - 1][1]}, s_new[1] = {R[k - 1][0]};
        is_row<1>(acc, s_old, CF[k].b[3], CF[k].b[2], CF[k].b[1]);  // y-1: directions 7, 6, 5
        is_row<1>(acc, s_mid, CF[k].a[0], CF[k].d, CF[k].b[0]);     // y  : 0, diagonal, 4
        is_row<1>(acc, s_new, CF[k].a[1], CF[k].a[2], CF[k].a[3]);  // y+1: 1, 2, 3
        float v = acc[0];
        if (k >= 2 && cheb) v = fmaf(2.f, v, -R[k - 2][2]);
        if (k <= K - 2) {
          R[k][0] = v;
          asm volatile("s_nop 1" : "+v"(R[k][0]));  // (read through DPP by the next level)
        } else {
          row[K - 1] = v;
          asm volatile("s_nop 1" : "+v"(row[K - 1]));  // (an MFMA operand as it is under the exact arithmetic)
        }
      }
#pragma unroll
      for (int j = 0; j <= K - 2; ++j) row[j] = R[j][K - 1 - j];

      sp_f32x16 Yd;
      if (PREC == DSPH_PREC_FP32) {
#pragma unroll
        for (int c = 0; c < 16; ++c) Yd[c] = 0.f;
#pragma unroll
        for (int s = 0; s < K; ++s) Yd = __builtin_amdgcn_mfma_f32_32x32x2f32(wx[s], row[s], Yd, 0, 0, 0);
      } else {
#pragma unroll
        for (int c = 0; c < 16; ++c) Yd[c] = 0.f;
        const IsFrag f = is_split<8, PREC>(row);
        if (PREC == DSPH_PREC_BF16X3) {
          Yd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0], f.t[1], Yd, 0, 0, 0);  // small terms first (is_contract)
          Yd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1], f.t[0], Yd, 0, 0, 0);
          Yd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0], f.t[0], Yd, 0, 0, 0);
        } else {
          Yd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0], f.t[2], Yd, 0, 0, 0);
          Yd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[2], f.t[0], Yd, 0, 0, 0);
          Yd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1], f.t[1], Yd, 0, 0, 0);
          Yd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0], f.t[1], Yd, 0, 0, 0);
          Yd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1], f.t[0], Yd, 0, 0, 0);
          Yd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0], f.t[0], Yd, 0, 0, 0);
        }
      }
      // y of row ytop - (K-1) into the staging block; the NEXT step turns it so that eight lanes store 128 contiguous bytes
      {
        const unsigned run = (unsigned)px >> 1;
        unsigned char* wp = yst + (pool && ((ytop - D) & 1) ? IS_YSTB : 0) + run_base(run) + ((unsigned)px & 1u) * 128u;
#pragma unroll
        for (int tq = 0; tq < 4; ++tq) {
          if (nout && tq >= 2) break;  // (columns 16 .. 31 do not exist)
          *reinterpret_cast<sp_f32x4*>(wp + (((unsigned)(2 * tq + g)) ^ (run & 7u)) * 16u) =
              sp_f32x4{Yd[4 * tq], Yd[4 * tq + 1], Yd[4 * tq + 2], Yd[4 * tq + 3]};
        }
      }
    };
    for (int t2 = 0; t2 < T2; ++t2) {
      step(std::integral_constant<int, 0>{}, 2 * t2);
      step(std::integral_constant<int, 1>{}, 2 * t2 + 1);
    }
  }
}

}  // namespace dsph
