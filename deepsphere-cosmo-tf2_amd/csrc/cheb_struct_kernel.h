// Structured-tile form of the fused Chebyshev forward (round 2).
//
// Same mathematics as cheb_fused_kernel.h (reference gnn_layers.py:131-150): per (tile, map, 16-channel slice)
// the tile of x plus its (K-1)-ring halo is held in LDS, the three-term recurrence runs there, and every plane
// T_k feeds the MFMA accumulators that hold y until it is written once.  What differs is the FORM of the
// recurrence, for tiles whose region is a plain square of a 2-D 9-point stencil ("class R" tiles: the rows of the
// tile and of its K-1 rings decode, as Morton codes, to a (16+2D)^2 square and every non-zero of those rows of L~
// points at one of the 8 surrounding cells or at the row itself; the plan verifies this per row and per tile, it
// is not assumed -- see struct_rows_kernel / struct_tiles_kernel in cheb_struct.hip):
//   * the LDS plane is a 2-D array of 64-byte cells, so the 8 neighbours of a pixel are COMPILE-TIME offsets from
//     the lane's own address: no column table, no per-neighbour address registers (36 VGPRs in the BFS form);
//   * four lanes (one per 16-byte channel slot) own a 2x2 pixel block for the whole tile and read its 4x4 window
//     once per step: 12 ds_read_b128 per 4 outputs (3 per output; 9+1 in the BFS form).  T_{k-1} and T_{k-2} of
//     the lane's own pixels stay in registers, so a step does no read-modify-write through LDS;
//   * L~ values come from a direction-ordered copy of the rows (gvals8 [row][8] + gdiag [row]): 36 B per row read
//     once per tile instead of a 6 B x 9 x region-rows tile-local table;
//   * x arrives by LDS-DMA (global_load_lds_dwordx4) into a third plane one item ahead: no staging registers,
//     no staging stores; the weight fragments of the next slice are streamed the same way (20 KiB x 2);
//   * twelve waves in two roles (three per SIMD, 168 registers each).  Waves 0-7 run the recurrence and nothing
//     else.  Waves 8-11 -- one per SIMD -- contract plane T_{k-1} (MFMA, 64 pixels x 64 columns each) in the same
//     barrier interval in which step k reads it, issue every LDS-DMA piece (a vector-memory instruction holds the
//     issuing wave for hundreds of cycles), and store y.  The last plane of an item is contracted, and a finished map
//     stored, under the FIRST step of the next item, so nothing but the recurrence is on the critical path;
//   * class-T tiles (TAB = true): the same kernel for tiles whose region is a stencil square but whose halo rows are not a
//     Morton continuation of the tile's (the next base pixel of the sphere, the halo rows of a sharded plan): the row of
//     every plane cell and the nine values of L~ of every cell come from per-tile tables (embed_tile, cheb_fused.hip);
//   * three contraction arithmetics (st_contract): exact-fp32 MFMA, the three-term split-bf16 form, and the six-term
//     fp32-equivalent split whose 3 KiB weight blocks are replaced in place in LDS (DSPH_PREC_BF16X6);
//   * at K = 5 no barrier in front of the item: every LDS-DMA piece is issued at least an interval before the barrier that
//     ends the item, each wave drains vmcnt in front of it;
//   * the contraction is transposed (A = weight fragment, B = plane fragment): the accumulator holds 4 consecutive
//     output channels of ONE pixel per register quad; y goes through a 4 KiB transposition block -- the wave's own
//     64 tile cells of the plane it has just contracted -- so that eight lanes store 128 contiguous bytes.
//
// Cell (gx, gy) of a plane, gx, gy in [0, 24): cell index (gx & 1) * HP + gy * P2 + (gx >> 1) ("column-parity
// split": the 2x2 blocks all have odd origins, so in a plain row-major plane the four blocks read together by a
// 16-lane ds_read_b128 group could never fall into four different bank quarters); the 16-byte slots of a cell are
// XOR-ed with f = (gx & 1) | ((gy & 1) << 1), which makes the MFMA operand read (16 lanes, one logical slot)
// conflict-free and costs the gather three extra base registers (f is a compile-time constant per window cell).
#pragma once

#include <type_traits>

#include "cheb_struct_tables.h"
#include "dsphere_common.h"

namespace dsph {

// Tuning builds only (tools/ab_ablate.sh; the results are wrong by construction, the shipped library defines neither):
// -DDSPH_ST_ABL_DIRS=n sums n of the 8 directions, -DDSPH_ST_ABL_READS=1 reads only window row 0 from LDS.
#ifdef DSPH_ST_ABL_DIRS
#define ST_ABL_DIRS DSPH_ST_ABL_DIRS
#else
#define ST_ABL_DIRS 8
#endif
#ifdef DSPH_ST_ABL_READS
#define ST_ABL_READS 1
#else
#define ST_ABL_READS 0
#endif
// -DDSPH_ST_ABL_SKIP=bits: 1 no recurrence step at all, 2 no contraction, 4 no LDS-DMA, 8 no y store
#ifdef DSPH_ST_ABL_SKIP
#define ST_ABL_SKIP DSPH_ST_ABL_SKIP
#else
#define ST_ABL_SKIP 0
#endif


constexpr int ST_TILE = 16;                     // tile side in pixels: 256 consecutive NEST rows
constexpr int ST_DMAX = 4;                      // halo rings held: K <= 5
constexpr int ST_S = ST_TILE + 2 * ST_DMAX;     // plane side, 24 cells
constexpr int ST_P2 = 13;                       // cells per half-row (12 + 1 pad: see tools/gen_struct_tables.py)
constexpr int ST_HP = ST_S * ST_P2;             // cells per column-parity half
constexpr int ST_CELLS = 2 * ST_HP;             // 624 (48 of them padding)
constexpr int ST_PLANE_BYTES = ST_CELLS * 64;   // 39,936
constexpr int ST_GATHER_WAVES = 8;              // waves 0..7: the recurrence
constexpr int ST_CONTRACT_WAVES = 4;            // waves 8..11: contraction, LDS-DMA, y
constexpr int ST_THREADS = 64 * (ST_GATHER_WAVES + ST_CONTRACT_WAVES);  // 768: three waves per SIMD
constexpr int ST_WSLICE_BYTES = 20480;          // weight fragments of one slice: K * NB * 2048 <= 20480
constexpr int ST_LDS_W = 3 * ST_PLANE_BYTES;    // 119,808
constexpr int ST_LDS_BIAS = ST_LDS_W + 2 * ST_WSLICE_BYTES;  // 160,768
constexpr int ST_LDS_ROWS = ST_LDS_BIAS + 256;               // byte offset, inside a map, of the x row of every plane cell
constexpr int ST_LDS_TOTAL = ST_LDS_ROWS + 640 * 4;          // 163,584 of 163,840 (640 entries: the last DMA piece index a
                                                              // contraction wave forms, 39, reads row 639 -- unused, but inside the array)
constexpr int ST_DMA_PIECES = ST_CELLS / 16;    // 39 wave-instructions of 1 KiB fill a plane
constexpr int ST_WBLK3 = 3072;                  // DSPH_PREC_BF16X6: a weight block is 3 KiB (hi | mid | lo)
constexpr int ST_TABV = 12;                     // floats per cell of a class-T tile's value table (9 used: 3 x 16 B)

typedef float st_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 st_bf16x8 __attribute__((ext_vector_type(8)));

struct StructArgs {
  const float* x;
  const float* bias;
  float* y;
  const unsigned char* wfrag;  // [slice c][order k][column block nb][2048 B], see struct_wprep_kernel
  const int32_t* tiles;        // the class-R tiles this launch handles
  const float* gvals8;         // [rows][8] values of L~ by direction (order of kDirX / kDirY below)
  const float* gdiag;          // [rows]    diagonal of L~
  const int32_t* tabrow;       // TAB kernels (class-T tiles): [ntiles][ST_CELLS] row of every plane cell, and
  const float* tabvals;        //   [ntiles][ST_CELLS][ST_TABV] diagonal + eight directions of every cell's row of L~
  int64_t x_rows, y_rows;
  int ntiles, N, Fin, Fout, K, C, act, ld;
  // Several maps per item (pack = P = 4: layers with at most four input channels and at most 16 output columns, a network's first
  // layers; P = 2: eight input channels, at most 32 columns): the four 16-byte slots of a plane cell carry maps P n .. P n + P - 1
  // (4 / P slots each) instead of channels that do not exist, the weight image is block diagonal (the slots of map q against
  // columns (64 / P) q ..: struct_wprep_kernel), the store sends every column group to its own map.  N is then the number of
  // groups, n_maps the batch.  (The same form as FusedArgs::pack, cheb_fused_kernel.h.)
  int pack, n_maps;
  // conv + HealpyPool(p = 1) in one forward (pool = 1 max, 2 mean; see cheb_istrip1_kernel): the store reduces the four NEST
  // children -- a 2 x 2 pixel block of the tile, all in the wave's transposition block -- and writes the pooled map only
  float* ypool;
  int64_t ypool_rows;
  int pool;
#ifdef DSPH_STAMPS
  unsigned long long* stamps;  // diagnostic build only: [8 waves][8 items][32 points] s_memtime values
#endif
};

// Diagnostic build (make STAMPS=1; never the shipped library): s_memtime at the phase boundaries of eight items of
// one workgroup, into a buffer nothing else reads.  Read the shares, not the run time.
#ifdef DSPH_STAMPS
#define ST_STAMP(id)                                                                       \
  do {                                                                                     \
    if (stamp_on) {                                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                   \
      unsigned long long t_;                                                               \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
      __builtin_amdgcn_sched_barrier(0);                                                   \
      if (lane == 0) a.stamps[((size_t)(wave < 8 ? wave : 7) * 8 + (item - 4)) * 32 + (id)] = t_;  \
    }                                                                                      \
  } while (0)
#else
#define ST_STAMP(id)
#endif

// Summation order of a row: the diagonal, then SW, W, NW, N, NE, E, SE, S (dx, dy below) -- the slot order of the
// repo's own grid-stencil producer (deepsphere/healpix.py), so that on those graphs the sums are the unfused
// kernel's bit for bit.  x is the even-bit (ix) coordinate of the NEST index, y the odd-bit one.
__device__ constexpr int kDirX[8] = {-1, -1, 0, 1, 1, 1, 0, -1};
__device__ constexpr int kDirY[8] = {0, 1, 1, 1, 0, -1, -1, -1};

__host__ __device__ __forceinline__ unsigned st_spread(unsigned v) {
  v = (v | (v << 8)) & 0x00FF00FFu;
  v = (v | (v << 4)) & 0x0F0F0F0Fu;
  v = (v | (v << 2)) & 0x33333333u;
  v = (v | (v << 1)) & 0x55555555u;
  return v;
}
__host__ __device__ __forceinline__ unsigned st_compress(unsigned v) {
  v &= 0x55555555u;
  v = (v | (v >> 1)) & 0x33333333u;
  v = (v | (v >> 2)) & 0x0F0F0F0Fu;
  v = (v | (v >> 4)) & 0x00FF00FFu;
  v = (v | (v >> 8)) & 0x0000FFFFu;
  return v;
}
__host__ __device__ __forceinline__ unsigned st_morton(unsigned x, unsigned y) { return st_spread(x) | (st_spread(y) << 1); }

__host__ __device__ constexpr unsigned st_cell_off(unsigned gx, unsigned gy) {
  return ((gx & 1u) * ST_HP + gy * ST_P2 + (gx >> 1)) * 64u;
}
__host__ __device__ constexpr unsigned st_cell_f(unsigned gx, unsigned gy) { return (gx & 1u) | ((gy & 1u) << 1); }
// window cell (wx, wy) in [0,4)^2 of a block whose window's top-left cell is (even, even): offset from that cell
__host__ __device__ constexpr unsigned st_woff(unsigned wx, unsigned wy) {
  return ((wx & 1u) * ST_HP + wy * ST_P2 + (wx >> 1)) * 64u;
}

// 1 KiB of LDS filled by one wave instruction: lane l's 16 bytes land at lds_dst + 16 l.  Written as asm so that
// hipcc neither waits for it nor orders LDS reads behind it (its LDS-DMA bookkeeping is conservative); the waits
// are st_wait_vm() below.  M0 is saved and restored inside the statement (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void st_glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}
// the same with a wave-uniform 64-bit base and a 32-bit byte offset per lane
__device__ __forceinline__ void st_glds16_off(const void* sbase_in, unsigned voff, unsigned lds_dst) {
  // the base is uniform by construction; say so, or the "s" operand is refused whenever hipcc keeps it in a VGPR
  const unsigned long long bits = reinterpret_cast<unsigned long long>(sbase_in);
  // (readfirstlane returns int: through unsigned, or the low half sign-extends into the high one)
  const unsigned long long ubits = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)bits) |
                                   ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(bits >> 32)) << 32);
  const void* sbase = reinterpret_cast<const void*>(ubits);
  if (ST_ABL_SKIP & 4) return;
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}

// One recurrence step for this lane's 2x2 block (gather waves): window from plane `pin` (byte offset), T_k to `pout`.
//   FIRST: T_1 = L~ T_0, the block's own T_0 is read too (16 reads) and kept in `cur`; otherwise the centre of the
//          window is `cur` (T_{k-1} of the lane's own pixels, in registers since the previous step)
//   CHEB : T_k = 2 L~ T_{k-1} - T_{k-2}, with T_{k-2} in `prev`
// T_k replaces T_{k-2} in `prev`: the caller swaps the roles of the two arrays from step to step, nothing is moved.
// `wr`: this lane's block is part of step k; the others compute on whatever their cells hold and store into a pad
// cell of the output plane (`dummy`) instead of branching.
template <bool FIRST, bool CHEB>
__device__ __forceinline__ void st_gather(const unsigned char* __restrict__ smem, unsigned pin, unsigned pout,
                                          const unsigned (&gb)[4], const float (&v)[4][9], float4 (&cur)[4],
                                          float4 (&prev)[4], bool wr, unsigned dummy) {
  if (ST_ABL_SKIP & 1) return;
  unsigned ob[4];
#pragma unroll
  for (int p = 0; p < 4; ++p)
    ob[p] = (wr ? gb[st_cell_f((p & 1) + 1, (p >> 1) + 1)] + st_woff((p & 1) + 1, (p >> 1) + 1) : dummy) + pout;
  float4 W[4][4];
  auto rd = [&](int wx, int wy) {
#if ST_ABL_READS
    if (wy > 0 && !(FIRST && (wy == 1 || wy == 2) && (wx == 1 || wx == 2))) { W[wy][wx] = W[0][wx]; return; }
#endif
    W[wy][wx] = *reinterpret_cast<const float4*>(smem + (gb[st_cell_f(wx, wy)] + pin) + st_woff(wx, wy));
  };
  // The sum of a pixel runs in the unfused kernel's order: the diagonal, then the eight directions.
  auto pixel = [&](int p) {
    const int i = p & 1, j = p >> 1;
    const float4 c = W[j + 1][i + 1];
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    s.x = fmaf(v[p][0], c.x, s.x);
    s.y = fmaf(v[p][0], c.y, s.y);
    s.z = fmaf(v[p][0], c.z, s.z);
    s.w = fmaf(v[p][0], c.w, s.w);
#pragma unroll
    for (int d = 0; d < ST_ABL_DIRS; ++d) {
      const float4 u = W[j + 1 + kDirY[d]][i + 1 + kDirX[d]];
      const float w = v[p][d + 1];
      s.x = fmaf(w, u.x, s.x);
      s.y = fmaf(w, u.y, s.y);
      s.z = fmaf(w, u.z, s.z);
      s.w = fmaf(w, u.w, s.w);
    }
    if (!FIRST && CHEB) {
      const float4 q = prev[p];
      s.x = 2.f * s.x - q.x;
      s.y = 2.f * s.y - q.y;
      s.z = 2.f * s.z - q.z;
      s.w = 2.f * s.w - q.w;
    }
    if (FIRST) cur[p] = c;
    prev[p] = s;
    *reinterpret_cast<float4*>(const_cast<unsigned char*>(smem) + ob[p]) = s;
  };
  // two halves, so that at most 8 + 4 window cells are in registers at a time: the upper pixel row needs window rows
  // 0..2, the lower one rows 1..3 (row 3 is fetched while the upper row is summed)
#pragma unroll
  for (int wy = 0; wy < 3; ++wy)
#pragma unroll
    for (int wx = 0; wx < 4; ++wx) {
      const bool centre = (wx == 1 || wx == 2) && (wy == 1 || wy == 2);
      if (centre && !FIRST) W[wy][wx] = cur[(wy - 1) * 2 + (wx - 1)];
      else rd(wx, wy);
    }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int wx = 0; wx < 4; ++wx) rd(wx, 3);
  pixel(0);
  pixel(1);
  __builtin_amdgcn_sched_barrier(0);
  pixel(2);
  pixel(3);
}

// Plane T_k into the accumulators of a contraction wave: its 64 tile pixels (two 32-pixel blocks, pixel rows
// 4c .. 4c+3 of the tile) x 16 channels against the fragments of (order k, slice) at `wblk`.
// mb[pb]: byte offset inside a plane of this lane's first 16-byte slot (channels 8h .. 8h+3) for pixel block pb; the
// second slot (channels 8h+4 .. 8h+7) is the first one's address with bit 4 flipped (the slots are XOR-swizzled).
template <int NB, int PREC>
__device__ __forceinline__ void st_contract(const unsigned char* __restrict__ smem, unsigned plane, unsigned wblk,
                                            const unsigned (&mb)[2], int lane, st_f32x16 (&acc)[2][NB]) {
  if (ST_ABL_SKIP & 2) return;
  if (PREC == DSPH_PREC_BF16X6) {
    // fp32-equivalent on the bf16 pipe: v = h + m + l with 8 + 8 + 8 mantissa bits, six products (hh, hm, mh, mm, hl, lh).
    // The weights arrive pre-split (struct_wprep_kernel: 3 KiB per block, hi | mid | lo; `wblk`: the order's first block,
    // the second one 3 KiB behind it).  The plane fragment is split here, exactly, by TRUNCATION: h = the top 16 bits of v
    // (a bf16), r = v - h exactly, m = the top 16 bits of r, l = r - m, which has at most 8 significant bits left:
    // v = h + m + l with no rounding anywhere.
    typedef unsigned st_u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
      const float4 a0 = *reinterpret_cast<const float4*>(smem + plane + mb[pb]);
      const float4 a1 = *reinterpret_cast<const float4*>(smem + plane + (mb[pb] ^ 16u));
      const float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      st_u32x4 hp, mp, lp;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned u0 = __float_as_uint(v[2 * j]), u1 = __float_as_uint(v[2 * j + 1]);
        const float r0 = v[2 * j] - __uint_as_float(u0 & 0xffff0000u), r1 = v[2 * j + 1] - __uint_as_float(u1 & 0xffff0000u);
        const unsigned s0 = __float_as_uint(r0), s1 = __float_as_uint(r1);
        const float q0 = r0 - __uint_as_float(s0 & 0xffff0000u), q1 = r1 - __uint_as_float(s1 & 0xffff0000u);
        hp[j] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);  // {hi16(u1), hi16(u0)}
        mp[j] = __builtin_amdgcn_perm(s1, s0, 0x07060302u);
        lp[j] = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
      }
      const st_bf16x8 th = __builtin_bit_cast(st_bf16x8, hp), tm = __builtin_bit_cast(st_bf16x8, mp),
                      tl = __builtin_bit_cast(st_bf16x8, lp);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const unsigned char* __restrict__ wq = smem + wblk + b * ST_WBLK3 + lane * 16;
        const st_bf16x8 wh = *reinterpret_cast<const st_bf16x8*>(wq);
        const st_bf16x8 wm = *reinterpret_cast<const st_bf16x8*>(wq + 1024);
        const st_bf16x8 wl = *reinterpret_cast<const st_bf16x8*>(wq + 2048);
        // small terms first
        acc[pb][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, th, acc[pb][b], 0, 0, 0);
        acc[pb][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, tl, acc[pb][b], 0, 0, 0);
        acc[pb][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, tm, acc[pb][b], 0, 0, 0);
        acc[pb][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, th, acc[pb][b], 0, 0, 0);
        acc[pb][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, tm, acc[pb][b], 0, 0, 0);
        acc[pb][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, th, acc[pb][b], 0, 0, 0);
      }
    }
    return;
  }
  float4 a[2][2];
#pragma unroll
  for (int pb = 0; pb < 2; ++pb) {
    a[pb][0] = *reinterpret_cast<const float4*>(smem + plane + mb[pb]);
    a[pb][1] = *reinterpret_cast<const float4*>(smem + plane + (mb[pb] ^ 16u));
  }
  const unsigned char* __restrict__ wp = smem + wblk + lane * 16;
  if (PREC == DSPH_PREC_BF16X3) {
    st_bf16x8 whi[NB], wlo[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      whi[b] = *reinterpret_cast<const st_bf16x8*>(wp + b * 2048);
      wlo[b] = *reinterpret_cast<const st_bf16x8*>(wp + b * 2048 + 1024);
    }
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
      const float av[8] = {a[pb][0].x, a[pb][0].y, a[pb][0].z, a[pb][0].w, a[pb][1].x, a[pb][1].y, a[pb][1].z, a[pb][1].w};
      st_bf16x8 thi, tlo;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const __bf16 hi = (__bf16)av[j];
        thi[j] = hi;
        tlo[j] = (__bf16)(av[j] - (float)hi);
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        acc[pb][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[b], tlo, acc[pb][b], 0, 0, 0);  // small terms first
        acc[pb][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo[b], thi, acc[pb][b], 0, 0, 0);
        acc[pb][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[b], thi, acc[pb][b], 0, 0, 0);
      }
    }
  } else {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const float4 w0 = *reinterpret_cast<const float4*>(wp + b * 2048);
      const float4 w1 = *reinterpret_cast<const float4*>(wp + b * 2048 + 1024);
      const float wf[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        const float av[8] = {a[pb][0].x, a[pb][0].y, a[pb][0].z, a[pb][0].w, a[pb][1].x, a[pb][1].y, a[pb][1].z, a[pb][1].w};
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[pb][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[t], av[t], acc[pb][b], 0, 0, 0);
      }
    }
  }
}

// y of one map for a contraction wave's 64 pixels.  The accumulators hold, per lane (pixel r of block pb, half h),
// register quad tq of column block b = output channels 32 b + 8 tq + 4 h .. + 3: stored as they stand a wave
// instruction would touch 32 rows x 2 x 16 B (250-500 cycles of issue each).  They go through a 32 x 32 block of LDS
// first -- the wave's own 64 tile cells (4 KiB) of the plane it has just contracted, which nobody else reads any more
// and which no LDS-DMA piece may touch before the next barrier -- so that eight lanes store 128 contiguous bytes.
// The block's 16-byte units are XOR-ed with the pixel's low bits: conflict-free stores and loads without padding.
// Epilogue: bias, then max(., floor) with floor = 0 (ReLU) or -inf (none): the other activations of the ABI are applied
// by a separate elementwise pass (cheb_struct.hip), so this is ONE code path instead of one per activation.
template <int NB>
__device__ __forceinline__ void st_store(const st_f32x16 (&acc)[2][NB], unsigned char* __restrict__ smem, unsigned plane,
                                         int cw, float* __restrict__ ytile, int ld, const float* __restrict__ sBias,
                                         int lane, int Fout, float floor_v, bool vec, int pack = 0, int64_t map_stride = 0,
                                         int maps_left = 4,  // pack = P: ytile is map P n's, maps_left = n_maps - P n
                                         int pool = 0) {     // pool: ytile / map_stride are the POOLED map's (64 rows per tile)
  if (ST_ABL_SKIP & 8) return;
  const unsigned r = lane & 31, h = lane >> 5, j = lane & 7, pq = lane >> 3;
  // byte a of the 4 KiB block lives in chunk a >> 9 (512 B = the 8 cells of one column parity of one tile pixel row)
  auto scr = [&](unsigned a) -> unsigned {
    const unsigned chunk = a >> 9, row = chunk >> 1, par = chunk & 1;
    return plane + ((par * ST_HP + (ST_DMAX + 4 * cw + row) * ST_P2 + ST_DMAX / 2) * 64u) + (a & 511u);
  };
#pragma unroll
  for (int pb = 0; pb < 2; ++pb)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      __builtin_amdgcn_sched_barrier(0);  // one block at a time: overlapping them only costs registers
#pragma unroll
      for (int tq = 0; tq < 4; ++tq)
        *reinterpret_cast<float4*>(smem + scr(r * 128u + 16u * ((2u * tq + h) ^ (r & 7u)))) =
            make_float4(acc[pb][b][4 * tq + 0], acc[pb][b][4 * tq + 1], acc[pb][b][4 * tq + 2], acc[pb][b][4 * tq + 3]);
      const int ch0 = 32 * b + 4 * (int)j;
      const float4 bv = *reinterpret_cast<const float4*>(sBias + ch0);
      const int gsh = pack == 4 ? 4 : 5;                        // log2 of the columns per map
      const int ch = pack ? ch0 & ((1 << gsh) - 1) : ch0;       // column of y
      const bool live = !pack || (ch0 >> gsh) < maps_left;      // (a batch that ends inside the group)
      float* __restrict__ ymap = ytile + (pack && live ? (int64_t)(ch0 >> gsh) * map_stride : 0);
      if (pool) {
        // the block holds the tile's pixel rows 4 cw + 2 pb and + 1 (p = 0 .. 15 and 16 .. 31): pooled pixel (pq, 2 cw + pb) of the
        // pooled tile from the children p = 2 pq, 2 pq + 1, 16 + 2 pq, 17 + 2 pq, in that (row) order
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pool == 1) o = make_float4(-__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf());
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const unsigned p = 2u * pq + (q & 1u) + 16u * (q >> 1);
          const float4 v = *reinterpret_cast<const float4*>(smem + scr(p * 128u + 16u * (j ^ (p & 7u))));
          const float vx = fmaxf(v.x + bv.x, floor_v), vy = fmaxf(v.y + bv.y, floor_v), vz = fmaxf(v.z + bv.z, floor_v), vw = fmaxf(v.w + bv.w, floor_v);
          if (pool == 1) { o.x = fmaxf(o.x, vx); o.y = fmaxf(o.y, vy); o.z = fmaxf(o.z, vz); o.w = fmaxf(o.w, vw); }
          else { o.x += vx; o.y += vy; o.z += vz; o.w += vw; }
        }
        if (pool != 1) { o.x *= 0.25f; o.y *= 0.25f; o.z *= 0.25f; o.w *= 0.25f; }
        float* __restrict__ yp = ymap + (int64_t)st_morton(pq, 2u * cw + pb) * ld + ch;
        if (ch < Fout && live) *reinterpret_cast<float4*>(yp) = o;  // (the launch guarantees 16-byte stores when pooling)
        continue;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned p = pq + 8 * i;  // pixel (p & 15, 4 cw + 2 pb + (p >> 4)) of the tile
        float4 o = *reinterpret_cast<const float4*>(smem + scr(p * 128u + 16u * (j ^ (p & 7u))));
        o.x = fmaxf(o.x + bv.x, floor_v);
        o.y = fmaxf(o.y + bv.y, floor_v);
        o.z = fmaxf(o.z + bv.z, floor_v);
        o.w = fmaxf(o.w + bv.w, floor_v);
        float* __restrict__ yp = ymap + (int64_t)st_morton(p & 15u, 4u * cw + 2u * pb + (p >> 4)) * ld + ch;
        if (vec) {
          if (ch < Fout && live) *reinterpret_cast<float4*>(yp) = o;
        } else if (live) {
          if (ch + 0 < Fout) yp[0] = o.x;
          if (ch + 1 < Fout) yp[1] = o.y;
          if (ch + 2 < Fout) yp[2] = o.z;
          if (ch + 3 < Fout) yp[3] = o.w;
        }
      }
    }
}

// x piece `piece` (1 KiB: cells 16 piece .. 16 piece + 15 of the plane) as lane `lane` sees it: bits 0..1 the logical
// 16-byte slot it fetches (the stored one is XOR-swizzled), bit 2 set when the cell exists and lies inside the D-ring region.
__device__ __forceinline__ unsigned st_piece_info(int piece, int lane, int D) {
  const unsigned p = 16u * piece + (lane >> 2);
  const unsigned par = p / ST_HP, rem = p % ST_HP, gxh = rem % ST_P2, gy = rem / ST_P2, gx = 2 * gxh + par;
  const int lo = ST_DMAX - D, hi = ST_DMAX + ST_TILE - 1 + D;
  const bool ok = piece < ST_DMA_PIECES && gxh < ST_S / 2 && (int)gx >= lo && (int)gx <= hi && (int)gy >= lo && (int)gy <= hi;
  return ((lane & 3u) ^ st_cell_f(gx, gy)) | (ok ? 4u : 0u);
}
// The 39 x pieces and 20 weight pieces of an item are issued by both roles (a piece holds the issuing wave for up to
// 250-300 cycles): x pieces 0 .. GX-1 and weight pieces 0 .. GW-1 by the eight recurrence waves, the rest by the four
// contraction waves.
// (Measured, same box, split-bf16 / exact-fp32 ms at the headline config: x 16 + w 0 on the recurrence waves 15.44 /
// 23.26; x 0 + w 20: 15.38 / 23.11; x 8 + w 8: 15.36 / 23.32; x 8 + w 20: 15.19 / 22.99 -- the split hardly matters,
// the kernel is bound by the recurrence waves' instruction issue, not by the pieces.)
template <int PREC> struct StPieces {
#if defined(DSPH_ST_GX) && defined(DSPH_ST_GW)
  static constexpr int GX = DSPH_ST_GX, GW = DSPH_ST_GW;  // (tuning build)
#else
  static constexpr int GX = 8;   // x pieces of the recurrence waves: piece w (one per wave)
  static constexpr int GW = 20;  // weight pieces of the recurrence waves: w + 8 u (contiguous 1 KiB: cheap to issue)
#endif
};

// TAB: the rows of the plane cells and their values of L~ come from per-tile tables (class-T tiles: the region is a
// stencil square, but the row numbers of its halo are not a Morton continuation of the tile's) instead of Morton
// arithmetic + gvals8 / gdiag (class R).  Everything else is the same code.
// PX: the forward packs maps (StructArgs::pack) or pools in its store (StructArgs::pool); without it neither path is in the code
// (as run-time branches they cost the kernel, which lives on 168 registers, up to 70 spilled ones)
template <int NB, int PREC, bool CHEB, bool TAB, bool PX = false>
__global__ __launch_bounds__(ST_THREADS, 3) void cheb_struct_kernel(StructArgs a) {
  const int pack_ = PX ? a.pack : 0, pool_ = PX ? a.pool : 0;
  __shared__ __attribute__((aligned(16))) unsigned char smem[ST_LDS_TOTAL];
  float* const sBias = reinterpret_cast<float*>(smem + ST_LDS_BIAS);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: scalar registers, "s" asm operands
  const bool gather = wave < ST_GATHER_WAVES;
  const int cw = wave - ST_GATHER_WAVES;  // contraction wave 0..3
  const int D = a.K - 1;
  if (tid < 64) {
    const int bc = pack_ == 4 ? tid & 15 : (pack_ == 2 ? tid & 31 : tid);  // (packed maps: every column group carries the layer's columns)
    sBias[tid] = (a.bias != nullptr && bc < a.Fout) ? a.bias[bc] : 0.f;
  }
  const unsigned map_bytes = (unsigned)(a.x_rows * a.Fin * 4);  // (pack only: the launch makes sure three of them fit 32 bits)

  // tiles are dealt to XCDs in contiguous ranges (blocks b and b+8 share an XCD and its L2)
  const int G = gridDim.x, xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3;
  const int nslots = (G + 7 - xcd) / 8;
  const int t_begin = (int)((int64_t)a.ntiles * xcd / 8), t_end = (int)((int64_t)a.ntiles * (xcd + 1) / 8);
  // (small maps have fewer tiles than the device has CUs: the maps of the batch are then split over gridDim.y workgroups)
  const int n_first = (int)((int64_t)a.N * blockIdx.y / gridDim.y), n_end = (int)((int64_t)a.N * (blockIdx.y + 1) / gridDim.y);
  if (n_first >= n_end) return;
  const int items = (n_end - n_first) * a.C;
  constexpr unsigned py = ST_PLANE_BYTES;  // the Y plane; X alternates between planes 0 and 2
  if (t_begin + slot0 >= t_end) return;

  if (gather) {
    // =================================================================================================================
    // waves 0..7: the recurrence.  Quad -> 2x2 block (odd origin), lane -> 16-byte slot.
    // =================================================================================================================
#ifdef DSPH_ST_GPRIO
    if (wave >= 4) __builtin_amdgcn_s_setprio(DSPH_ST_GPRIO);  // the younger recurrence wave of each SIMD
#endif
    const int gtid = tid;  // 0..511
    const unsigned blk = kStructBlock[gtid >> 2];
    const bool has_blk = blk != 0xffu;
    const int bx = has_blk ? (int)(blk & 15u) : 0, by = has_blk ? (int)(blk >> 4) : 0;
    const unsigned q = gtid & 3;
    unsigned gb[4];  // byte offset of the window's top-left cell (2bx, 2by), slot q ^ f, f = 0..3
#pragma unroll
    for (unsigned f = 0; f < 4; ++f) gb[f] = st_cell_off(2 * bx, 2 * by) + 16u * (q ^ f);
    // block b covers cells 2b+1, 2b+2; step j computes the region [4-(D-j), 19+(D-j)]
    unsigned lact = 0, wact = 0;  // bit k: this lane's block / some block of this wave is part of step k
#pragma unroll
    for (int k = 1; k <= ST_DMAX; ++k) {
      const int lo = (ST_DMAX - (D - k) - 1) >> 1, hi = (ST_DMAX + ST_TILE - 1 + (D - k) - 1) >> 1;
      const bool on = k <= D && has_blk && bx >= lo && bx <= hi && by >= lo && by <= hi;
      lact |= on ? (1u << k) : 0u;
      wact |= __builtin_amdgcn_ballot_w64(on) != 0 ? (1u << k) : 0u;
    }
    // where the other lanes' stores go: the pad cell (half-row index 12) of their window's second row, own slot
    const unsigned dummy = (unsigned)((2 * by + 1) * ST_P2 + ST_S / 2) * 64u + 16u * q;

    using std::integral_constant;
    // this wave's x pieces of the NEXT item: w + 8 s (< GX), s = 0..4, spread over the intervals 2, 3, 4
    constexpr int GX = StPieces<PREC>::GX, GW = StPieces<PREC>::GW;
    unsigned ginfo = 0;
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      const int piece = wave + 8 * s;
      ginfo |= (piece < GX ? st_piece_info(piece, lane, D) : 0u) << (3 * s);
    }
    const int wblkG = PREC == DSPH_PREC_BF16X6 ? ST_WBLK3 : 2048;
    const int wsliceG = a.K * NB * wblkG;
    // DSPH_PREC_BF16X6 with 10 blocks of 3 KiB per slice (K = 5, 64 columns): two buffers do not fit the 40 KiB weight
    // area, so the orders are replaced IN PLACE as soon as they have been contracted -- order k of an item in that item's
    // interval k + 1, the last one in the next item's interval 1 -- and only order 3, whose slot frees last, has a second
    // slot (items alternate).  Slot of (order k, block b) of the workgroup's item number `it`: 2k + b, or 10 + b for
    // order 3 of an odd item.  What goes out when (all of it in flight before the barrier that ends the item):
    //   interval 2: order 4 of THIS item (its slot held the previous item's last order until interval 1), order 0 of the next
    //   interval 3: orders 1 and 3 of the next item          interval 4 (at its start): order 2 of the next item
    // Smaller slices (K * NB <= 6) are double-buffered like the other precisions.
    const bool stag = PREC == DSPH_PREC_BF16X6 && a.K * NB > 6;  // (the host admits only K = 5, NB = 2 here)
    int itG = 0;  // this workgroup's item counter
    auto worder = [&](int k, int it_t, int c_t, int rot) __attribute__((always_inline)) {
      const unsigned slot0 = (k == 3 && (it_t & 1)) ? 10u : (unsigned)(2 * k);
#pragma unroll
      for (int p = 0; p < 6; ++p)
        if (((p + rot) & 7) == wave)
          st_glds16_off(a.wfrag + ((size_t)(c_t * a.K + k) * 2) * ST_WBLK3 + 1024 * p, (unsigned)lane * 16u,
                        __builtin_amdgcn_readfirstlane((unsigned)ST_LDS_W + slot0 * ST_WBLK3 + 1024u * (unsigned)p));
    };
    const unsigned* const sRowG = reinterpret_cast<const unsigned*>(smem + ST_LDS_ROWS);
    const bool raggedG = (a.Fin & 15) != 0 && !pack_;
    auto gdma = [&](auto s_c, int n, int c, unsigned pdst) __attribute__((always_inline)) {
      constexpr int s = decltype(s_c)::value;
      const int piece = wave + 8 * s;
      const float* __restrict__ base = a.x + ((int64_t)(pack_ ? pack_ * n : n) * a.x_rows * a.Fin + c * 16);
      unsigned off = sRowG[16 * piece + (lane >> 2)];
      if (pack_) {  // slot q: map P n + q / (4 / P), its 16-byte piece q % (4 / P)
        const unsigned q = (ginfo >> (3 * s)) & 3u, mq = pack_ == 4 ? q : q >> 1;
        off += (unsigned)min((int)mq, a.n_maps - 1 - pack_ * n) * map_bytes + (pack_ == 4 ? 0u : 16u * (q & 1u));
      } else {
        off += 16u * ((ginfo >> (3 * s)) & 3u);
      }
      if (raggedG) {
        const int ch0 = c * 16 + 4 * (int)((ginfo >> (3 * s)) & 3u);
        if (ch0 >= a.Fin) off -= (unsigned)(ch0 - (a.Fin - 4)) * 4u;
      }
      if ((ginfo >> (3 * s)) & 4u) st_glds16_off(base, off, __builtin_amdgcn_readfirstlane(pdst + 1024u * (unsigned)piece));
    };
    auto gdma_w = [&](int u, int c, unsigned wdst) __attribute__((always_inline)) {
      const int j = wave + 8 * u;
      if (!stag && j < GW && j < wsliceG / 1024)
        st_glds16_off(a.wfrag + (size_t)c * wsliceG + 1024 * j, (unsigned)lane * 16u,
                      __builtin_amdgcn_readfirstlane(wdst + 1024u * (unsigned)j));
    };
    // group g of this wave's pieces of item (n, c): x0 x1 w0 | x2 x3 w1 | x4 w2.  With K = 5 (early) the third group
    // goes out with the second, in interval 3: every piece then lands before the barrier that ends interval 4, which
    // so doubles as the next item's "slice has landed" barrier.
    const bool early = a.K >= 5;
    auto gdma_group = [&](int g, int n, int c, unsigned pdst, unsigned wdst) __attribute__((always_inline)) {
      if (g == 0) {
        gdma(integral_constant<int, 0>{}, n, c, pdst);
        if (GX > 8) gdma(integral_constant<int, 1>{}, n, c, pdst);
        if (GW > 0) gdma_w(0, c, wdst);
      } else if (g == 1) {
        if (GX > 16) gdma(integral_constant<int, 2>{}, n, c, pdst);
        if (GX > 24) gdma(integral_constant<int, 3>{}, n, c, pdst);
        if (GW > 8) gdma_w(1, c, wdst);
      } else {
        if (GX > 32) gdma(integral_constant<int, 4>{}, n, c, pdst);
        if (GW > 16) gdma_w(2, c, wdst);
      }
    };
    float v[4][9];
    float4 ta[4], tb[4];  // T_{k-1} and T_{k-2} of this lane's four pixels, in alternating roles
#pragma unroll
    for (int p = 0; p < 4; ++p) ta[p] = tb[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned px = 0;
    __syncthreads();  // the contraction waves' row table (prologue)
    unsigned wbG = ST_LDS_W;  // weight buffer of the current item
    gdma_group(0, n_first, 0, px, wbG);
    gdma_group(1, n_first, 0, px, wbG);
    gdma_group(2, n_first, 0, px, wbG);
    if (stag) { worder(0, 0, 0, 0); worder(1, 0, 0, 6); worder(2, 0, 0, 4); worder(3, 0, 0, 2); }
    if (early) {  // the first item's slice: the only one waited for outside the item loop
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    for (int t = t_begin + slot0; t < t_end; t += nslots) {
      const unsigned row0 = (unsigned)a.tiles[t] * 256u;
      {  // L~ values of this lane's four pixels (blocks that no step touches load a valid row and never use it)
        const unsigned X0 = st_compress(row0), Y0 = st_compress(row0 >> 1);
        const bool ld_ok = (lact & 2u) != 0;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const unsigned gx = 2 * bx + 1 + (p & 1), gy = 2 * by + 1 + (p >> 1);
          if (TAB) {  // (cells that no step evaluates hold zeros in the table)
            const float4* __restrict__ q =
                reinterpret_cast<const float4*>(a.tabvals + ((size_t)t * ST_CELLS + st_cell_off(gx, gy) / 64u) * ST_TABV);
            const float4 n0 = q[0], n1 = q[1], n2 = q[2];
            v[p][0] = n0.x; v[p][1] = n0.y; v[p][2] = n0.z; v[p][3] = n0.w;
            v[p][4] = n1.x; v[p][5] = n1.y; v[p][6] = n1.z; v[p][7] = n1.w;
            v[p][8] = n2.x;
          } else {
            const unsigned rid = ld_ok ? st_morton(X0 + gx - ST_DMAX, Y0 + gy - ST_DMAX) : row0;
            const float4 n0 = *reinterpret_cast<const float4*>(a.gvals8 + (size_t)rid * 8);
            const float4 n1 = *reinterpret_cast<const float4*>(a.gvals8 + (size_t)rid * 8 + 4);
            v[p][0] = a.gdiag[rid];
            v[p][1] = n0.x; v[p][2] = n0.y; v[p][3] = n0.z; v[p][4] = n0.w;
            v[p][5] = n1.x; v[p][6] = n1.y; v[p][7] = n1.z; v[p][8] = n1.w;
          }
        }
      }
      int n = n_first, c = 0;  // map and slice of the current item
      for (int item = 0; item < items; ++item) {
#ifdef DSPH_STAMPS
        const bool stamp_on = wave < 7 && blockIdx.x == 72 && t == t_begin + slot0 + nslots && item >= 4 && item < 12;
#endif
        ST_STAMP(0);
        if (!early) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the item's x slice
          __syncthreads();  // B_a: the whole slice and the weights have landed
        }
        ST_STAMP(1);
        const unsigned pxn = px ^ (2u * ST_PLANE_BYTES);
        const unsigned wbnG = wbG == (unsigned)ST_LDS_W ? (unsigned)(ST_LDS_W + ST_WSLICE_BYTES) : (unsigned)ST_LDS_W;
        const bool more = item + 1 < items || t + nslots < t_end;
        const int cn = (c + 1 == a.C || item + 1 == items) ? 0 : c + 1;
        const int nn = item + 1 == items ? n_first : (c + 1 == a.C ? n + 1 : n);
        st_gather<true, false>(smem, px, py, gb, v, ta, tb, (lact & 2u) != 0, dummy);  // ta <- T_0, tb <- T_1
        ST_STAMP(2);
        __syncthreads();
        ST_STAMP(3);
        if (a.K > 2) {
          if (wact & 4u) st_gather<false, CHEB>(smem, py, px, gb, v, tb, ta, (lact & 4u) != 0, dummy);  // ta <- T_2
          if (more) gdma_group(0, nn, cn, pxn, wbnG);
          if (stag) {
            worder(4, itG, c, 0);
            if (more) worder(0, itG + 1, cn, 6);
          }
          ST_STAMP(4);
          __syncthreads();
          ST_STAMP(5);
        }
        if (a.K > 3) {
          // (a wave whose blocks all lie outside this step's region -- the border blocks sit in the last waves -- skips it)
          if (wact & 8u) st_gather<false, CHEB>(smem, px, py, gb, v, ta, tb, (lact & 8u) != 0, dummy);  // tb <- T_3
          if (more) gdma_group(1, nn, cn, pxn, wbnG);
          if (more && early) gdma_group(2, nn, cn, pxn, wbnG);
          if (stag && more) { worder(1, itG + 1, cn, 4); worder(3, itG + 1, cn, 2); }
          ST_STAMP(6);
          __syncthreads();
          ST_STAMP(7);
        }
        if (a.K > 4) {
          if (stag && more) worder(2, itG + 1, cn, 0);  // (first: the blocks need the interval to land)
          if (wact & 16u) st_gather<false, CHEB>(smem, py, px, gb, v, tb, ta, (lact & 16u) != 0, dummy);  // ta <- T_4
          if (more && !early) gdma_group(2, nn, cn, pxn, wbnG);
          ST_STAMP(8);
          if (early) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (issued an interval ago: long landed)
          __syncthreads();
          ST_STAMP(9);
        }
        if (more) {  // pieces that the shorter recurrences have not sent yet
          if (a.K <= 2) gdma_group(0, nn, cn, pxn, wbnG);
          if (a.K <= 3) gdma_group(1, nn, cn, pxn, wbnG);
          if (a.K <= 4) gdma_group(2, nn, cn, pxn, wbnG);
        }
        ++itG;
        px = pxn;
        wbG = wbnG;
        n = nn;
        c = cn;
      }
    }
    return;
  }

  // ===================================================================================================================
  // waves 8..11: contraction, LDS-DMA, y.  Wave c owns tile pixel rows 4c .. 4c+3; lane (r, h) of pixel block pb:
  // pixel (r & 15, 4c + 2pb + (r >> 4)), channels 8h .. 8h+7.
  // ===================================================================================================================
  // One wave per SIMD next to two recurrence waves, and the youngest of the three: at equal priority it gets the issue
  // slots the other two leave (its 8 LDS reads queue behind their 256, its bf16 split waits for their multiply-adds)
  // and every interval then ends when IT arrives at the barrier (1.9 k cycles per plane instead of 0.8 k).
  __builtin_amdgcn_s_setprio(3);
  const unsigned mr = lane & 31, mh = lane >> 5;
  unsigned mb[2];
#pragma unroll
  for (int pb = 0; pb < 2; ++pb) {
    const unsigned gx = ST_DMAX + (mr & 15), gy = ST_DMAX + 4 * cw + 2 * pb + (mr >> 4);
    mb[pb] = st_cell_off(gx, gy) + 16u * ((2 * mh) ^ st_cell_f(gx, gy));
  }
  // LDS-DMA: wave c issues x pieces c, c+4, ...; lane l fills slot l & 3 of cell 16 piece + (l >> 2) of the plane
  constexpr int GX = StPieces<PREC>::GX, GW = StPieces<PREC>::GW;
  constexpr int NPX = (ST_DMA_PIECES - GX + ST_CONTRACT_WAVES - 1) / ST_CONTRACT_WAVES;  // pieces per contraction wave: 8 with GX = 8 (the last of wave 3 does not exist: its valid bit is clear)
  unsigned dinfo = 0;  // per piece s: bits 3s..3s+1 logical slot, bit 3s+2 valid
#pragma unroll
  for (int s = 0; s < (NPX > 0 ? NPX : 1) && s < NPX; ++s) dinfo |= st_piece_info(GX + cw + ST_CONTRACT_WAVES * s, lane, D) << (3 * s);
  const int wblkB = PREC == DSPH_PREC_BF16X6 ? ST_WBLK3 : 2048;
  const int wslice = a.K * NB * wblkB;  // bytes of one slice's fragments
  const bool stag = PREC == DSPH_PREC_BF16X6 && a.K * NB > 6;  // in-place replacement of the weight orders (recurrence waves)
  unsigned itC = 0;  // this workgroup's item counter
  const int wpieces = wslice / 1024;
  const bool vec_ok = (a.Fout % 4 == 0) && (a.ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.y) & 15) == 0);
  const bool ragged = (a.Fin & 15) != 0 && !pack_;  // the last slice has channels past Fin: they are read from valid channels
  // Byte offsets (from the map's first element) of the x rows of the region cells of the tile being PREFETCHED, in LDS:
  // ten offsets per lane in registers (and what hipcc hoists around them) cost the contraction role 40 spilled registers.
  // Rebuilt by the four contraction waves in the first interval of a tile's last item, used from the second interval on
  // (a barrier lies between).  The host admits only maps of less than 4 GiB to this kernel.
  unsigned* const sRow = reinterpret_cast<unsigned*>(smem + ST_LDS_ROWS);
  auto build_rows = [&](int tpos) __attribute__((always_inline)) {
    const unsigned row0 = (unsigned)a.tiles[tpos] * 256u;
    const unsigned X0 = st_compress(row0), Y0 = st_compress(row0 >> 1);
    const int lo = ST_DMAX - D, hi = ST_DMAX + ST_TILE - 1 + D;
    for (unsigned p = (unsigned)(tid - 64 * ST_GATHER_WAVES); p < (unsigned)ST_CELLS; p += 64 * ST_CONTRACT_WAVES) {
      const unsigned par = p / ST_HP, rem = p % ST_HP, gxh = rem % ST_P2, gy = rem / ST_P2, gx = 2 * gxh + par;
      const bool ok = gxh < ST_S / 2 && (int)gx >= lo && (int)gx <= hi && (int)gy >= lo && (int)gy <= hi;
      const unsigned rid = TAB ? (unsigned)a.tabrow[(size_t)tpos * ST_CELLS + p]
                               : (ok ? st_morton(X0 + gx - ST_DMAX, Y0 + gy - ST_DMAX) : row0);
      sRow[p] = rid * (unsigned)a.Fin * 4u;
    }
  };
  using std::integral_constant;
  // piece s (compile-time) of the x slice of item (n, c) -> plane at pdst
  auto dma_x = [&](auto s_c, int n, int c, unsigned pdst) __attribute__((always_inline)) {
    constexpr int s = decltype(s_c)::value;
    const float* __restrict__ base = a.x + ((int64_t)(pack_ ? pack_ * n : n) * a.x_rows * a.Fin + c * 16);
    constexpr int piece_base = GX + ST_CONTRACT_WAVES * s;
    unsigned off = sRow[16 * (piece_base + cw) + (lane >> 2)];
    if (pack_) {
      const unsigned q = (dinfo >> (3 * s)) & 3u, mq = pack_ == 4 ? q : q >> 1;
      off += (unsigned)min((int)mq, a.n_maps - 1 - pack_ * n) * map_bytes + (pack_ == 4 ? 0u : 16u * (q & 1u));
    } else {
      off += 16u * ((dinfo >> (3 * s)) & 3u);
    }
    if (ragged) {
      const int ch0 = c * 16 + 4 * (int)((dinfo >> (3 * s)) & 3u);
      if (ch0 >= a.Fin) off -= (unsigned)(ch0 - (a.Fin - 4)) * 4u;  // meets zero weights
    }
    if ((dinfo >> (3 * s)) & 4u)
      st_glds16_off(base, off, __builtin_amdgcn_readfirstlane(pdst + 1024u * (unsigned)(piece_base + cw)));
  };
  // piece u of this wave's share of the weight fragments of slice c -> buffer at wdst
  auto dma_w = [&](int u, int c, unsigned wdst) __attribute__((always_inline)) {
    const int j = GW + cw + ST_CONTRACT_WAVES * u;
    if (!stag && j < wpieces)
      st_glds16_off(a.wfrag + (size_t)c * wslice + 1024 * j, (unsigned)lane * 16u,
                    __builtin_amdgcn_readfirstlane(wdst + 1024u * (unsigned)j));
  };
  // a group of pieces of the next item; groups 0, 1, 2 go out in the intervals 2, 3, 4 (not earlier: until the barrier
  // that ends interval 1 the previous item's last plane and weights are still being contracted out of those buffers)
  // group g (0, 1, 2 -> intervals 2, 3, 4) of this wave's pieces: x pieces 4g .. 4g+3 (the last group takes the rest),
  // weight pieces 2g, 2g+1
  auto dma_group = [&](int g, int n, int c, unsigned pdst, unsigned wdst) __attribute__((always_inline)) {
    if (g == 0) {
      if (NPX > 0) dma_x(integral_constant<int, 0>{}, n, c, pdst);
      if (NPX > 1) dma_x(integral_constant<int, 1>{}, n, c, pdst);
      if (NPX > 2) dma_x(integral_constant<int, 2>{}, n, c, pdst);
      if (NPX > 6) dma_x(integral_constant<int, 3>{}, n, c, pdst);
      dma_w(0, c, wdst); dma_w(1, c, wdst);
    } else if (g == 1) {
      if (NPX > 6) { dma_x(integral_constant<int, 4>{}, n, c, pdst); dma_x(integral_constant<int, 5>{}, n, c, pdst); dma_x(integral_constant<int, 6>{}, n, c, pdst); }
      else if (NPX > 3) dma_x(integral_constant<int, 3>{}, n, c, pdst);
      if (NPX > 7) dma_x(integral_constant<int, 7>{}, n, c, pdst);
      if (NPX <= 6 && NPX > 4) dma_x(integral_constant<int, 4>{}, n, c, pdst);
      dma_w(2, c, wdst); dma_w(3, c, wdst);
    } else {
      if (NPX > 8) dma_x(integral_constant<int, 8>{}, n, c, pdst);
      if (NPX > 9) dma_x(integral_constant<int, 9>{}, n, c, pdst);
      if (NPX <= 6 && NPX > 5) dma_x(integral_constant<int, 5>{}, n, c, pdst);
      dma_w(4, c, wdst);
    }
  };

  st_f32x16 acc[2][NB];
#pragma unroll
  for (int pb = 0; pb < 2; ++pb)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[pb][b][r] = 0.f;

  int t = t_begin + slot0;
  unsigned px = 0;         // byte offset of the X plane of the current item (plane 0 or 2); Y is plane 1
  unsigned wb = ST_LDS_W;  // weight buffer of the current item
  build_rows(t);
  __syncthreads();  // (prologue only; the gather waves meet it below)
  dma_group(0, n_first, 0, px, wb);
  dma_group(1, n_first, 0, px, wb);
  dma_group(2, n_first, 0, px, wb);
  const bool early = a.K >= 5;  // (see the recurrence waves)
  if (early) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  // the last plane of the previous item, still to be contracted (under the first step of the current one)
  bool pend = false, pend_store = false;
  unsigned pend_plane = 0, pend_w = 0, pend_row0 = 0;
  // byte offset of the fragments of order k of the current item
  auto wof = [&](int k) __attribute__((always_inline)) -> unsigned {
    if (stag) return (unsigned)ST_LDS_W + ((k == 3 && (itC & 1u)) ? 10u : (unsigned)(2 * k)) * ST_WBLK3;
    return wb + (unsigned)(k * NB * wblkB);
  };
  int pend_n = 0;
  auto flush_pending = [&]() __attribute__((always_inline)) {
    if (!pend) return;
    st_contract<NB, PREC>(smem, pend_plane, pend_w, mb, lane, acc);
    if (pend_store) {  // that completed a map: y, then fresh accumulators
      float* __restrict__ yt = pool_ ? a.ypool + ((int64_t)(pack_ ? pack_ * pend_n : pend_n) * a.ypool_rows + (pend_row0 >> 2)) * a.ld
                                      : a.y + ((int64_t)(pack_ ? pack_ * pend_n : pend_n) * a.y_rows + pend_row0) * a.ld;
      st_store<NB>(acc, smem, pend_plane, cw, yt, a.ld, sBias, lane, a.Fout,
                   a.act == DSPH_ACT_RELU ? 0.f : -__builtin_huge_valf(), vec_ok, pack_, (pool_ ? a.ypool_rows : a.y_rows) * (int64_t)a.ld,
                   a.n_maps - pack_ * pend_n, pool_);
#pragma unroll
      for (int pb = 0; pb < 2; ++pb)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[pb][b][r] = 0.f;
    }
    pend = false;
  };

  for (; t < t_end; t += nslots) {
    const unsigned row0 = (unsigned)a.tiles[t] * 256u;
    int n = n_first, c = 0;  // map and slice of the current item
    for (int item = 0; item < items; ++item) {
#ifdef DSPH_STAMPS
      const bool stamp_on = cw == 0 && blockIdx.x == 72 && t == t_begin + slot0 + nslots && item >= 4 && item < 12;
#endif
      ST_STAMP(0);
      // ---- B_a: this wave's pieces of the item's x slice and weights have landed ----------------------------------------
      if (!early) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ST_STAMP(1);
        __syncthreads();
      }
      ST_STAMP(2);
      const unsigned pxn = px ^ (2u * ST_PLANE_BYTES);
      const unsigned wbn = wb == (unsigned)ST_LDS_W ? (unsigned)(ST_LDS_W + ST_WSLICE_BYTES) : (unsigned)ST_LDS_W;
      const bool more = item + 1 < items || t + nslots < t_end;
      // map and slice of the next item (of this tile, or the first of this workgroup's next tile)
      const int cn = (c + 1 == a.C || item + 1 == items) ? 0 : c + 1;
      const int nn = item + 1 == items ? n_first : (c + 1 == a.C ? n + 1 : n);
      // ---- interval 1: the previous item's last plane (and y of a finished map), then T_0 ----------------------------------
      if (item + 1 == items && more) build_rows(t + nslots);
      flush_pending();
      ST_STAMP(3);
      st_contract<NB, PREC>(smem, px, wof(0), mb, lane, acc);
      ST_STAMP(4);
      __syncthreads();
      ST_STAMP(5);
      // ---- intervals 2 .. K-1: T_{k-1}, and the next item's pieces ---------------------------------------------------------
      if (a.K > 2) {
        st_contract<NB, PREC>(smem, py, wof(1), mb, lane, acc);
        ST_STAMP(6);
        if (more) dma_group(0, nn, cn, pxn, wbn);
        ST_STAMP(7);
        __syncthreads();
        ST_STAMP(8);
      }
      if (a.K > 3) {
        st_contract<NB, PREC>(smem, px, wof(2), mb, lane, acc);
        ST_STAMP(9);
        if (more) dma_group(1, nn, cn, pxn, wbn);
        if (more && early) dma_group(2, nn, cn, pxn, wbn);
        ST_STAMP(10);
        __syncthreads();
        ST_STAMP(11);
      }
      if (a.K > 4) {
        st_contract<NB, PREC>(smem, py, wof(3), mb, lane, acc);
        ST_STAMP(12);
        if (more && !early) dma_group(2, nn, cn, pxn, wbn);
        ST_STAMP(13);
        if (early) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // pieces (an interval old) and this wave's y stores
        __syncthreads();
        ST_STAMP(14);
      }
      if (more) {  // pieces that the shorter recurrences have not sent yet
        if (a.K <= 2) dma_group(0, nn, cn, pxn, wbn);
        if (a.K <= 3) dma_group(1, nn, cn, pxn, wbn);
        if (a.K <= 4) dma_group(2, nn, cn, pxn, wbn);
      }
      // T_{K-1}: left for the next item's first interval when it sits in the old X plane (K odd), which nobody touches
      // before that interval's barrier; in the Y plane (K even) the next item's first step would overwrite it, so it is
      // contracted now, in front of the next B_a
      pend = true;
      pend_plane = ((a.K - 1) & 1) ? py : px;
      pend_w = wof(a.K - 1);
      pend_store = c == a.C - 1;
      pend_n = n;
      pend_row0 = row0;
      if ((a.K - 1) & 1) flush_pending();
      ++itC;
      px = pxn;
      wb = wbn;
      n = nn;
      c = cn;
    }
  }
  flush_pending();
}

}  // namespace dsph
