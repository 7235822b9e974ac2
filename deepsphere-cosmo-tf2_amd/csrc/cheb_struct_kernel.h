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
//   * plane T_{k-1} is contracted (MFMA) in the same barrier interval in which step k reads it; the two waves of
//     a SIMD take the two halves of the interval in opposite order, so one gathers while the other feeds the
//     matrix pipe;
//   * the contraction is transposed (A = weight fragment, B = plane fragment): the accumulator holds 4 consecutive
//     output channels of ONE pixel per register quad, so y is stored by dwordx4 straight from the registers.
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

constexpr int ST_TILE = 16;                     // tile side in pixels: 256 consecutive NEST rows
constexpr int ST_DMAX = 4;                      // halo rings held: K <= 5
constexpr int ST_S = ST_TILE + 2 * ST_DMAX;     // plane side, 24 cells
constexpr int ST_P2 = 13;                       // cells per half-row (12 + 1 pad: see tools/gen_struct_tables.py)
constexpr int ST_HP = ST_S * ST_P2;             // cells per column-parity half
constexpr int ST_CELLS = 2 * ST_HP;             // 624 (48 of them padding)
constexpr int ST_PLANE_BYTES = ST_CELLS * 64;   // 39,936
constexpr int ST_THREADS = 512;
constexpr int ST_WSLICE_BYTES = 20480;          // weight fragments of one slice: K * NB * 2048 <= 20480
constexpr int ST_LDS_W = 3 * ST_PLANE_BYTES;    // 119,808
constexpr int ST_LDS_BIAS = ST_LDS_W + 2 * ST_WSLICE_BYTES;  // 160,768
constexpr int ST_LDS_TOTAL = ST_LDS_BIAS + 256;              // 161,024 of 163,840
constexpr int ST_DMA_PIECES = ST_CELLS / 16;    // 39 wave-instructions of 1 KiB fill a plane

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
  int64_t x_rows, y_rows;
  int ntiles, N, Fin, Fout, K, C, act, ld;
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
      if (lane == 0) a.stamps[((size_t)wave * 8 + (item - 4)) * 32 + (id)] = t_;           \
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
__device__ __forceinline__ void st_glds16_off(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}

// One recurrence step for this lane's 2x2 block: window from plane `pin` (byte offset), new values to `pout`.
//   FIRST: T_1 = L~ T_0, the block's own T_0 is read too (16 reads); otherwise the centre of the window is `cur`.
//   CHEB : T_k = 2 L~ T_{k-1} - T_{k-2}
template <bool FIRST, bool CHEB>
__device__ __forceinline__ void st_gather(const unsigned char* __restrict__ smem, unsigned pin, unsigned pout,
                                          const unsigned (&gb)[4], const float (&v)[4][9], float4 (&cur)[4],
                                          float4 (&prev)[4]) {
  // Two halves, so that at most 8 + 4 window cells are in registers at a time: the pixels of the block's upper row
  // need window rows 0..2, those of the lower row need rows 1..3 (row 3 is fetched while the upper row is summed).
  float4 W[4][4];
  auto rd = [&](int wx, int wy) {
    W[wy][wx] = *reinterpret_cast<const float4*>(smem + (gb[st_cell_f(wx, wy)] + pin) + st_woff(wx, wy));
  };
#pragma unroll
  for (int wy = 0; wy < 3; ++wy)
#pragma unroll
    for (int wx = 0; wx < 4; ++wx) {
      const bool centre = (wx == 1 || wx == 2) && (wy == 1 || wy == 2);
      if (centre && !FIRST) W[wy][wx] = cur[(wy - 1) * 2 + (wx - 1)];
      else rd(wx, wy);
    }
  auto pixel = [&](int p) {
    const int i = p & 1, j = p >> 1;
    const float4 c = W[j + 1][i + 1];
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    s.x = fmaf(v[p][0], c.x, s.x);
    s.y = fmaf(v[p][0], c.y, s.y);
    s.z = fmaf(v[p][0], c.z, s.z);
    s.w = fmaf(v[p][0], c.w, s.w);
#pragma unroll
    for (int d = 0; d < 8; ++d) {
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
    prev[p] = c;
    cur[p] = s;
    *reinterpret_cast<float4*>(const_cast<unsigned char*>(smem) + (gb[st_cell_f(i + 1, j + 1)] + pout) +
                               st_woff(i + 1, j + 1)) = s;
  };
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int wx = 0; wx < 4; ++wx) rd(wx, 3);
  // the lower pixels' T_{k-1} (rows 1, 2 of the window) are still needed below: pixel() overwrites cur[] only
  pixel(0);
  pixel(1);
  __builtin_amdgcn_sched_barrier(0);
  pixel(2);
  pixel(3);
}

// Plane T_k (this wave's 32 tile pixels, 16 channels) into the accumulators: acc[b] += Wfrag(k, b)^T-form product.
// mb0 / mb1: byte offsets (inside a plane) of this lane's two 16-byte slots (channels 8h..8h+3 and 8h+4..8h+7).
template <int NB, int PREC>
__device__ __forceinline__ void st_contract(const unsigned char* __restrict__ smem, unsigned plane, unsigned wblk,
                                            unsigned mb0, unsigned mb1, int lane, st_f32x16 (&acc)[NB]) {
  const float4 a0 = *reinterpret_cast<const float4*>(smem + plane + mb0);
  const float4 a1 = *reinterpret_cast<const float4*>(smem + plane + mb1);
  const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
  if (PREC == DSPH_PREC_BF16X3) {
    st_bf16x8 whi[NB], wlo[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      whi[b] = *reinterpret_cast<const st_bf16x8*>(smem + wblk + b * 2048 + lane * 16);
      wlo[b] = *reinterpret_cast<const st_bf16x8*>(smem + wblk + b * 2048 + 1024 + lane * 16);
    }
    st_bf16x8 thi, tlo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const __bf16 hi = (__bf16)av[j];
      thi[j] = hi;
      tlo[j] = (__bf16)(av[j] - (float)hi);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[b], tlo, acc[b], 0, 0, 0);  // small terms first
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo[b], thi, acc[b], 0, 0, 0);
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[b], thi, acc[b], 0, 0, 0);
    }
  } else {
    float wf[NB][8];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const float4 w0 = *reinterpret_cast<const float4*>(smem + wblk + b * 2048 + lane * 16);
      const float4 w1 = *reinterpret_cast<const float4*>(smem + wblk + b * 2048 + 1024 + lane * 16);
      wf[b][0] = w0.x; wf[b][1] = w0.y; wf[b][2] = w0.z; wf[b][3] = w0.w;
      wf[b][4] = w1.x; wf[b][5] = w1.y; wf[b][6] = w1.z; wf[b][7] = w1.w;
    }
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[b][t], av[t], acc[b], 0, 0, 0);
  }
}

// Contract T_{k-1} and compute T_k in one straight-line block of three phases (fenced for the scheduler, so that the
// live ranges stay what the source says): A every LDS read that does not depend on anything; B the upper pixel row's
// multiply-adds with the first column block's MFMAs underneath; C the lower pixel row with the second block's.
// `wr`: this lane's block is part of step k (the others compute on whatever their cells hold and write to a pad cell).
template <bool FIRST, bool CHEB, int NB, int PREC, class StampFn>
__device__ __forceinline__ void st_interval(const unsigned char* __restrict__ smem, unsigned pin, unsigned pout,
                                            unsigned wblk, const unsigned (&gb)[4], const float (&v)[4][9],
                                            float4 (&cur)[4], float4 (&prev)[4], unsigned mb0, unsigned mb1, int lane,
                                            st_f32x16 (&acc)[NB], bool wr, unsigned dummy, StampFn stamp) {
  // lanes whose block is not part of this step store into a pad cell of the output plane instead of branching:
  // the interval stays one basic block, which is what lets the scheduler put the MFMAs under the multiply-adds
  unsigned ob[4];
#pragma unroll
  for (int p = 0; p < 4; ++p)
    ob[p] = (wr ? gb[st_cell_f((p & 1) + 1, (p >> 1) + 1)] + st_woff((p & 1) + 1, (p >> 1) + 1) : dummy) + pout;
  float4 W[4][4];
  auto rd = [&](int wx, int wy) {
    W[wy][wx] = *reinterpret_cast<const float4*>(smem + (gb[st_cell_f(wx, wy)] + pin) + st_woff(wx, wy));
  };
  auto pixel = [&](int p) {
    const int i = p & 1, j = p >> 1;
    const float4 c = W[j + 1][i + 1];
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    s.x = fmaf(v[p][0], c.x, s.x);
    s.y = fmaf(v[p][0], c.y, s.y);
    s.z = fmaf(v[p][0], c.z, s.z);
    s.w = fmaf(v[p][0], c.w, s.w);
#pragma unroll
    for (int d = 0; d < 8; ++d) {
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
    if (FIRST) cur[p] = c;  // the block's own T_0, read from the plane in this step only
    prev[p] = s;            // T_k takes the place of T_{k-2}: the caller swaps the two arrays' roles, no moves
    *reinterpret_cast<float4*>(const_cast<unsigned char*>(smem) + ob[p]) = s;
  };
  // ---- A: reads ---------------------------------------------------------------------------------------------------
  const float4 a0 = *reinterpret_cast<const float4*>(smem + pin + mb0);
  const float4 a1 = *reinterpret_cast<const float4*>(smem + pin + mb1);
  const unsigned char* __restrict__ wp = smem + wblk + lane * 16;
  float4 w0h = *reinterpret_cast<const float4*>(wp), w0l = *reinterpret_cast<const float4*>(wp + 1024);
#pragma unroll
  for (int wy = 0; wy < 3; ++wy)
#pragma unroll
    for (int wx = 0; wx < 4; ++wx) {
      const bool centre = (wx == 1 || wx == 2) && (wy == 1 || wy == 2);
      if (centre && !FIRST) W[wy][wx] = cur[(wy - 1) * 2 + (wx - 1)];
      else rd(wx, wy);
    }
  __builtin_amdgcn_sched_barrier(0);
  stamp(0);
  // ---- B: first column block under the upper pixel row ------------------------------------------------------------
  const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
  st_bf16x8 thi, tlo;
  if (PREC == DSPH_PREC_BF16X3) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const __bf16 hi = (__bf16)av[j];
      thi[j] = hi;
      tlo[j] = (__bf16)(av[j] - (float)hi);
    }
  }
  auto mfma_block = [&](int b, const float4& wh, const float4& wl) {
    if (PREC == DSPH_PREC_BF16X3) {
      const st_bf16x8 whi = __builtin_bit_cast(st_bf16x8, wh), wlo = __builtin_bit_cast(st_bf16x8, wl);
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi, tlo, acc[b], 0, 0, 0);  // small terms first
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo, thi, acc[b], 0, 0, 0);
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi, thi, acc[b], 0, 0, 0);
    } else {
      const float wf[8] = {wh.x, wh.y, wh.z, wh.w, wl.x, wl.y, wl.z, wl.w};
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[t], av[t], acc[b], 0, 0, 0);
    }
  };
#pragma unroll
  for (int wx = 0; wx < 4; ++wx) rd(wx, 3);
  // the MFMAs are issued in front of the pixels' multiply-adds and run underneath them
  mfma_block(0, w0h, w0l);
  if (NB == 2) {  // the second block's fragments take over the first one's registers
    w0h = *reinterpret_cast<const float4*>(wp + 2048);
    w0l = *reinterpret_cast<const float4*>(wp + 2048 + 1024);
  }
  pixel(0);
  pixel(1);
  if (PREC == DSPH_PREC_FP32) {  // eight 64-cycle MFMAs per phase: one, then a run of the pixels' multiply-adds, and so on
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 11, 0);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  stamp(1);
  // ---- C: second column block under the lower pixel row -----------------------------------------------------------
  if (NB == 2) mfma_block(1, w0h, w0l);
  pixel(2);
  pixel(3);
  if (PREC == DSPH_PREC_FP32 && NB == 2) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);
    }
  }
}

// y of one map.  The accumulators hold, per lane (pixel r, half h), register quad tq of block b = output channels
// 32 b + 8 tq + 4 h .. + 3: stored as they stand a wave instruction touches 32 rows x 2 x 16 B (measured: 250-500
// cycles of issue per instruction, every wave at once).  They go through a 32 x 32 block of LDS instead (the plane that
// no longer holds anything: free until the next item's first step), so that eight lanes write 128 contiguous bytes.
constexpr int ST_SCR_PITCH = 144;                      // bytes per pixel row of the block (32 floats + 4 pad)
constexpr int ST_SCR_WAVE = 32 * ST_SCR_PITCH;         // 4,608 B per wave; 8 waves = 36,864 <= ST_PLANE_BYTES
template <int NB, int ACT, bool VEC>
__device__ __forceinline__ void st_store(const st_f32x16 (&acc)[NB], unsigned char* __restrict__ scr,
                                         float* __restrict__ ytile, int ld, const float* __restrict__ sBias, int wave,
                                         int lane, int Fout, int act_rt) {
  const int act = ACT >= 0 ? ACT : act_rt;
  const int r = lane & 31, h = lane >> 5;
  const int j = lane & 7, pq = lane >> 3;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
#pragma unroll
    for (int tq = 0; tq < 4; ++tq)
      *reinterpret_cast<float4*>(scr + r * ST_SCR_PITCH + (8 * tq + 4 * h) * 4) =
          make_float4(acc[b][4 * tq + 0], acc[b][4 * tq + 1], acc[b][4 * tq + 2], acc[b][4 * tq + 3]);
    const int ch = 32 * b + 4 * j;
    const float4 bv = *reinterpret_cast<const float4*>(sBias + ch);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = pq + 8 * i;  // pixel (p & 15, 2 wave + (p >> 4)) of the tile
      float4 o = *reinterpret_cast<const float4*>(scr + p * ST_SCR_PITCH + 16 * j);
      o.x = apply_act(o.x + bv.x, act);
      o.y = apply_act(o.y + bv.y, act);
      o.z = apply_act(o.z + bv.z, act);
      o.w = apply_act(o.w + bv.w, act);
      float* __restrict__ yp = ytile + (int64_t)st_morton((unsigned)(p & 15), (unsigned)(2 * wave + (p >> 4))) * ld + ch;
      if (VEC) {
        if (ch < Fout) *reinterpret_cast<float4*>(yp) = o;
      } else {
        if (ch + 0 < Fout) yp[0] = o.x;
        if (ch + 1 < Fout) yp[1] = o.y;
        if (ch + 2 < Fout) yp[2] = o.z;
        if (ch + 3 < Fout) yp[3] = o.w;
      }
    }
  }
}

template <int NB, int PREC, bool CHEB>
__global__ __launch_bounds__(ST_THREADS, 2) void cheb_struct_kernel(StructArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[ST_LDS_TOTAL];
  float* const sBias = reinterpret_cast<float*>(smem + ST_LDS_BIAS);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: scalar registers, "s" asm operands
  const int D = a.K - 1;
  if (tid < 64) sBias[tid] = (a.bias != nullptr && tid < a.Fout) ? a.bias[tid] : 0.f;

  // ---- gather role: quad -> 2x2 block (odd origin), lane -> 16-byte slot ------------------------------------
  const unsigned blk = kStructBlock[tid >> 2];
  const bool has_blk = blk != 0xffu;
  const int bx = has_blk ? (int)(blk & 15u) : 0, by = has_blk ? (int)(blk >> 4) : 0;
  const unsigned q = tid & 3;
  unsigned gb[4];  // byte offset of the window's top-left cell (2bx, 2by), slot q ^ f, f = 0..3
#pragma unroll
  for (unsigned f = 0; f < 4; ++f) gb[f] = st_cell_off(2 * bx, 2 * by) + 16u * (q ^ f);
  // first / last block index (per axis) that step j touches: region [4-(D-j), 19+(D-j)], block b = cells 2b+1, 2b+2
  auto step_lo = [&](int j) { return (4 - (D - j) - 1) >> 1; };
  auto step_hi = [&](int j) { return (19 + (D - j) - 1) >> 1; };
  auto active_at = [&](int j) {
    const int lo = step_lo(j), hi = step_hi(j);
    return has_blk && bx >= lo && bx <= hi && by >= lo && by <= hi;
  };

  // ---- contraction role: wave w owns tile pixel rows 2w, 2w+1; lane (r, h): pixel r, channels 8h..8h+7 ---------
  const unsigned mr = lane & 31, mh = lane >> 5;
  const unsigned mpx = mr & 15, mpy = 2 * wave + (mr >> 4);
  const unsigned mgx = ST_DMAX + mpx, mgy = ST_DMAX + mpy;
  const unsigned mb0 = st_cell_off(mgx, mgy) + 16u * ((2 * mh) ^ st_cell_f(mgx, mgy));
  const unsigned mb1 = st_cell_off(mgx, mgy) + 16u * ((2 * mh + 1) ^ st_cell_f(mgx, mgy));

  // ---- DMA role: wave w issues pieces w, w+8, ...; lane l fills slot l & 3 of cell 16 i + (l >> 2) of the plane --
  constexpr int NP = (ST_DMA_PIECES + 7) / 8;  // 5
  // cell (gx, gy) and logical slot of piece s of this lane; valid: inside the plane and inside the D-ring region
  auto dma_cell = [&](int s, unsigned& gx, unsigned& gy, unsigned& slot) -> bool {
    const int piece = wave + 8 * s;
    const unsigned p = 16u * piece + (lane >> 2);
    const unsigned par = p / ST_HP, rem = p % ST_HP, gxh = rem % ST_P2;
    gy = rem / ST_P2;
    gx = 2 * gxh + par;
    slot = (lane & 3u) ^ st_cell_f(gx, gy);
    const int lo = ST_DMAX - D, hi = ST_DMAX + ST_TILE - 1 + D;
    return piece < ST_DMA_PIECES && gxh < ST_S / 2 && (int)gx >= lo && (int)gx <= hi && (int)gy >= lo && (int)gy <= hi;
  };
  unsigned dinfo = 0;  // per piece s: bits 3s..3s+1 logical slot, bit 3s+2 valid
#pragma unroll
  for (int s = 0; s < NP; ++s) {
    unsigned gx, gy, slot;
    const bool ok = dma_cell(s, gx, gy, slot);
    dinfo |= (slot | (ok ? 4u : 0u)) << (3 * s);
  }

  // tiles are dealt to XCDs in contiguous ranges (blocks b and b+8 share an XCD and its L2)
  const int G = gridDim.x, xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3;
  const int nslots = (G + 7 - xcd) / 8;
  const int t_begin = (int)((int64_t)a.ntiles * xcd / 8), t_end = (int)((int64_t)a.ntiles * (xcd + 1) / 8);
  const int items = a.N * a.C;
  const int wslice = a.K * NB * 2048;  // bytes of one slice's fragments
  const int wpieces = wslice / 1024;
  const bool vec_ok = (a.Fout % 4 == 0) && (a.ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.y) & 15) == 0);

  // per-lane activity: bit k = this lane's block is part of step k
  unsigned lact = 0;
#pragma unroll
  for (int k = 1; k <= ST_DMAX; ++k) lact |= (k <= D && active_at(k)) ? (1u << k) : 0u;
  // where the other lanes' stores go: the pad cell (half-row index 12) of their window's second row, own slot
  const unsigned dummy = (unsigned)((2 * by + 1) * ST_P2 + ST_S / 2) * 64u + 16u * q;

  // byte offsets (from the map's first element) of the x pieces this lane fetches, for the tile being PREFETCHED
  // (the host admits only maps of less than 4 GiB to this kernel)
  unsigned doff[NP];
  auto set_doffs = [&](int tpos) {
    const unsigned row0 = (unsigned)a.tiles[tpos] * 256u;
    const unsigned X0 = st_compress(row0), Y0 = st_compress(row0 >> 1);
#pragma unroll
    for (int s = 0; s < NP; ++s) {
      unsigned gx, gy, slot;
      const bool ok = dma_cell(s, gx, gy, slot);
      const unsigned rid = ok ? st_morton(X0 + gx - ST_DMAX, Y0 + gy - ST_DMAX) : row0;
      doff[s] = (rid * (unsigned)a.Fin + 4u * slot) * 4u;
    }
  };
  const bool ragged = (a.Fin & 15) != 0;  // the last slice has channels past Fin: they are read from valid channels
  // piece s (compile-time) of the x slice of item `it` -> plane at pdst
  auto dma_x = [&](auto s_c, int n, int c, unsigned pdst) {
    constexpr int s = decltype(s_c)::value;
    const float* __restrict__ base = a.x + ((int64_t)n * a.x_rows * a.Fin + c * 16);
    unsigned off = doff[s];
    if (ragged) {
      const int ch0 = c * 16 + 4 * (int)((dinfo >> (3 * s)) & 3u);
      if (ch0 >= a.Fin) off -= (unsigned)(ch0 - (a.Fin - 4)) * 4u;  // meets zero weights
    }
    if ((dinfo >> (3 * s)) & 4u)
      st_glds16_off(base, off, __builtin_amdgcn_readfirstlane(pdst + 1024u * (unsigned)(wave + 8 * s)));
  };
  // piece u of this wave's share of the weight fragments of item `it`'s slice -> buffer at wdst
  auto dma_w = [&](int u, int c, unsigned wdst) {
    const int j = wave + 8 * u;
    if (j < wpieces) {
      st_glds16_off(a.wfrag + (size_t)c * wslice + 1024 * j, (unsigned)lane * 16u,
                    __builtin_amdgcn_readfirstlane(wdst + 1024u * (unsigned)j));
    }
  };
  using std::integral_constant;

  st_f32x16 acc[NB];
  float v[4][9];
  float4 ta[4], tb[4];  // T_{k-1} and T_{k-2} of this lane's four pixels, in alternating roles
#pragma unroll
  for (int p = 0; p < 4; ++p) ta[p] = tb[p] = make_float4(0.f, 0.f, 0.f, 0.f);

  int t = t_begin + slot0;
  if (t >= t_end) return;
  unsigned px = 0;         // byte offset of the X plane of the current item (plane 0 or 2); Y is plane 1
  unsigned wb = ST_LDS_W;  // weight buffer of the current item
  constexpr unsigned py = ST_PLANE_BYTES;
  set_doffs(t);
  dma_x(integral_constant<int, 0>{}, 0, 0, px);
  dma_x(integral_constant<int, 1>{}, 0, 0, px);
  dma_x(integral_constant<int, 2>{}, 0, 0, px);
  dma_x(integral_constant<int, 3>{}, 0, 0, px);
  dma_x(integral_constant<int, 4>{}, 0, 0, px);
  for (int u = 0; u < 3; ++u) dma_w(u, 0, wb);
#ifdef DSPH_ST_PRIO
  if (wave >= 4) __builtin_amdgcn_s_setprio(DSPH_ST_PRIO);  // the second wave of every SIMD: see Makefile
#endif
  bool stored = false;  // the previous item ended with this wave's y stores (the youngest vector-memory operations)

  for (; t < t_end; t += nslots) {
    const unsigned row0 = (unsigned)a.tiles[t] * 256u;
    {  // L~ values of this lane's four pixels (blocks that no step touches keep zeros and load nothing)
      const unsigned X0 = st_compress(row0), Y0 = st_compress(row0 >> 1);
      const bool ld_ok = (lact & 2u) != 0;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const unsigned gx = 2 * bx + 1 + (p & 1), gy = 2 * by + 1 + (p >> 1);
        const unsigned rid = ld_ok ? st_morton(X0 + gx - ST_DMAX, Y0 + gy - ST_DMAX) : row0;
        const float4 n0 = *reinterpret_cast<const float4*>(a.gvals8 + (size_t)rid * 8);
        const float4 n1 = *reinterpret_cast<const float4*>(a.gvals8 + (size_t)rid * 8 + 4);
        v[p][0] = a.gdiag[rid];
        v[p][1] = n0.x; v[p][2] = n0.y; v[p][3] = n0.z; v[p][4] = n0.w;
        v[p][5] = n1.x; v[p][6] = n1.y; v[p][7] = n1.z; v[p][8] = n1.w;
      }
    }
    int n = 0, c = 0;  // map and slice of the current item
    for (int item = 0; item < items; ++item) {
#ifdef DSPH_STAMPS
      const bool stamp_on = blockIdx.x == 72 && t == t_begin + slot0 + nslots && item >= 4 && item < 12;
#endif
      ST_STAMP(0);
      // ---- B_a: this item's x slice and weights have landed (the y stores just issued may still be in flight);
      //      every LDS read of the previous item is done ------------------------------------------------------------
      if (stored && vec_ok) {
        if (NB == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      ST_STAMP(1);
      __syncthreads();
      ST_STAMP(2);
      const unsigned pxn = px ^ (2u * ST_PLANE_BYTES);
      const unsigned wbn = wb == (unsigned)ST_LDS_W ? (unsigned)(ST_LDS_W + ST_WSLICE_BYTES) : (unsigned)ST_LDS_W;
      // the next item (of this tile, or the first of this workgroup's next tile) is fetched piece by piece between
      // the phases below: a burst of eight pieces per wave would hold every wave at the address unit for 2-4 k cycles
      const bool more = item + 1 < items || t + nslots < t_end;
      // map and slice of the next item (of this tile, or the first of this workgroup's next tile)
      const int cn = (c + 1 == a.C || item + 1 == items) ? 0 : c + 1;
      const int nn = item + 1 == items ? 0 : (c + 1 == a.C ? n + 1 : n);
      if (item + 1 == items && more) set_doffs(t + nslots);
      if (more) dma_x(integral_constant<int, 0>{}, nn, cn, pxn);
      ST_STAMP(3);
      if (c == 0) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
        asm volatile("" ::: "memory");  // a real branch: as selects the zeroing costs 32 instructions in every item
      }
      // ---- interval k = 1 .. K-1: contract T_{k-1}, compute T_k ---------------------------------------------------
#define ST_INTERVAL(k, FIRST_, PIN, POUT, CUR, PREV)                                                        \
  {                                                                                                         \
    const unsigned wk = wb + (unsigned)(((k) - 1) * NB * 2048);                                             \
    st_interval<FIRST_, CHEB, NB, PREC>(smem, PIN, POUT, wk, gb, v, CUR, PREV, mb0, mb1, lane, acc,         \
                                        (lact & (1u << (k))) != 0, dummy, [&](int id) {                     \
                                          if ((k) == 2) { ST_STAMP(18 + id); }                              \
                                        });                                                                 \
  }
      ST_INTERVAL(1, true, px, py, ta, tb)   // ta <- T_0, tb <- T_1
      ST_STAMP(4);
      if (more) { dma_x(integral_constant<int, 1>{}, nn, cn, pxn); dma_w(0, cn, wbn); }
      ST_STAMP(5);
      __syncthreads();
      ST_STAMP(6);
      if (a.K > 2) {
        ST_INTERVAL(2, false, py, px, tb, ta)  // ta <- T_2
        ST_STAMP(7);
        if (more) { dma_x(integral_constant<int, 2>{}, nn, cn, pxn); dma_w(1, cn, wbn); }
        ST_STAMP(8);
        __syncthreads();
        ST_STAMP(9);
      }
      if (a.K > 3) {
        ST_INTERVAL(3, false, px, py, ta, tb)  // tb <- T_3
        ST_STAMP(10);
        if (more) { dma_x(integral_constant<int, 3>{}, nn, cn, pxn); dma_w(2, cn, wbn); }
        ST_STAMP(11);
        __syncthreads();
        ST_STAMP(12);
      }
      if (a.K > 4) {
        ST_INTERVAL(4, false, py, px, tb, ta)  // ta <- T_4
        ST_STAMP(13);
        if (more) dma_x(integral_constant<int, 4>{}, nn, cn, pxn);
        ST_STAMP(14);
        __syncthreads();
        ST_STAMP(15);
      }
#undef ST_INTERVAL
      if (more) {  // pieces that the shorter recurrences have not sent yet
        if (a.K <= 2) { dma_x(integral_constant<int, 2>{}, nn, cn, pxn); dma_w(1, cn, wbn); }
        if (a.K <= 3) { dma_x(integral_constant<int, 3>{}, nn, cn, pxn); dma_w(2, cn, wbn); }
        if (a.K <= 4) dma_x(integral_constant<int, 4>{}, nn, cn, pxn);
      }
      {  // the last plane
        const unsigned pl = ((a.K - 1) & 1) ? py : px;
        st_contract<NB, PREC>(smem, pl, wb + (unsigned)((a.K - 1) * NB * 2048), mb0, mb1, lane, acc);
      }
      ST_STAMP(16);
      stored = c == a.C - 1;
      if (stored) {  // y of this map, straight from the accumulators
        float* __restrict__ yt = a.y + ((int64_t)n * a.y_rows + row0) * a.ld;
        // the plane that does not hold T_{K-1}: its last readers passed the barrier above
        unsigned char* scr = smem + (((a.K - 1) & 1) ? px : py) + wave * ST_SCR_WAVE;
        // one uniform switch per map, not one per element (the inlined activation switch is 5 k instructions otherwise)
        if (!vec_ok) st_store<NB, -1, false>(acc, scr, yt, a.ld, sBias, wave, lane, a.Fout, a.act);
        else if (a.act == DSPH_ACT_NONE) st_store<NB, DSPH_ACT_NONE, true>(acc, scr, yt, a.ld, sBias, wave, lane, a.Fout, a.act);
        else if (a.act == DSPH_ACT_RELU) st_store<NB, DSPH_ACT_RELU, true>(acc, scr, yt, a.ld, sBias, wave, lane, a.Fout, a.act);
        else st_store<NB, -1, true>(acc, scr, yt, a.ld, sBias, wave, lane, a.Fout, a.act);
      }
      ST_STAMP(17);
      px = pxn;
      wb = wbn;
      n = nn;
      c = cn;
    }
  }
}

}  // namespace dsph
