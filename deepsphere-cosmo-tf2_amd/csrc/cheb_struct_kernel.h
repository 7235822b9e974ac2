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
  int cheb;  // 1: T_k = 2 L~ T_{k-1} - T_{k-2} (k >= 2); 0: T_k = L~ T_{k-1}
};

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

// One recurrence step for this lane's 2x2 block: window from plane `pin` (byte offset), new values to `pout`.
//   FIRST: T_1 = L~ T_0, the block's own T_0 is read too (16 reads); otherwise the centre of the window is `cur`.
//   CHEB : T_k = 2 L~ T_{k-1} - T_{k-2}
template <bool FIRST, bool CHEB>
__device__ __forceinline__ void st_gather(const unsigned char* __restrict__ smem, unsigned pin, unsigned pout,
                                          const unsigned (&gb)[4], const float (&v)[4][9], float4 (&cur)[4],
                                          float4 (&prev)[4]) {
  float4 W[4][4];
#pragma unroll
  for (int wy = 0; wy < 4; ++wy)
#pragma unroll
    for (int wx = 0; wx < 4; ++wx) {
      const bool centre = (wx == 1 || wx == 2) && (wy == 1 || wy == 2);
      if (centre && !FIRST)
        W[wy][wx] = cur[(wy - 1) * 2 + (wx - 1)];
      else
        W[wy][wx] = *reinterpret_cast<const float4*>(smem + (gb[st_cell_f(wx, wy)] + pin) + st_woff(wx, wy));
    }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
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
  }
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

// y of one map for this lane's pixel: register quad tq of block b holds output channels 32 b + 8 tq + 4 h .. + 3.
template <int NB, int ACT, bool VEC>
__device__ __forceinline__ void st_store(const st_f32x16 (&acc)[NB], float* __restrict__ yp, const float* __restrict__ sBias,
                                         int mh, int Fout, int act_rt) {
  const int act = ACT >= 0 ? ACT : act_rt;
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int tq = 0; tq < 4; ++tq) {
      const int ch = 32 * b + 8 * tq + 4 * mh;
      const float4 bv = *reinterpret_cast<const float4*>(sBias + ch);
      float4 o;
      o.x = apply_act(acc[b][4 * tq + 0] + bv.x, act);
      o.y = apply_act(acc[b][4 * tq + 1] + bv.y, act);
      o.z = apply_act(acc[b][4 * tq + 2] + bv.z, act);
      o.w = apply_act(acc[b][4 * tq + 3] + bv.w, act);
      if (VEC) {
        if (ch < Fout) *reinterpret_cast<float4*>(yp + ch) = o;
      } else {
        if (ch + 0 < Fout) yp[ch + 0] = o.x;
        if (ch + 1 < Fout) yp[ch + 1] = o.y;
        if (ch + 2 < Fout) yp[ch + 2] = o.z;
        if (ch + 3 < Fout) yp[ch + 3] = o.w;
      }
    }
}

template <int NB, int PREC>
__global__ __launch_bounds__(ST_THREADS, 2) void cheb_struct_kernel(StructArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[ST_LDS_TOTAL];
  float* const sBias = reinterpret_cast<float*>(smem + ST_LDS_BIAS);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
  const unsigned y_pix = st_morton(mpx, mpy);  // row of this lane's pixel inside the tile

  // ---- DMA role: wave w issues pieces w, w+8, ...; lane l fills slot l & 3 of cell 16 i + (l >> 2) of the plane --
  constexpr int NP = (ST_DMA_PIECES + 7) / 8;  // 5
  unsigned dcell[NP];  // gx | gy << 8 | logical slot << 16 | valid << 24
#pragma unroll
  for (int s = 0; s < NP; ++s) {
    const int piece = wave + 8 * s;
    const unsigned p = 16u * piece + (lane >> 2);
    const unsigned par = p / ST_HP, rem = p % ST_HP, gy = rem / ST_P2, gxh = rem % ST_P2;
    const unsigned gx = 2 * gxh + par;
    const int lo = ST_DMAX - D, hi = ST_DMAX + ST_TILE - 1 + D;
    const bool ok = piece < ST_DMA_PIECES && gxh < ST_S / 2 && (int)gx >= lo && (int)gx <= hi && (int)gy >= lo && (int)gy <= hi;
    dcell[s] = gx | (gy << 8) | (((lane & 3u) ^ st_cell_f(gx, gy)) << 16) | ((ok ? 1u : 0u) << 24);
  }

  // tiles are dealt to XCDs in contiguous ranges (blocks b and b+8 share an XCD and its L2)
  const int G = gridDim.x, xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3;
  const int nslots = (G + 7 - xcd) / 8;
  const int t_begin = (int)((int64_t)a.ntiles * xcd / 8), t_end = (int)((int64_t)a.ntiles * (xcd + 1) / 8);
  const int items = a.N * a.C;
  const int wslice = a.K * NB * 2048;  // bytes of one slice's fragments
  const int wpieces = wslice / 1024;
  const bool vec_ok = (a.Fout % 4 == 0) && (a.ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.y) & 15) == 0);

  // x rows of the region cells this lane fetches, for the tile being PREFETCHED
  int64_t drow[NP];
  auto set_drows = [&](int tpos) {
    const unsigned row0 = (unsigned)a.tiles[tpos] * 256u;
    const unsigned X0 = st_compress(row0), Y0 = st_compress(row0 >> 1);
#pragma unroll
    for (int s = 0; s < NP; ++s) {
      const unsigned gx = dcell[s] & 255u, gy = (dcell[s] >> 8) & 255u;
      const bool ok = (dcell[s] >> 24) != 0;
      const unsigned rid = ok ? st_morton(X0 + gx - ST_DMAX, Y0 + gy - ST_DMAX) : row0;
      drow[s] = (int64_t)rid * a.Fin;
    }
  };
  // issue the DMA of item `it` of the tile whose rows are in drow[]: x slice -> plane at `pdst`, weights -> `wdst`
  auto issue_dma = [&](int it, unsigned pdst, unsigned wdst) {
    const int n = it / a.C, c = it - n * a.C;
    const float* __restrict__ xb = a.x + (int64_t)n * a.x_rows * a.Fin;
#pragma unroll
    for (int s = 0; s < NP; ++s) {
      const int ch0 = c * 16 + 4 * (int)((dcell[s] >> 16) & 3u);
      const int ch = ch0 < a.Fin ? ch0 : a.Fin - 4;  // channels past Fin meet zero weights
      if ((dcell[s] >> 24) != 0) {
        const unsigned dst = __builtin_amdgcn_readfirstlane(pdst + 1024u * (unsigned)(wave + 8 * s));
        st_glds16(xb + drow[s] + ch, dst);
      }
    }
    const unsigned char* __restrict__ wsrc = a.wfrag + (size_t)c * wslice + lane * 16;
    for (int j = wave; j < wpieces; j += 8) {
      const unsigned dst = __builtin_amdgcn_readfirstlane(wdst + 1024u * (unsigned)j);
      st_glds16(wsrc + 1024 * j, dst);
    }
  };

  st_f32x16 acc[NB];
  float v[4][9];
  float4 cur[4], prev[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) cur[p] = prev[p] = make_float4(0.f, 0.f, 0.f, 0.f);

  int t = t_begin + slot0;
  if (t >= t_end) return;
  unsigned px = 0;                    // byte offset of the X plane of the current item (plane 0 or 2); Y is plane 1
  unsigned wb = ST_LDS_W;             // weight buffer of the current item
  set_drows(t);
  issue_dma(0, px, wb);
  const bool late = wave >= 4;  // second wave of each SIMD: gathers first, contracts second

  for (; t < t_end; t += nslots) {
    const unsigned row0 = (unsigned)a.tiles[t] * 256u;
    {  // L~ values of this lane's four pixels (blocks that no step touches keep zeros and load nothing)
      const unsigned X0 = st_compress(row0), Y0 = st_compress(row0 >> 1);
      const bool ld_ok = active_at(1);
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
    for (int item = 0; item < items; ++item) {
      const int n = item / a.C, c = item - n * a.C;
      // ---- B_a: this item's x slice and weights have landed; every LDS read of the previous item is done -----
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const unsigned pxn = px ^ (2u * ST_PLANE_BYTES);
      const unsigned wb_next = wb == (unsigned)ST_LDS_W ? (unsigned)(ST_LDS_W + ST_WSLICE_BYTES) : (unsigned)ST_LDS_W;
      {  // prefetch the next item (of this tile, or the first of this workgroup's next tile)
        const bool last = item + 1 == items;
        if (!last) {
          issue_dma(item + 1, pxn, wb_next);
        } else if (t + nslots < t_end) {
          set_drows(t + nslots);
          issue_dma(0, pxn, wb_next);
        }
      }
      if (c == 0) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
      }
      const unsigned py = ST_PLANE_BYTES;  // Y plane
      // ---- interval k = 1 .. K-1: contract T_{k-1}, compute T_k -------------------------------------------------
      for (int k = 1; k < a.K; ++k) {
        const unsigned pin = (k & 1) ? px : py, pout = (k & 1) ? py : px;
        const unsigned wk = wb + (unsigned)((k - 1) * NB * 2048);
        if (!late) st_contract<NB, PREC>(smem, pin, wk, mb0, mb1, lane, acc);
        if (active_at(k)) {
          if (k == 1) st_gather<true, false>(smem, pin, pout, gb, v, cur, prev);
          else if (a.cheb) st_gather<false, true>(smem, pin, pout, gb, v, cur, prev);
          else st_gather<false, false>(smem, pin, pout, gb, v, cur, prev);
        }
        if (late) st_contract<NB, PREC>(smem, pin, wk, mb0, mb1, lane, acc);
        __syncthreads();
      }
      {  // the last plane
        const unsigned pl = ((a.K - 1) & 1) ? py : px;
        st_contract<NB, PREC>(smem, pl, wb + (unsigned)((a.K - 1) * NB * 2048), mb0, mb1, lane, acc);
      }
      if (c == a.C - 1) {  // y of this map, straight from the accumulators
        float* __restrict__ yp = a.y + ((int64_t)n * a.y_rows + row0 + y_pix) * a.ld;
        // one uniform switch per map, not one per element (the inlined activation switch is 5 k instructions otherwise)
        if (!vec_ok) st_store<NB, -1, false>(acc, yp, sBias, (int)mh, a.Fout, a.act);
        else if (a.act == DSPH_ACT_NONE) st_store<NB, DSPH_ACT_NONE, true>(acc, yp, sBias, (int)mh, a.Fout, a.act);
        else if (a.act == DSPH_ACT_RELU) st_store<NB, DSPH_ACT_RELU, true>(acc, yp, sBias, (int)mh, a.Fout, a.act);
        else st_store<NB, -1, true>(acc, yp, sBias, (int)mh, a.Fout, a.act);
      }
      px = pxn;
      wb = wb_next;
    }
  }
}

}  // namespace dsph
