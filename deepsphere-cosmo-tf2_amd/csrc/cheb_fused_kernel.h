// Device side of the fused Chebyshev forward (see cheb_fused.hip for the design notes).
// Included by cheb_fused.hip (host: tile tables, dispatch) and by cheb_fused_inst.hip, which is
// compiled once per (plane rows, ELL width) pair so that the instantiations build in parallel.
#pragma once

#include <type_traits>

#include "dsphere_common.h"

namespace dsph {

constexpr int FUSED_P = 256;        // rows per tile
constexpr int FUSED_CH = 16;        // channels per slice
constexpr int FUSED_DMAX = 9;       // deepest halo supported (K <= 10: the 9-ring region of a 16 x 16 tile is 34 x 34 = 1,156 rows, two planes of 1,168 fill the LDS)
constexpr int FUSED_THREADS = 512;  // 8 waves, 2 per SIMD
constexpr int LDS_BYTES = 160 * 1024;
constexpr int FUSED_BIAS_BYTES = 256;  // the bias (<= 64 floats, zero-padded) sits in the last bytes of the LDS
constexpr int G_ROWS = FUSED_THREADS / 4;  // recurrence: four lanes share a region row, 128 rows per pass

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int WG_TILE_BYTES = 8 * 1024;  // weight-gradient mode: one 16 x 16 fp32 accumulator tile per wave, in LDS
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct FusedArgs {
  const float* x;
  const float* bias;
  float* y;
  const unsigned char* wfrag;
  const int32_t* tile_list;  // optional indirection: this launch handles tiles tile_list[0 .. ntiles) (NULL: 0 .. ntiles)
  const int32_t* tile_off;
  const int32_t* ring_end;
  const int64_t* ell_off;
  const int32_t* region;
  const uint16_t* lcols;
  const float* lvals;
  float* planes_out;      // planes mode: T_1 .. T_{K-1}, each (N, prow_stride, Fin), written on the tile rows
  int64_t plane_stride;   //   elements per plane
  int64_t prow_stride;    //   rows per map of a plane
  const float* dy;        // weight-gradient mode: upstream gradient (N, y_rows, Fout)
  float* slabs;           //   per-(workgroup, pixel half) partial sums [2*grid][nchunks][K][16][64]
  int c_begin, c_count;   // slices handled by this launch (all modes; forward and planes: 0, nchunks)
  int64_t x_rows, y_rows;
  int N, Fin, Fout, K, ntiles, nchunks, act, wfrag_bytes;
  int ld;  // row stride (floats) of y, dy: Fout of the whole layer when this launch handles one 64-column block of it
  float alpha_rest, beta_rest;  // step k >= 2: T_k = alpha * L~ T_{k-1} - beta * T_{k-2} (2,1 Chebyshev; 1,0 monomial)
#ifdef DSPH_STAMPS
  unsigned long long* stamps;  // diagnostic build only: [8 waves][8 items][32 points] s_memtime values
#endif
  // Forward of a layer with at most FOUR input channels and at most 16 output columns (a network's first layers): FOUR MAPS per
  // item (pack = 4).  The 16-channel slice the recurrence works on is [map 4 n | 4 n + 1 | 4 n + 2 | 4 n + 3] x 4 channels instead
  // of 4 channels + 12 zeros (the recurrence is per channel: nothing changes), the weight image is block diagonal -- inner
  // index 4 q + c against columns 16 q .. 16 q + 15 -- so the 64-column contraction leaves map 4 n + q's 16 columns in columns
  // 16 q ..; the store sends each 16-column group to its own map.  N is then the number of GROUPS, n_maps the batch.
  // pack = 2: eight input channels, at most 32 columns, two maps of two slots and 32 columns each.
  int pack, n_maps;
  // conv + HealpyPool(p = 1) in one forward (pool = 1 max, 2 mean): a tile is 256 consecutive NEST rows, the four children of a
  // coarse pixel four consecutive rows of the wave's 32 x 32 block -- the store reduces them and writes the pooled map only
  float* ypool;
  int64_t ypool_rows;
  int pool;
  int num_cu;  // (host side: CUs of the device, for the split of a small map's batch over gridDim.y)
  int dbg;  // timing-only ablation bits (DSPH_FUSED_DEBUG): 1 no recurrence, 2 no contraction, 8 no y store
};

// Byte offset of 16-byte slot `slot` (0..3) of region row `row` inside a [rows][16] fp32 plane.
// A row is 64 B, so four consecutive rows fill the 256-byte LDS bank row; XOR-ing the slot with
// bits 2..3 of the row makes 16 lanes that read one slot of 16 consecutive rows (the MFMA operand
// read) hit 16 different 16-byte bank groups.  The recurrence reads with four lanes per row (one
// slot each), which is conflict-free whenever the four rows of a 16-lane group differ mod 4.
__device__ __forceinline__ unsigned plane_byte(unsigned row, unsigned slot) {
  return row * (FUSED_CH * 4) + 16u * (slot ^ ((row >> 2) & 3u));
}

// The B operand of one plane's MFMAs: the weight fragments of (order k, slice c), already in operand
// order in LDS.  Read into registers *before* the barrier that completes the plane (they do not depend
// on it), so that after the barrier only the A rows stand between the wave and its first MFMA.
template <int NB, int PREC>
struct WFrag {
  bf16x8 hi[NB], lo[NB];  // PREC == BF16X3
  float f[8][NB];         // PREC == FP32
};

template <int NB, int PREC>
__device__ __forceinline__ void load_wfrag(const unsigned char* __restrict__ wblk, int lane, WFrag<NB, PREC>& w) {
  if (PREC == DSPH_PREC_BF16X3) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      w.hi[b] = *reinterpret_cast<const bf16x8*>(wblk + b * 2048 + lane * 16);
      w.lo[b] = *reinterpret_cast<const bf16x8*>(wblk + b * 2048 + 1024 + lane * 16);
    }
  } else {
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int b = 0; b < NB; ++b) w.f[t][b] = *reinterpret_cast<const float*>(wblk + b * 2048 + t * 256 + lane * 4);
  }
}

// One plane T_k (tile rows 32*wave .. +32, all 16 channels of the slice) into the MFMA
// accumulators: A from the LDS plane, converted on the fly to split bf16 when PREC says so.
template <int NB, int PREC>
__device__ __forceinline__ void mfma_plane(const unsigned char* __restrict__ plane, const WFrag<NB, PREC>& w,
                                           int wave, int lane, f32x16 (&acc)[NB]) {
  const unsigned r = lane & 31, h = lane >> 5;
  const unsigned row = wave * 32 + r;
  const float4 a0 = *reinterpret_cast<const float4*>(plane + plane_byte(row, 2 * h));
  const float4 a1 = *reinterpret_cast<const float4*>(plane + plane_byte(row, 2 * h + 1));
  const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
  if (PREC == DSPH_PREC_BF16X3) {
    bf16x8 ahi, alo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const __bf16 hi = (__bf16)av[j];
      ahi[j] = hi;
      alo[j] = (__bf16)(av[j] - (float)hi);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, w.hi[b], acc[b], 0, 0, 0);  // small terms first
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, w.lo[b], acc[b], 0, 0, 0);
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, w.hi[b], acc[b], 0, 0, 0);
    }
  } else {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], w.f[t][b], acc[b], 0, 0, 0);
    }
  }
}

// Orders one wave's LDS writes before its own later LDS reads (the block a wave transposes is private
// to it).  Only the LDS counter is waited on: a workgroup-scope fence here also drains vmcnt, i.e. it
// stalls on the x prefetch just issued and on the previous block's y stores (7 k cycles per map).
__device__ __forceinline__ void lds_wave_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// The four lanes that share a region row keep that row's ELL values spread over the quad: register g
// of lane q holds val[4g + q].  A quad-broadcast DPP move fetches val[j] for all four lanes, so a row
// costs ceil(WT/4) value registers per lane instead of WT.
__device__ __forceinline__ float quad_bcast(float v, int lane_in_quad) {
  // dpp_ctrl quad_perm:[b,b,b,b] = b * 0x55; all rows and banks enabled
  const int iv = __builtin_bit_cast(int, v);
  int r;
  switch (lane_in_quad) {
    case 0: r = __builtin_amdgcn_update_dpp(0, iv, 0x00, 0xf, 0xf, true); break;
    case 1: r = __builtin_amdgcn_update_dpp(0, iv, 0x55, 0xf, 0xf, true); break;
    case 2: r = __builtin_amdgcn_update_dpp(0, iv, 0xAA, 0xf, 0xf, true); break;
    default: r = __builtin_amdgcn_update_dpp(0, iv, 0xFF, 0xf, 0xf, true); break;
  }
  return __builtin_bit_cast(float, r);
}

// One recurrence step on rows [0, nrows): out = L~ in (CHEB_STEP false: T_1 of either basis and every
// monomial step) or out = 2 (L~ in) - out (CHEB_STEP true: Chebyshev steps k >= 2).
// Lane (row_l, slot) owns rows row_l + 128 p, p = 0..RP-1, and one 16-byte slot (4 channels) of them;
// the ELL values (quad-packed) and the swizzled LDS addresses of those rows' neighbours live in
// registers.  The summation order (slot j ascending, fused multiply-add) is the unfused kernel's.
// Every lane of a quad must execute the broadcasts, so the row guard covers whole quads (it does:
// the four lanes of a quad share the row).
// SAVE: also write the new plane's tile rows (the first two passes: rows < save_rows <= 256) to global
// memory at `save` (this lane's row_l and channel slot already added; 128 rows = save_pass floats apart).
// VW: value registers per row -- WT (one per neighbour) or ceil(WT / 4) (quad-packed, fetched by DPP broadcast).
template <int WT, int RP, bool CHEB_STEP, bool SAVE = false, int VW = WT>
__device__ __forceinline__ void gather_step(const unsigned char* __restrict__ pin,
                                            unsigned char* __restrict__ pout, int nrows, int row_l,
                                            const float (&valc)[RP][VW], const unsigned (&pre)[RP][WT],
                                            const unsigned (&own)[RP], float* __restrict__ save = nullptr,
                                            int64_t save_pass = 0, int save_rows = 0) {
#pragma unroll
  for (int p = 0; p < RP; ++p) {
    if (row_l + p * G_ROWS < nrows) {
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int j = 0; j < WT; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(pin + pre[p][j]);
        const float w = VW == WT ? valc[p][VW == WT ? j : 0] : quad_bcast(valc[p][VW == WT ? 0 : j >> 2], j & 3);
        s.x = fmaf(w, v.x, s.x);
        s.y = fmaf(w, v.y, s.y);
        s.z = fmaf(w, v.z, s.z);
        s.w = fmaf(w, v.w, s.w);
      }
      float4* op = reinterpret_cast<float4*>(pout + own[p]);
      if (CHEB_STEP) {
        const float4 q = *op;
        s.x = 2.f * s.x - q.x;
        s.y = 2.f * s.y - q.y;
        s.z = 2.f * s.z - q.z;
        s.w = 2.f * s.w - q.w;
      }
      *op = s;
      if (SAVE && p * G_ROWS < FUSED_P && row_l + p * G_ROWS < save_rows)
        *reinterpret_cast<float4*>(save + p * save_pass) = s;
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the passes apart: 9 gathers in flight, not 9*RP
  }
}

// Weight-gradient mode: dW[f, k, o] += sum_pixels T_k[pixel, f] * dy[pixel, o] for one plane T_k (16 channels)
// of one tile.  v_mfma_f32_16x16x4_f32 with the pixels as the inner dimension: lane (i = l & 15, kk = l >> 4)
// supplies A[i][kk] = T_k[pixel(g, kk)][channel i] -- one ds_read_b32 straight from the row-major plane, four
// consecutive rows per instruction, conflict-free -- and B[kk][j] = dy[pixel(g, kk)][16 nb + j], which the wave
// keeps in registers for the whole map (32 pixel groups of 4 = its 128-pixel half of the tile).
// The wave's 16 x 16 accumulator tile of (slice, order) lives in LDS between visits (`tile`: 16 bytes per
// lane): 40 accumulator registers next to the recurrence's working set spilled.
__device__ __forceinline__ void wgrad_plane(unsigned char* __restrict__ tile, const unsigned char* __restrict__ smem,
                                            const unsigned (&la)[4], unsigned plane_off, const float (&dyv)[32]) {
  // 8 A values in flight while the previous 8 MFMAs run: 16 registers, not 32
  float av[2][8];
  f32x4 t = *reinterpret_cast<const f32x4*>(tile);
#pragma unroll
  for (int j = 0; j < 8; ++j) av[0][j] = *reinterpret_cast<const float*>(smem + plane_off + la[j & 3] + 256 * j);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (q < 3) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int g = 8 * (q + 1) + j;
        av[(q + 1) & 1][j] = *reinterpret_cast<const float*>(smem + plane_off + la[g & 3] + 256 * g);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) t = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q & 1][j], dyv[8 * q + j], t, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  *reinterpret_cast<f32x4*>(tile) = t;
}

// The same contraction on the bf16 matrix pipe: v_mfma_f32_16x16x32_bf16, 32 pixels per MFMA, both operands
// split hi + lo and three MFMAs per step (lo*hi, hi*lo, hi*hi), fp32 accumulate -- 12 MFMAs of 16 cycles per
// plane and wave instead of 32 of 32.  Inner index e = 8 q + j of lane (i, q = l >> 4) is pixel 32 st + 4 j + q:
// the A values are the very ds_read_b32 of the fp32 version (g = 8 st + j), the B values its dyv[g].
struct DyFrag {
  bf16x8 hi[4], lo[4];
};

__device__ __forceinline__ void split_bf16(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 h = (__bf16)v[j];
    hi[j] = h;
    lo[j] = (__bf16)(v[j] - (float)h);
  }
}

__device__ __forceinline__ void wgrad_plane_bf16(unsigned char* __restrict__ tile, const unsigned char* __restrict__ smem,
                                                 const unsigned (&la)[4], unsigned plane_off, const DyFrag& dy) {
  float av[2][8];
  f32x4 t = *reinterpret_cast<const f32x4*>(tile);
#pragma unroll
  for (int j = 0; j < 8; ++j) av[0][j] = *reinterpret_cast<const float*>(smem + plane_off + la[j & 3] + 256 * j);
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    if (st < 3) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int g = 8 * (st + 1) + j;
        av[(st + 1) & 1][j] = *reinterpret_cast<const float*>(smem + plane_off + la[g & 3] + 256 * g);
      }
    }
    bf16x8 ahi, alo;
    split_bf16(av[st & 1], ahi, alo);
    t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(alo, dy.hi[st], t, 0, 0, 0);  // small terms first
    t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi, dy.lo[st], t, 0, 0, 0);
    t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi, dy.hi[st], t, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  *reinterpret_cast<f32x4*>(tile) = t;
}

// Diagnostic build (make STAMPS=1; never the shipped library): s_memtime at the phase boundaries of a few
// items of one workgroup, into a buffer nothing else reads.  Read the shares, not the run time.
#ifdef DSPH_STAMPS
#define DSPH_STAMP(id)                                                                        \
  do {                                                                                        \
    if (stamp_on) {                                                                           \
      __builtin_amdgcn_sched_barrier(0);                                                      \
      unsigned long long t_;                                                                  \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");              \
      __builtin_amdgcn_sched_barrier(0);                                                      \
      if (lane == 0) a.stamps[((size_t)wave * 8 + (item - 4)) * 32 + (id)] = t_;              \
    }                                                                                         \
  } while (0)
#else
#define DSPH_STAMP(id)
#endif

// PR: rows each LDS plane is sized for; RP: recurrence rows per lane (rows with an ELL row <= 128*RP);
// WT: ELL width; NB: 32-column output blocks; PREC: contraction arithmetic.
// MODE 0: the forward.  MODE 1 ("planes"): no contraction at all -- the recurrence alone, with the tile rows of
// T_1 .. T_{K-1} written to a.planes_out (the weight gradient's left operand, rebuilt in the backward pass
// at fused speed instead of by K-1 gather launches that go through L2 for every neighbour).
// MODE 2 ("weight gradient"): the recurrence, and every plane contracted over the tile's pixels against dy
// (wgrad_batch); the partial sums stay in registers over all tiles of the workgroup and are written once,
// to a.slabs, for a fixed-order second stage.  No plane ever leaves the LDS.
// WG: the weight fragments of all slices do not fit the LDS beside the planes (more than 64 input channels at K = 5): every
// level's fragments are then read from the packed image in global memory (L2-resident, at most a few hundred KB) as they are
// needed, instead of from an LDS copy made once per workgroup.
// PX: the forward packs maps (FusedArgs::pack) or pools in its store (FusedArgs::pool); without it neither path exists in the code --
// the K = 6 .. 9 instantiations are at the register limit, the two paths cost them 13 more spilled registers (BASELINE configs[3]
// 21.0 -> 23.3 ms when they were run-time branches)
template <int PR, int WT, int RP, int NB, int PREC, int MODE = 0, bool WG = false, bool PX = false>
__global__ __launch_bounds__(FUSED_THREADS, 2) void cheb_fused_kernel(FusedArgs a) {
  const int pack_ = PX ? a.pack : 0, pool_ = PX ? a.pool : 0;
  constexpr int PLANE_BYTES = PR * FUSED_CH * 4;
  constexpr int NS = (PR * 4 + FUSED_THREADS - 1) / FUSED_THREADS;  // staging float4 per lane
  // all of the CU's LDS, statically: the base is then a compile-time constant that folds into the
  // ds_read/ds_write offset fields (a dynamic LDS symbol costs one v_add per access)
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
  unsigned char* const planeX = smem;
  unsigned char* const planeY = smem + PLANE_BYTES;
  unsigned char* const sW = smem + 2 * PLANE_BYTES;
  float* const sBias = reinterpret_cast<float*>(smem + LDS_BYTES - FUSED_BIAS_BYTES);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // the epilogue must not touch vector memory for anything but its stores: a bias load there makes
  // the compiler wait for vmcnt(0), i.e. for the x prefetch just issued and the previous y stores
  if (MODE == 2) {  // rows of a ragged tile that no step writes are multiplied by dy = 0: they must be finite
    for (int i = tid * 16; i < 2 * PLANE_BYTES; i += FUSED_THREADS * 16) *reinterpret_cast<uint4*>(smem + i) = uint4{0, 0, 0, 0};
  }
  if (MODE == 0 && tid < FUSED_BIAS_BYTES / 4) {
    const int bc = pack_ == 4 ? tid & 15 : (pack_ == 2 ? tid & 31 : tid);  // (packed maps: every column group carries the layer's columns)
    sBias[tid] = (a.bias != nullptr && bc < a.Fout) ? a.bias[bc] : 0.f;
  }
  if (MODE == 0 && !WG) {
    for (int i = tid * 16; i < a.wfrag_bytes; i += FUSED_THREADS * 16)
      *reinterpret_cast<uint4*>(sW + i) = *reinterpret_cast<const uint4*>(a.wfrag + i);
  }

  // tiles are dealt to XCDs in contiguous ranges (blocks b and b+8 share an XCD): the 32
  // workgroups of one XCD work on 32 neighbouring tiles at a time and share halos through its L2
  const int G = gridDim.x, xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3;
  const int nslots = (G + 7 - xcd) / 8;
  const int t_begin = (int)((int64_t)a.ntiles * xcd / 8), t_end = (int)((int64_t)a.ntiles * (xcd + 1) / 8);
  const int D = a.K - 1;
  const int row_l = tid >> 2;      // recurrence: this lane's row within a pass
  const unsigned qslot = tid & 3;  //             and its 16-byte slot
  // (small maps have fewer tiles than the device has CUs: the forward then splits the maps of the batch over gridDim.y workgroups)
  const int n_first = (int)((int64_t)a.N * blockIdx.y / gridDim.y), n_end = (int)((int64_t)a.N * (blockIdx.y + 1) / gridDim.y);
  if (n_first >= n_end) return;
  const int items = (n_end - n_first) * a.c_count;  // (map, slice) pairs per tile
  const size_t wstride = (size_t)a.nchunks * NB * 2048;  // weight blocks: per order
  const bool do_g = !(a.dbg & 1), do_m = !(a.dbg & 2);
  const bool cheb = a.beta_rest != 0.f;  // Chebyshev (2, 1) or monomial (1, 0) steps from k = 2 on

  // ---- software prefetch of the next (tile, map, slice): region row ids and x in registers ----
  int rid[NS];
  float4 pf[NS];
  // Both are branch-free on purpose: a load under a lane predicate lands in a temporary that is merged
  // into pf[] by register moves behind an s_waitcnt vmcnt(0), i.e. the "prefetch" then waits for itself
  // (1.3-2.5 k cycles per slice).  Rows past the region and channels past Fin read a valid address
  // instead (the former are never used, the latter are zeroed when the slice is staged).
  auto load_rids = [&](int tv) {
    const int pos = __builtin_amdgcn_readfirstlane(tv);  // uniform: the table reads below stay scalar loads
    const int t = a.tile_list ? a.tile_list[pos] : pos;
    const int off = a.tile_off[t];
    const int R = a.ring_end[(size_t)t * (FUSED_DMAX + 1) + D];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int row = (tid + s * FUSED_THREADS) >> 2;
      rid[s] = a.region[off + (row < R ? row : R - 1)];
    }
  };
  // Slice `item`'s loads, one float4 per lane and slot s: all of them (slot < 0) or those of one phase.
  auto issue_loads = [&](int item, int slot) {
    const int ni = item / a.c_count, n = n_first + ni, c = a.c_begin + item - ni * a.c_count;
    const int ch0 = c * FUSED_CH + 4 * (tid & 3);
    const int ch = pack_ == 4 ? 0 : (pack_ == 2 ? 4 * (tid & 1) : (ch0 < a.Fin ? ch0 : a.Fin - 4));
    // (packed maps: this lane's 16-byte slot belongs to map P n + slot / (4 / P) -- the last map again where the batch ends
    // inside the group)
    const int nx = pack_ ? min(pack_ * n + (pack_ == 4 ? tid & 3 : (tid & 3) >> 1), a.n_maps - 1) : n;
    const float* __restrict__ xb = a.x + (int64_t)nx * a.x_rows * a.Fin + ch;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (slot >= 0 && (s < a.K ? s : 0) != slot) continue;
      pf[s] = *reinterpret_cast<const float4*>(xb + (int64_t)rid[s] * a.Fin);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- y store of a finished map, deferred into the next item's slot 0 ---------------------------
  // The accumulator tile (column per lane, rows in registers) goes through LDS so that a store
  // instruction writes whole 128-byte row halves, 16 bytes per lane (a row-per-lane dword epilogue is
  // store-issue bound: it cost 8 of 28 ms).  It runs after the next slice has been staged and the
  // prefetch after that has been issued, in plane Y (free during slot 0), 32 columns at a time: the
  // stores then have a whole item to drain before the next wait on the memory counter, instead of
  // stalling the staging that follows them.
  f32x16 acc[NB];
  bool pend = false;
  int pend_n = 0, pend_Pt = 0;
  int64_t pend_row0 = 0;
  auto store_impl = [&](auto act_c, auto vec_c) {
    constexpr int ACT = decltype(act_c)::value;   // compile-time activation, or -1: a.act at run time
    constexpr bool VEC = decltype(vec_c)::value;  // Fout % 4 == 0 and y 16-byte aligned
    constexpr int T_LD = 36;  // padded row (floats) of a wave's 32 x 32 block; 8 waves fill plane Y exactly
    float* __restrict__ tw = reinterpret_cast<float*>(planeY) + wave * (32 * T_LD);
    const int li = lane & 31, h = lane >> 5;
    const int cq0 = (lane & 7) * 4, rsub = lane >> 3;
    const int act = ACT >= 0 ? ACT : a.act;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
      for (int q = 0; q < 16; ++q) tw[((q & 3) + 8 * (q >> 2) + 4 * h) * T_LD + li] = acc[b][q];
      lds_wave_sync();
      const int cq = 32 * b + cq0;
      const float4 bv = *reinterpret_cast<const float4*>(sBias + cq);
      const int gsh = pack_ == 4 ? 4 : 5;  // log2 of the columns per map
      const int ymap = pack_ ? pack_ * pend_n + (cq >> gsh) : pend_n, ycol = pack_ ? cq & ((1 << gsh) - 1) : cq;
      const bool ylive = !pack_ || ymap < a.n_maps;
      if (VEC && pool_) {
        // pooled row rsub of this wave's eight: rows 4 rsub .. 4 rsub + 3 of the block, bias and activation first, in row order
        float4 o = pool_ == 1 ? make_float4(-__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf())
                               : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float4 v = *reinterpret_cast<const float4*>(tw + (4 * rsub + q) * T_LD + cq0);
          v.x = apply_act(v.x + bv.x, act);
          v.y = apply_act(v.y + bv.y, act);
          v.z = apply_act(v.z + bv.z, act);
          v.w = apply_act(v.w + bv.w, act);
          if (pool_ == 1) { o.x = fmaxf(o.x, v.x); o.y = fmaxf(o.y, v.y); o.z = fmaxf(o.z, v.z); o.w = fmaxf(o.w, v.w); }
          else { o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w; }
        }
        if (pool_ != 1) { o.x *= 0.25f; o.y *= 0.25f; o.z *= 0.25f; o.w *= 0.25f; }
        float* __restrict__ ypp = a.ypool + ((int64_t)(ylive ? ymap : 0) * a.ypool_rows + ((pend_row0 + wave * 32) >> 2) + rsub) * a.ld + ycol;
        if (wave * 32 + 4 * rsub + 3 < pend_Pt && ylive && ycol < a.Fout) *reinterpret_cast<float4*>(ypp) = o;
        lds_wave_sync();
        continue;
      }
      float* __restrict__ yp0 = a.y + ((int64_t)(ylive ? ymap : 0) * a.y_rows + pend_row0 + wave * 32 + rsub) * a.ld + ycol;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = i * 8 + rsub;
        float4 v = *reinterpret_cast<const float4*>(tw + row * T_LD + cq0);
        v.x = apply_act(v.x + bv.x, act);
        v.y = apply_act(v.y + bv.y, act);
        v.z = apply_act(v.z + bv.z, act);
        v.w = apply_act(v.w + bv.w, act);
        float* __restrict__ yp = yp0 + (int64_t)(i * 8) * a.ld;
        if (wave * 32 + row < pend_Pt && ylive) {
          if (VEC) {
            if (ycol < a.Fout) *reinterpret_cast<float4*>(yp) = v;
          } else {
            if (ycol + 0 < a.Fout) yp[0] = v.x;
            if (ycol + 1 < a.Fout) yp[1] = v.y;
            if (ycol + 2 < a.Fout) yp[2] = v.z;
            if (ycol + 3 < a.Fout) yp[3] = v.w;
          }
        }
      }
      lds_wave_sync();
    }
    pend = false;
  };
  // One uniform switch per map instead of one per element: with the activation switch inlined 128 times
  // the epilogue was 5 k instructions of branches and took 7 k cycles per map (mostly instruction fetch).
  const bool vec_ok = (a.Fout % 4 == 0) && (a.ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.y) & 15) == 0);
  auto store_pending = [&]() {
    using std::integral_constant;
    if (!vec_ok) return store_impl(integral_constant<int, -1>{}, integral_constant<bool, false>{});
    switch (a.act) {
      case DSPH_ACT_NONE: return store_impl(integral_constant<int, DSPH_ACT_NONE>{}, integral_constant<bool, true>{});
      case DSPH_ACT_RELU: return store_impl(integral_constant<int, DSPH_ACT_RELU>{}, integral_constant<bool, true>{});
      default: return store_impl(integral_constant<int, -1>{}, integral_constant<bool, true>{});
    }
  };

  // ---- weight-gradient mode: this wave's share is output columns 16 nb .. +16 and pixel half hp ----
  float dyv[32];  // PREC == FP32: this wave's dy values of the current map
  DyFrag dyf;     // PREC == BF16X3: the same, split
  unsigned la[4];
  const int wg_nb = wave & 3, wg_hp = wave >> 2, wg_i = lane & 15, wg_kk = lane >> 4;
  unsigned char* const sAcc = smem + 2 * PLANE_BYTES + tid * 16;  // tile i of this lane: + i * WG_TILE_BYTES
  if (MODE == 2) {
    for (int i = 0; i < a.c_count * a.K; ++i) *reinterpret_cast<f32x4*>(sAcc + i * WG_TILE_BYTES) = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j)  // the swizzle term of row 128 hp + 4 g + kk is g & 3
      la[j] = (unsigned)(wg_hp * 128 + wg_kk) * (FUSED_CH * 4) + 16u * ((unsigned)(wg_i >> 2) ^ (unsigned)j) + 4u * (wg_i & 3);
  }

  int t = t_begin + slot0;
  if (t < t_end) {
    load_rids(t);
    issue_loads(0, -1);
    // from here on rid[] holds the rows of the tile of the slice after the one being loaded
    load_rids(items >= 2 || t + nslots >= t_end ? t : t + nslots);
  }
  for (; t < t_end; t += nslots) {
    // ring sizes of this tile, 11 bits each, in two scalar registers (re-reading them from memory
    // in every step would put a dependent scalar load in front of each recurrence step)
    const int tt = a.tile_list ? a.tile_list[t] : t;  // t: position in this launch's list; tt: the tile
    const int32_t* __restrict__ re_mem = a.ring_end + (size_t)tt * (FUSED_DMAX + 1);
    unsigned long long re_lo = 0, re_hi = 0;
#pragma unroll
    for (int r = 0; r <= FUSED_DMAX; ++r) {
      const unsigned long long v = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(re_mem[r]) & 0x7ffull;
      if (r < 5) re_lo |= v << (11 * r);
      else re_hi |= v << (11 * (r - 5));
    }
    auto re = [&](int r) -> int {
      return (int)(((r < 5 ? re_lo >> (11 * r) : re_hi >> (11 * (r - 5)))) & 0x7ffull);
    };
    const int P_t = re(0), E = re(D - 1);
    const int64_t lbase = a.ell_off[tt] * WT;
    const int64_t row0 = (int64_t)tt * FUSED_P;

    // this lane's recurrence rows: ELL values and swizzled LDS byte addresses stay in registers
    // One value register per neighbour (9 per row) where the registers allow it: the quad-packed form (3 per
    // row, register g of lane q holds value 4g + q) costs one DPP move per neighbour, 15 % of the kernel's VALU
    // instructions (17.9 vs 18.3 ms same-box).  The weight-gradient mode keeps it (its dy fragments need the room),
    // and so do the variants with more passes or wider rows.
    constexpr int VW = (MODE == 2 || RP * WT > 36) ? (WT + 3) / 4 : WT;  // unpacked only for the 9-wide, 4-pass variant
    float val[RP][VW];
    unsigned pre[RP][WT], own[RP];
#pragma unroll
    for (int p = 0; p < RP; ++p) {
      const int i = row_l + p * G_ROWS;
      own[p] = plane_byte((unsigned)(i < PR ? i : 0), qslot);
#pragma unroll
      for (int g = 0; g < VW; ++g) {
        const int j = VW == WT ? g : 4 * g + (int)qslot;
        val[p][g] = (i < E && j < WT) ? a.lvals[lbase + (int64_t)j * E + i] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < WT; ++j) {
        const unsigned c = i < E ? a.lcols[lbase + (int64_t)j * E + i] : 0u;
        pre[p][j] = plane_byte(c, qslot);
      }
    }

    for (int item = 0; item < items; ++item) {
      const int ni = item / a.c_count, n = n_first + ni, c = a.c_begin + item - ni * a.c_count;
#ifdef DSPH_STAMPS
      const bool stamp_on = blockIdx.x == 72 && t == t_begin + slot0 + nslots && item >= 4 && item < 12;
#endif
      DSPH_STAMP(0);
      __syncthreads();  // the previous slice's last plane is still being read
      DSPH_STAMP(1);
      // ---- T_0: the prefetched x slice goes to plane X; fetch the next slice meanwhile -------
      const bool ch_ok = pack_ || c * FUSED_CH + 4 * (tid & 3) < a.Fin;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const unsigned idx = tid + s * FUSED_THREADS;
        float4 v = pf[s];
        v.x = ch_ok ? v.x : 0.f;
        v.y = ch_ok ? v.y : 0.f;
        v.z = ch_ok ? v.z : 0.f;
        v.w = ch_ok ? v.w : 0.f;
        if (idx < (unsigned)PR * 4) *reinterpret_cast<float4*>(planeX + plane_byte(idx >> 2, idx & 3)) = v;
      }
      // the region rows fetched at the end of the previous slice are "used" here, next to the wait the
      // staging needed anyway; otherwise each phase's load waits vmcnt(0) for them, i.e. for the load before it
#pragma unroll
      for (int s = 0; s < NS; ++s) asm volatile("" : "+v"(rid[s]));
      DSPH_STAMP(2);
      // The next slice (of this tile, or the first one of this workgroup's next tile): its loads go out
      // one per phase, unconditionally -- after the very last slice they re-read valid rows for nothing --
      // because a conditional load is merged into pf[] by moves behind a vmcnt(0).
      const int nitem = item + 1 < items ? item + 1 : 0;
      issue_loads(nitem, 0);
      DSPH_STAMP(3);
      if (MODE == 0) {
      if (pend && !(a.dbg & 8)) store_pending();
      DSPH_STAMP(4);
      if (c == 0) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int q = 0; q < 16; ++q) acc[b][q] = 0.f;
      }
      DSPH_STAMP(5);
      // (WG: a level's fragments come from global memory behind that level's x prefetch, and the memory counter is in order --
      // the wait for them at the level's barrier is a trip to HBM at every level.  Requesting them a level ahead needs a second
      // set of fragment registers: tried in round 6 on the 1,168-row variant, 24 -> 59 spilled registers and K = 10 slower, 2.26
      // against 2.10 ms; 64 -> 64: 7.1 against 4.6)
      const unsigned char* __restrict__ wblk = (WG ? a.wfrag : sW) + (size_t)c * NB * 2048;
      WFrag<NB, PREC> wf;
      load_wfrag<NB, PREC>(wblk, lane, wf);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      DSPH_STAMP(6);
      if (do_m) mfma_plane<NB, PREC>(planeX, wf, wave, lane, acc);
      DSPH_STAMP(7);

      // ---- recurrence, two steps per trip so that the plane roles are compile-time ----------
      issue_loads(nitem, 1);
      if (do_g) gather_step<WT, RP, false, false, VW>(planeX, planeY, re(D - 1), row_l, val, pre, own);
      DSPH_STAMP(8);
      load_wfrag<NB, PREC>(wblk + wstride, lane, wf);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      DSPH_STAMP(9);
      if (do_m) mfma_plane<NB, PREC>(planeY, wf, wave, lane, acc);
      DSPH_STAMP(10);
      for (int k = 2; k < a.K; k += 2) {
        issue_loads(nitem, k);
        if (do_g) {
          if (cheb) gather_step<WT, RP, true, false, VW>(planeY, planeX, re(D - k), row_l, val, pre, own);
          else gather_step<WT, RP, false, false, VW>(planeY, planeX, re(D - k), row_l, val, pre, own);
        }
        DSPH_STAMP(11 + (k - 2) * 3);
        load_wfrag<NB, PREC>(wblk + (size_t)k * wstride, lane, wf);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        DSPH_STAMP(12 + (k - 2) * 3);
        if (do_m) mfma_plane<NB, PREC>(planeX, wf, wave, lane, acc);
        DSPH_STAMP(13 + (k - 2) * 3);
        if (k + 1 < a.K) {
          issue_loads(nitem, k + 1);
          if (do_g) {
            if (cheb) gather_step<WT, RP, true, false, VW>(planeX, planeY, re(D - k - 1), row_l, val, pre, own);
            else gather_step<WT, RP, false, false, VW>(planeX, planeY, re(D - k - 1), row_l, val, pre, own);
          }
          DSPH_STAMP(14 + (k - 2) * 3);
          load_wfrag<NB, PREC>(wblk + (size_t)(k + 1) * wstride, lane, wf);
          __builtin_amdgcn_sched_barrier(0);
          __syncthreads();
          DSPH_STAMP(15 + (k - 2) * 3);
          if (do_m) mfma_plane<NB, PREC>(planeY, wf, wave, lane, acc);
          DSPH_STAMP(16 + (k - 2) * 3);
        }
      }
      } else if (MODE == 2) {
        const int cl = c - a.c_begin;
        if (cl == 0) {  // a new map: this wave's dy fragments (branch-free loads, zero past the tile / Fout)
          const int o = 16 * wg_nb + wg_i;
          const int oc = o < a.Fout ? o : 0;
          const float* __restrict__ dyb = a.dy + ((int64_t)n * a.y_rows + row0) * a.ld + oc;
          // all 32 loads first, the masking after them (a select right behind each load makes it 32 round trips)
#pragma unroll
          for (int g = 0; g < 32; ++g) {
            const int row = wg_hp * 128 + 4 * g + wg_kk;
            dyv[g] = dyb[(int64_t)(row < P_t ? row : 0) * a.ld];
          }
          __builtin_amdgcn_sched_barrier(0);
          if (P_t < FUSED_P || o >= a.Fout) {
#pragma unroll
            for (int g = 0; g < 32; ++g) {
              const int row = wg_hp * 128 + 4 * g + wg_kk;
              dyv[g] = (row < P_t && o < a.Fout) ? dyv[g] : 0.f;
            }
          }
          if (PREC == DSPH_PREC_BF16X3) {
#pragma unroll
            for (int st = 0; st < 4; ++st) {
              // Every second workgroup contracts against -dy and hands in -dW (fused_wgrad_reduce_kernel subtracts its slabs):
              // what the matrix pipe drops when it aligns 32 products to an accumulator hundreds of times their size is dropped
              // towards minus infinity -- at BASELINE configs[2] every element of dW came out low by the same 1.3e-5 of max |dW|
              // whatever its sign (tests/diag_dw_by_order.py).  A bias that ignores the data's sign cancels against the mirror.
              const float sg = (blockIdx.x & 1) ? -1.f : 1.f;
              const float v8[8] = {sg * dyv[8 * st], sg * dyv[8 * st + 1], sg * dyv[8 * st + 2], sg * dyv[8 * st + 3],
                                   sg * dyv[8 * st + 4], sg * dyv[8 * st + 5], sg * dyv[8 * st + 6], sg * dyv[8 * st + 7]};
              split_bf16(v8, dyf.hi[st], dyf.lo[st]);
            }
          }
        }
        __syncthreads();
        { if (PREC == DSPH_PREC_BF16X3) wgrad_plane_bf16(sAcc + (cl * a.K) * WG_TILE_BYTES, smem, la, 0u, dyf); else wgrad_plane(sAcc + (cl * a.K) * WG_TILE_BYTES, smem, la, 0u, dyv); }
        issue_loads(nitem, 1);
        gather_step<WT, RP, false, false, VW>(planeX, planeY, re(D - 1), row_l, val, pre, own);
        __syncthreads();
        { if (PREC == DSPH_PREC_BF16X3) wgrad_plane_bf16(sAcc + (cl * a.K + 1) * WG_TILE_BYTES, smem, la, (unsigned)PLANE_BYTES, dyf); else wgrad_plane(sAcc + (cl * a.K + 1) * WG_TILE_BYTES, smem, la, (unsigned)PLANE_BYTES, dyv); }
        for (int k = 2; k < a.K; k += 2) {
          issue_loads(nitem, k);
          if (cheb) gather_step<WT, RP, true, false, VW>(planeY, planeX, re(D - k), row_l, val, pre, own);
          else gather_step<WT, RP, false, false, VW>(planeY, planeX, re(D - k), row_l, val, pre, own);
          __syncthreads();
          { if (PREC == DSPH_PREC_BF16X3) wgrad_plane_bf16(sAcc + (cl * a.K + k) * WG_TILE_BYTES, smem, la, 0u, dyf); else wgrad_plane(sAcc + (cl * a.K + k) * WG_TILE_BYTES, smem, la, 0u, dyv); }
          if (k + 1 < a.K) {
            issue_loads(nitem, k + 1);
            if (cheb) gather_step<WT, RP, true, false, VW>(planeX, planeY, re(D - k - 1), row_l, val, pre, own);
            else gather_step<WT, RP, false, false, VW>(planeX, planeY, re(D - k - 1), row_l, val, pre, own);
            __syncthreads();
            { if (PREC == DSPH_PREC_BF16X3) wgrad_plane_bf16(sAcc + (cl * a.K + k + 1) * WG_TILE_BYTES, smem, la, (unsigned)PLANE_BYTES, dyf); else wgrad_plane(sAcc + (cl * a.K + k + 1) * WG_TILE_BYTES, smem, la, (unsigned)PLANE_BYTES, dyv); }
          }
        }
      } else {
        // ---- planes mode: the recurrence alone; every new plane's tile rows also go to planes_out ----
        const int64_t spass = (int64_t)G_ROWS * a.Fin;
        float* __restrict__ sp = a.planes_out + ((int64_t)n * a.prow_stride + row0 + row_l) * a.Fin +
                                 (ch_ok ? c * FUSED_CH + 4 * (int)qslot : 0);
        const int srows = ch_ok ? P_t : 0;
        __syncthreads();
        issue_loads(nitem, 1);
        gather_step<WT, RP, false, true, VW>(planeX, planeY, re(D - 1), row_l, val, pre, own, sp, spass, srows);
        for (int k = 2; k < a.K; k += 2) {
          __syncthreads();
          issue_loads(nitem, k);
          sp += a.plane_stride;
          if (cheb) gather_step<WT, RP, true, true, VW>(planeY, planeX, re(D - k), row_l, val, pre, own, sp, spass, srows);
          else gather_step<WT, RP, false, true, VW>(planeY, planeX, re(D - k), row_l, val, pre, own, sp, spass, srows);
          if (k + 1 < a.K) {
            __syncthreads();
            issue_loads(nitem, k + 1);
            sp += a.plane_stride;
            if (cheb) gather_step<WT, RP, true, true, VW>(planeX, planeY, re(D - k - 1), row_l, val, pre, own, sp, spass, srows);
            else gather_step<WT, RP, false, true, VW>(planeX, planeY, re(D - k - 1), row_l, val, pre, own, sp, spass, srows);
          }
        }
      }

      {  // all loads of the next slice are out: fetch the region rows of the slice after it
        const int adv = (item + 2) / items;  // 0: this tile, 1 or 2 (single-slice tiles): tiles ahead
        const int t2 = t + adv * nslots;
        load_rids(t2 < t_end ? t2 : t);
      }
      if (MODE == 0 && c == a.nchunks - 1) {  // this map's accumulators are complete: store them in the next slot 0
        pend = true;
        pend_n = n;
        pend_row0 = row0;
        pend_Pt = P_t;
      }
    }
  }
  if (MODE == 2) {  // D[i = 4 (l >> 4) + r][j = l & 15] of tile (slice cl, order k): channel 16 c + i, column 16 nb + j
    // (one slab per workgroup and pixel half; small maps split the batch over gridDim.y: so many more slabs)
    float* __restrict__ sl = a.slabs + (size_t)((blockIdx.y * gridDim.x + blockIdx.x) * 2 + wg_hp) * ((size_t)a.nchunks * a.K * 16 * 64);
    for (int i = 0; i < a.c_count * a.K; ++i) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(sAcc + i * WG_TILE_BYTES);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        sl[((size_t)(a.c_begin * a.K + i) * 16 + 4 * wg_kk + r) * 64 + 16 * wg_nb + wg_i] = v[r];
    }
  }
  if (MODE == 0 && pend && !(a.dbg & 8)) {  // the last map of this workgroup
    __syncthreads();
    store_pending();
  }
}

// weight-gradient mode on a small map: the batch split over so many workgroup rows (every row needs a map: gy <= N)
static inline int fused_wgrad_gy(int N, int grid, int num_cu) { return std::max(1, std::min(N, num_cu / std::max(grid, 1))); }

template <int PR, int WT, int RPL, int NB, int PREC>
static int launch_variant(const FusedArgs& args, int grid, size_t lds, hipStream_t stream) {
  // (the kernel declares the whole LDS statically; what does not fit it are the weight fragments of many slices: WG)
  const bool wg = lds + FUSED_BIAS_BYTES > (size_t)LDS_BYTES, px = args.pack != 0 || args.pool != 0;
  auto kern = wg ? (px ? cheb_fused_kernel<PR, WT, RPL, NB, PREC, 0, true, true> : cheb_fused_kernel<PR, WT, RPL, NB, PREC, 0, true, false>)
                 : (px ? cheb_fused_kernel<PR, WT, RPL, NB, PREC, 0, false, true> : cheb_fused_kernel<PR, WT, RPL, NB, PREC, 0, false, false>);
  // (forward only: where the tiles do not fill the device the maps of the batch are split over the y dimension)
  const int gy = std::max(1, std::min(args.N, args.num_cu / std::max(grid, 1)));
  hipLaunchKernelGGL(kern, dim3(grid, gy), dim3(FUSED_THREADS), 0, stream, args);
  DSPH_HIP(hipGetLastError());
  return DSPH_OK;
}

template <int PR, int WT, int RPL>
static int dispatch_nb_prec(const FusedArgs& args, int nb, int prec, int grid, size_t lds,
                            hipStream_t stream) {
  if (args.slabs != nullptr) {  // weight-gradient mode
    auto kern = prec == DSPH_PREC_BF16X3 ? cheb_fused_kernel<PR, WT, RPL, 1, DSPH_PREC_BF16X3, 2>
                                         : cheb_fused_kernel<PR, WT, RPL, 1, DSPH_PREC_FP32, 2>;
    hipLaunchKernelGGL(kern, dim3(grid, fused_wgrad_gy(args.N, grid, args.num_cu)), dim3(FUSED_THREADS), 0, stream, args);
    DSPH_HIP(hipGetLastError());
    return DSPH_OK;
  }
  if (args.planes_out != nullptr) {  // planes mode: no contraction, nb and prec do not apply
    auto kern = cheb_fused_kernel<PR, WT, RPL, 1, DSPH_PREC_FP32, 1>;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(FUSED_THREADS), 0, stream, args);
    DSPH_HIP(hipGetLastError());
    return DSPH_OK;
  }
  if (nb == 1) {
    if (prec == DSPH_PREC_BF16X3) return launch_variant<PR, WT, RPL, 1, DSPH_PREC_BF16X3>(args, grid, lds, stream);
    return launch_variant<PR, WT, RPL, 1, DSPH_PREC_FP32>(args, grid, lds, stream);
  }
  if (prec == DSPH_PREC_BF16X3) return launch_variant<PR, WT, RPL, 2, DSPH_PREC_BF16X3>(args, grid, lds, stream);
  return launch_variant<PR, WT, RPL, 2, DSPH_PREC_FP32>(args, grid, lds, stream);
}

// one per (plane rows, ELL width): defined in cheb_fused_inst.hip
#define DSPH_FUSED_DECL(PR, WT) \
  int launch_fused_##PR##_##WT(const FusedArgs& args, int nb, int prec, int grid, size_t lds, hipStream_t stream);
DSPH_FUSED_DECL(576, 9)
DSPH_FUSED_DECL(768, 9)
DSPH_FUSED_DECL(928, 9)
DSPH_FUSED_DECL(1024, 9)
DSPH_FUSED_DECL(1168, 9)
DSPH_FUSED_DECL(576, 12)
DSPH_FUSED_DECL(768, 12)
DSPH_FUSED_DECL(928, 12)
DSPH_FUSED_DECL(1024, 12)
#undef DSPH_FUSED_DECL

}  // namespace dsph
