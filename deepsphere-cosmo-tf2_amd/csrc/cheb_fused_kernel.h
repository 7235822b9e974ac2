// Device side of the fused Chebyshev forward (see cheb_fused.hip for the design notes).
// Included by cheb_fused.hip (host: tile tables, dispatch) and by cheb_fused_inst.hip, which is
// compiled once per (plane rows, ELL width) pair so that the instantiations build in parallel.
#pragma once

#include "dsphere_common.h"

namespace dsph {

constexpr int FUSED_P = 256;        // rows per tile
constexpr int FUSED_CH = 16;        // channels per slice
constexpr int FUSED_DMAX = 8;       // deepest halo supported (K <= 9)
constexpr int FUSED_THREADS = 512;  // 8 waves, 2 per SIMD
constexpr int LDS_BYTES = 160 * 1024;
constexpr int G_ROWS = FUSED_THREADS / 4;  // recurrence: four lanes share a region row, 128 rows per pass

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct FusedArgs {
  const float* x;
  const float* bias;
  float* y;
  const unsigned char* wfrag;
  const int32_t* tile_off;
  const int32_t* ring_end;
  const int64_t* ell_off;
  const int32_t* region;
  const uint16_t* lcols;
  const float* lvals;
  int64_t x_rows, y_rows;
  int N, Fin, Fout, K, ntiles, nchunks, act, wfrag_bytes;
  float alpha_rest, beta_rest;  // step k >= 2: T_k = alpha * L~ T_{k-1} - beta * T_{k-2} (2,1 Chebyshev; 1,0 monomial)
  int dbg;  // timing-only ablation bits (DSPH_FUSED_DEBUG): 1 no recurrence, 2 no MFMA, 4 no x loads, 8 no y store
};

// Byte offset of 16-byte slot `slot` (0..3) of region row `row` inside a [rows][16] fp32 plane.
// A row is 64 B, so four consecutive rows fill the 256-byte LDS bank row; XOR-ing the slot with
// bits 2..3 of the row makes 16 lanes that read one slot of 16 consecutive rows (the MFMA operand
// read) hit 16 different 16-byte bank groups.  The recurrence reads with four lanes per row (one
// slot each), which is conflict-free whenever the four rows of a 16-lane group differ mod 4.
__device__ __forceinline__ unsigned plane_byte(unsigned row, unsigned slot) {
  return row * (FUSED_CH * 4) + 16u * (slot ^ ((row >> 2) & 3u));
}

// One plane T_k (tile rows 32*wave .. +32, all 16 channels of the slice) into the MFMA
// accumulators: A from the LDS plane, converted on the fly to split bf16 when PREC says so; B (the
// weight fragments of (order k, slice c), already in operand order) from LDS.
template <int NB, int PREC>
__device__ __forceinline__ void mfma_plane(const unsigned char* __restrict__ plane,
                                           const unsigned char* __restrict__ wblk, int wave, int lane,
                                           f32x16 (&acc)[NB]) {
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
      const bf16x8 bhi = *reinterpret_cast<const bf16x8*>(wblk + b * 2048 + lane * 16);
      const bf16x8 blo = *reinterpret_cast<const bf16x8*>(wblk + b * 2048 + 1024 + lane * 16);
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, bhi, acc[b], 0, 0, 0);  // small terms first
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, blo, acc[b], 0, 0, 0);
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, bhi, acc[b], 0, 0, 0);
    }
  } else {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const float bv = *reinterpret_cast<const float*>(wblk + b * 2048 + t * 256 + lane * 4);
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv, acc[b], 0, 0, 0);
      }
    }
  }
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
template <int WT, int RP, bool CHEB_STEP>
__device__ __forceinline__ void gather_step(const unsigned char* __restrict__ pin,
                                            unsigned char* __restrict__ pout, int nrows, int row_l,
                                            const float (&valc)[RP][(WT + 3) / 4], const unsigned (&pre)[RP][WT],
                                            const unsigned (&own)[RP]) {
#pragma unroll
  for (int p = 0; p < RP; ++p) {
    if (row_l + p * G_ROWS < nrows) {
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int j = 0; j < WT; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(pin + pre[p][j]);
        const float w = quad_bcast(valc[p][j >> 2], j & 3);
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
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the passes apart: 9 gathers in flight, not 9*RP
  }
}

// PR: rows each LDS plane is sized for; RP: recurrence rows per lane (rows with an ELL row <= 128*RP);
// WT: ELL width; NB: 32-column output blocks; PREC: contraction arithmetic.
template <int PR, int WT, int RP, int NB, int PREC>
__global__ __launch_bounds__(FUSED_THREADS, 2) void cheb_fused_kernel(FusedArgs a) {
  constexpr int PLANE_BYTES = PR * FUSED_CH * 4;
  constexpr int NS = (PR * 4 + FUSED_THREADS - 1) / FUSED_THREADS;  // staging float4 per lane
  // all of the CU's LDS, statically: the base is then a compile-time constant that folds into the
  // ds_read/ds_write offset fields (a dynamic LDS symbol costs one v_add per access)
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
  unsigned char* const planeX = smem;
  unsigned char* const planeY = smem + PLANE_BYTES;
  unsigned char* const sW = smem + 2 * PLANE_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid * 16; i < a.wfrag_bytes; i += FUSED_THREADS * 16)
    *reinterpret_cast<uint4*>(sW + i) = *reinterpret_cast<const uint4*>(a.wfrag + i);

  // tiles are dealt to XCDs in contiguous ranges (blocks b and b+8 share an XCD): the 32
  // workgroups of one XCD work on 32 neighbouring tiles at a time and share halos through its L2
  const int G = gridDim.x, xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3;
  const int nslots = (G + 7 - xcd) / 8;
  const int t_begin = (int)((int64_t)a.ntiles * xcd / 8), t_end = (int)((int64_t)a.ntiles * (xcd + 1) / 8);
  const int D = a.K - 1;
  const int row_l = tid >> 2;      // recurrence: this lane's row within a pass
  const unsigned qslot = tid & 3;  //             and its 16-byte slot
  const int items = a.N * a.nchunks;  // (map, slice) pairs per tile
  const size_t wstride = (size_t)a.nchunks * NB * 2048;  // weight blocks: per order
  const bool do_g = !(a.dbg & 1), do_m = !(a.dbg & 2);
  const bool cheb = a.beta_rest != 0.f;  // Chebyshev (2, 1) or monomial (1, 0) steps from k = 2 on

  // ---- software prefetch of the next (tile, map, slice): region row ids and x in registers ----
  int rid[NS];
  float4 pf[NS];
  auto load_rids = [&](int t) {
    const int off = a.tile_off[t];
    const int R = a.ring_end[(size_t)t * (FUSED_DMAX + 1) + D];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int row = (tid + s * FUSED_THREADS) >> 2;
      rid[s] = row < R ? a.region[off + row] : -1;
    }
  };
  auto issue_loads = [&](int item) {
    const int n = item / a.nchunks, c = item - n * a.nchunks;
    const int ch = c * FUSED_CH + 4 * (tid & 3);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (rid[s] >= 0 && ch < a.Fin && !(a.dbg & 4))
        v = *reinterpret_cast<const float4*>(a.x + ((int64_t)n * a.x_rows + rid[s]) * a.Fin + ch);
      pf[s] = v;
    }
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
  auto store_pending = [&]() {
    constexpr int T_LD = 36;  // padded row (floats) of a wave's 32 x 32 block; 8 waves fill plane Y exactly
    float* __restrict__ tw = reinterpret_cast<float*>(planeY) + wave * (32 * T_LD);
    const int li = lane & 31, h = lane >> 5;
    const int cq0 = (lane & 7) * 4, rsub = lane >> 3;
    const bool vec_ok = (a.Fout % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.y) & 15) == 0);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
      for (int q = 0; q < 16; ++q) tw[((q & 3) + 8 * (q >> 2) + 4 * h) * T_LD + li] = acc[b][q];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const int cq = 32 * b + cq0;
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a.bias) {
        bv.x = cq + 0 < a.Fout ? a.bias[cq + 0] : 0.f;
        bv.y = cq + 1 < a.Fout ? a.bias[cq + 1] : 0.f;
        bv.z = cq + 2 < a.Fout ? a.bias[cq + 2] : 0.f;
        bv.w = cq + 3 < a.Fout ? a.bias[cq + 3] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = i * 8 + rsub;
        float4 v = *reinterpret_cast<const float4*>(tw + row * T_LD + cq0);
        v.x = apply_act(v.x + bv.x, a.act);
        v.y = apply_act(v.y + bv.y, a.act);
        v.z = apply_act(v.z + bv.z, a.act);
        v.w = apply_act(v.w + bv.w, a.act);
        const int grow = wave * 32 + row;
        if (grow < pend_Pt) {
          float* __restrict__ yp = a.y + ((int64_t)pend_n * a.y_rows + pend_row0 + grow) * a.Fout + cq;
          if (vec_ok && cq + 3 < a.Fout) {
            *reinterpret_cast<float4*>(yp) = v;
          } else {
            if (cq + 0 < a.Fout) yp[0] = v.x;
            if (cq + 1 < a.Fout) yp[1] = v.y;
            if (cq + 2 < a.Fout) yp[2] = v.z;
            if (cq + 3 < a.Fout) yp[3] = v.w;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    pend = false;
  };

  int t = t_begin + slot0;
  if (t < t_end) {
    load_rids(t);
    issue_loads(0);
  }
  for (; t < t_end; t += nslots) {
    // ring sizes of this tile, 11 bits each, in two scalar registers (re-reading them from memory
    // in every step would put a dependent scalar load in front of each recurrence step)
    const int32_t* __restrict__ re_mem = a.ring_end + (size_t)t * (FUSED_DMAX + 1);
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
    const int64_t lbase = a.ell_off[t] * WT;
    const int64_t row0 = (int64_t)t * FUSED_P;

    // this lane's recurrence rows: ELL values and swizzled LDS byte addresses stay in registers
    float val[RP][(WT + 3) / 4];  // quad-packed: register g of lane q holds the row's value 4g + q
    unsigned pre[RP][WT], own[RP];
#pragma unroll
    for (int p = 0; p < RP; ++p) {
      const int i = row_l + p * G_ROWS;
      own[p] = plane_byte((unsigned)(i < PR ? i : 0), qslot);
#pragma unroll
      for (int g = 0; g < (WT + 3) / 4; ++g) {
        const int j = 4 * g + (int)qslot;
        val[p][g] = (i < E && j < WT) ? a.lvals[lbase + (int64_t)j * E + i] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < WT; ++j) {
        const unsigned c = i < E ? a.lcols[lbase + (int64_t)j * E + i] : 0u;
        pre[p][j] = plane_byte(c, qslot);
      }
    }

    for (int item = 0; item < items; ++item) {
      const int n = item / a.nchunks, c = item - n * a.nchunks;
      __syncthreads();  // the previous slice's last plane is still being read
      // ---- T_0: the prefetched x slice goes to plane X; fetch the next slice meanwhile -------
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const unsigned idx = tid + s * FUSED_THREADS;
        if (idx < (unsigned)PR * 4) *reinterpret_cast<float4*>(planeX + plane_byte(idx >> 2, idx & 3)) = pf[s];
      }
      if (item + 1 < items) {
        issue_loads(item + 1);
      } else if (t + nslots < t_end) {
        load_rids(t + nslots);
        issue_loads(0);
      }
      if (pend && !(a.dbg & 8)) store_pending();
      if (c == 0) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int q = 0; q < 16; ++q) acc[b][q] = 0.f;
      }
      __syncthreads();
      const unsigned char* __restrict__ wblk = sW + (size_t)c * NB * 2048;
      if (do_m) mfma_plane<NB, PREC>(planeX, wblk, wave, lane, acc);

      // ---- recurrence, two steps per trip so that the plane roles are compile-time ----------
      if (do_g) gather_step<WT, RP, false>(planeX, planeY, re(D - 1), row_l, val, pre, own);
      __syncthreads();
      if (do_m) mfma_plane<NB, PREC>(planeY, wblk + wstride, wave, lane, acc);
      for (int k = 2; k < a.K; k += 2) {
        if (do_g) {
          if (cheb) gather_step<WT, RP, true>(planeY, planeX, re(D - k), row_l, val, pre, own);
          else gather_step<WT, RP, false>(planeY, planeX, re(D - k), row_l, val, pre, own);
        }
        __syncthreads();
        if (do_m) mfma_plane<NB, PREC>(planeX, wblk + (size_t)k * wstride, wave, lane, acc);
        if (k + 1 < a.K) {
          if (do_g) {
            if (cheb) gather_step<WT, RP, true>(planeX, planeY, re(D - k - 1), row_l, val, pre, own);
            else gather_step<WT, RP, false>(planeX, planeY, re(D - k - 1), row_l, val, pre, own);
          }
          __syncthreads();
          if (do_m) mfma_plane<NB, PREC>(planeY, wblk + (size_t)(k + 1) * wstride, wave, lane, acc);
        }
      }

      if (c == a.nchunks - 1) {  // this map's accumulators are complete: store them in the next slot 0
        pend = true;
        pend_n = n;
        pend_row0 = row0;
        pend_Pt = P_t;
      }
    }
  }
  if (pend && !(a.dbg & 8)) {  // the last map of this workgroup
    __syncthreads();
    store_pending();
  }
}

template <int PR, int WT, int RPL, int NB, int PREC>
static int launch_variant(const FusedArgs& args, int grid, size_t lds, hipStream_t stream) {
  auto kern = cheb_fused_kernel<PR, WT, RPL, NB, PREC>;
  (void)lds;  // the kernel declares the whole LDS statically
  hipLaunchKernelGGL(kern, dim3(grid), dim3(FUSED_THREADS), 0, stream, args);
  DSPH_HIP(hipGetLastError());
  return DSPH_OK;
}

template <int PR, int WT, int RPL>
static int dispatch_nb_prec(const FusedArgs& args, int nb, int prec, int grid, size_t lds,
                            hipStream_t stream) {
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
DSPH_FUSED_DECL(576, 12)
DSPH_FUSED_DECL(768, 12)
DSPH_FUSED_DECL(928, 12)
DSPH_FUSED_DECL(1024, 12)
#undef DSPH_FUSED_DECL

}  // namespace dsph
