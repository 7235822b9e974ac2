// Single-launch fused Chebyshev forward: the K planes never leave the compute unit.
//
// Replaces the whole of Chebyshev.call (reference gnn_layers.py:131-150): the reference
// materialises K planes of size |x| in memory (tf.stack) and re-lays them out twice; here a
// workgroup owns a tile of P consecutive rows (NEST order: a 16x16 pixel square on a HEALPix
// map), loads the tile plus its (K-1)-hop halo into LDS one 16-channel slice at a time, runs the
// three-term recurrence in LDS on a region that shrinks by one ring per step, and feeds every
// plane T_k straight into MFMA accumulators that stay in registers until y is written once.
//
// The decomposition is generic: rings come from a breadth-first search over the ELL pattern, so
// k-NN graphs, partial-sky graphs, sharded plans with halo rows and arbitrary sparse matrices all
// take the same path; a graph whose halo does not fit in LDS simply reports "not tileable" and
// the caller falls back to the unfused kernels.
//
// Data layout (device):
//   region_rows[tile_off[t] ..]   row ids of tile t's region, ring 0 (the tile's own rows, in
//                                 order) first, then ring 1, ... ring D, each ring ascending
//   ring_end[t][r]                number of region entries within r hops (r = 0..D)
//   lcols/lvals                   tile-local ELL of the rows within D-1 hops, column = index into
//                                 the region list (uint16), stored [slot][row] per tile so that
//                                 lane i reads row i coalesced
// LDS: two planes [Rmax][16] fp32 (16-byte slots XOR-swizzled so that 16 lanes reading the same
// slot of 16 different rows hit 16 different bank groups) + the weight fragments of all
// (order, slice) pairs in MFMA operand order.
// Registers: each lane owns one (or two) region rows for the whole tile: their ELL values and
// pre-swizzled LDS addresses stay in VGPRs across all slices and all maps of the batch.
#include <stdlib.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <vector>

#include "dsphere_common.h"

namespace dsph {

constexpr int FUSED_P = 256;       // rows per tile
constexpr int FUSED_CH = 16;       // channels per slice
constexpr int FUSED_DMAX = 8;      // deepest halo supported (K <= 9)
constexpr int FUSED_THREADS = 512; // 8 waves, 2 per SIMD
constexpr int LDS_BYTES = 160 * 1024;

struct FusedTiles {
  int D = 0;
  int width = 0;     // ELL width of the tile-local table (template width, >= plan width)
  int rpl = 0;       // region rows per lane
  int ntiles = 0;
  int rmax = 0;      // largest region (rows), rounded up to a multiple of 16, >= FUSED_P
  int emax = 0;      // largest number of rows that carry an ELL row
  bool ok = false;
  int32_t* d_tile_off = nullptr;
  int32_t* d_ring_end = nullptr;   // [ntiles][DMAX+1]
  int64_t* d_ell_off = nullptr;    // [ntiles] in rows
  int32_t* d_region = nullptr;
  uint16_t* d_lcols = nullptr;
  float* d_lvals = nullptr;
};

struct FusedPlan {
  std::vector<int32_t> h_cols;  // host copy of the ELL (needed to build tiles for a new K)
  std::vector<float> h_vals;
  std::mutex mu;
  std::map<int, FusedTiles> by_depth;
  int num_cu = 256;
};

static int template_width(int w) {
  if (w <= 9) return 9;
  if (w <= 12) return 12;
  return 0;
}

static void free_tiles(FusedTiles& ft) {
  if (ft.d_tile_off) (void)hipFree(ft.d_tile_off);
  if (ft.d_ring_end) (void)hipFree(ft.d_ring_end);
  if (ft.d_ell_off) (void)hipFree(ft.d_ell_off);
  if (ft.d_region) (void)hipFree(ft.d_region);
  if (ft.d_lcols) (void)hipFree(ft.d_lcols);
  if (ft.d_lvals) (void)hipFree(ft.d_lvals);
  ft = FusedTiles();
}

FusedPlan* fused_plan_build(const dsph_plan* plan, const int32_t* h_cols, const float* h_vals) {
  if (template_width(plan->width) == 0) return nullptr;
  FusedPlan* fp = new FusedPlan();
  const size_t nnz = (size_t)plan->n_rows * plan->width;
  fp->h_cols.assign(h_cols, h_cols + nnz);
  fp->h_vals.assign(h_vals, h_vals + nnz);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, plan->device) == hipSuccess && prop.multiProcessorCount > 0)
    fp->num_cu = prop.multiProcessorCount;
  return fp;
}

void fused_plan_destroy(FusedPlan* fp) {
  if (!fp) return;
  for (auto& kv : fp->by_depth) free_tiles(kv.second);
  delete fp;
}

// Breadth-first rings of every tile; uploads the tables.  Returns a reference to the cached entry.
static const FusedTiles& get_tiles(const dsph_plan* plan, int D) {
  FusedPlan* fp = plan->fused;
  std::lock_guard<std::mutex> lock(fp->mu);
  auto it = fp->by_depth.find(D);
  if (it != fp->by_depth.end()) return it->second;
  FusedTiles& ft = fp->by_depth[D];
  ft.D = D;
  ft.width = template_width(plan->width);
  const int W = plan->width, WT = ft.width;
  const int64_t out_rows = plan->levels.empty() ? plan->n_rows : plan->levels[0];
  const int64_t nt64 = (out_rows + FUSED_P - 1) / FUSED_P;
  if (D < 1 || D > FUSED_DMAX || nt64 > (1 << 30)) return ft;
  const int ntiles = (int)nt64;
  const int32_t* cols = fp->h_cols.data();
  const float* vals = fp->h_vals.data();

  std::vector<int32_t> stamp((size_t)plan->n_cols, -1), local((size_t)plan->n_cols, 0);
  std::vector<int32_t> tile_off((size_t)ntiles + 1, 0), ring_end((size_t)ntiles * (FUSED_DMAX + 1), 0);
  std::vector<int64_t> ell_off((size_t)ntiles, 0);
  std::vector<int32_t> region;
  std::vector<uint16_t> lcols;
  std::vector<float> lvals;
  region.reserve((size_t)ntiles * 600);
  std::vector<int32_t> ring, next;
  int rmax = 0, emax = 0;
  int64_t ell_rows = 0;
  for (int t = 0; t < ntiles; ++t) {
    const int64_t r0 = (int64_t)t * FUSED_P, r1 = std::min<int64_t>(out_rows, r0 + FUSED_P);
    const size_t base = region.size();
    if (base > 0x7fffffffULL - 70000) return ft;  // offsets are int32
    tile_off[t] = (int32_t)base;
    ring.clear();
    for (int64_t r = r0; r < r1; ++r) {
      stamp[r] = t;
      local[r] = (int32_t)(r - r0);
      ring.push_back((int32_t)r);
      region.push_back((int32_t)r);
    }
    int32_t* re = &ring_end[(size_t)t * (FUSED_DMAX + 1)];
    re[0] = (int32_t)ring.size();
    for (int d = 1; d <= D; ++d) {
      next.clear();
      for (int32_t r : ring) {
        if (r >= plan->n_rows) return ft;  // a row that must be computed has no ELL row
        const int32_t* c = cols + (size_t)r * W;
        const float* v = vals + (size_t)r * W;
        for (int j = 0; j < W; ++j) {
          if (v[j] == 0.f) continue;
          const int32_t cj = c[j];
          if (stamp[cj] != t) {
            stamp[cj] = t;
            next.push_back(cj);
          }
        }
      }
      std::sort(next.begin(), next.end());
      for (int32_t r : next) {
        local[r] = (int32_t)(region.size() - base);
        region.push_back(r);
      }
      re[d] = (int32_t)(region.size() - base);
      ring.swap(next);
    }
    for (int d = D + 1; d <= FUSED_DMAX; ++d) re[d] = re[D];
    const int R = re[D], E = re[D - 1];
    if (R > 65535) return ft;  // uint16 local columns
    rmax = std::max(rmax, R);
    emax = std::max(emax, E);
    // tile-local ELL of the rows within D-1 hops, stored [slot][row]
    ell_off[t] = ell_rows;
    const size_t lbase = lcols.size();
    lcols.resize(lbase + (size_t)E * WT, 0);
    lvals.resize(lbase + (size_t)E * WT, 0.f);
    for (int i = 0; i < E; ++i) {
      const int32_t r = region[base + i];
      if (r >= plan->n_rows) return ft;
      const int32_t* c = cols + (size_t)r * W;
      const float* v = vals + (size_t)r * W;
      for (int j = 0; j < WT; ++j) {
        uint16_t lc = (uint16_t)i;
        float lv = 0.f;
        if (j < W && v[j] != 0.f) {
          lc = (uint16_t)local[c[j]];
          lv = v[j];
        }
        lcols[lbase + (size_t)j * E + i] = lc;
        lvals[lbase + (size_t)j * E + i] = lv;
      }
    }
    ell_rows += E;
  }
  tile_off[ntiles] = (int32_t)region.size();
  ft.ntiles = ntiles;
  ft.rmax = std::max((rmax + 15) / 16 * 16, FUSED_P);
  ft.emax = emax;
  ft.rpl = (emax + FUSED_THREADS - 1) / FUSED_THREADS;

  auto up = [](void** dst, const void* src, size_t bytes) -> bool {
    if (bytes == 0) bytes = 16;
    if (hipMalloc(dst, bytes) != hipSuccess) return false;
    return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess;
  };
  if (lcols.empty()) { lcols.push_back(0); lvals.push_back(0.f); }
  bool good = up((void**)&ft.d_tile_off, tile_off.data(), tile_off.size() * 4) &&
              up((void**)&ft.d_ring_end, ring_end.data(), ring_end.size() * 4) &&
              up((void**)&ft.d_ell_off, ell_off.data(), ell_off.size() * 8) &&
              up((void**)&ft.d_region, region.data(), region.size() * 4) &&
              up((void**)&ft.d_lcols, lcols.data(), lcols.size() * 2) &&
              up((void**)&ft.d_lvals, lvals.data(), lvals.size() * 4);
  if (!good) {
    FusedTiles keep = ft;
    free_tiles(ft);
    ft.D = keep.D;
    return ft;
  }
  ft.ok = true;
  return ft;
}

// LDS planes are sized at compile time: 576 rows (a 16x16 tile with a 4-ring halo), 768 or 1024
static int plane_rows_for(int rmax, int emax) {
  if (rmax <= 576 && emax <= 512) return 576;
  if (rmax <= 768 && emax <= 768) return 768;
  return (rmax <= 1024 && emax <= 1024) ? 1024 : 0;
}

static size_t wfrag_bytes(int32_t Fin, int32_t Fout, int32_t K) {
  const int C = (Fin + FUSED_CH - 1) / FUSED_CH, NB = (Fout + 31) / 32;
  return (size_t)K * C * NB * 2048;
}

bool fused_supported(const dsph_plan* plan, int32_t Fin, int32_t Fout, int32_t K) {
  if (!plan->fused) return false;
  if (K < 2 || K - 1 > FUSED_DMAX) return false;
  if (Fin % 4 != 0 || Fin < 8 || Fout > 64 || Fout < 1) return false;
  const FusedTiles& ft = get_tiles(plan, K - 1);
  if (!ft.ok) return false;
  const int pr = plane_rows_for(ft.rmax, ft.emax);
  if (pr == 0) return false;
  const size_t lds = (size_t)2 * pr * FUSED_CH * 4 + wfrag_bytes(Fin, Fout, K);
  return lds <= (size_t)LDS_BYTES;
}

size_t fused_workspace_bytes(const dsph_plan*, int64_t, int32_t Fin, int32_t Fout, int32_t K, int32_t) {
  return wfrag_bytes(Fin, Fout, K);
}

// ------------------------------------------------------------------------------------------
// device code
// ------------------------------------------------------------------------------------------

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
  int dbg;  // timing-only ablation bits (DSPH_FUSED_DEBUG): 1 no recurrence, 2 no MFMA, 4 no x loads, 8 no y store
};

// Weight fragments in MFMA operand order, one 2 KiB block per (order k, slice c, column block nb):
//   bf16x3: lane l, element j  <- w[(c*16 + 8*(l>>5) + j)*K + k][32*nb + (l&31)], hi at +0, lo at +1024
//   fp32  : step t, lane l     <- w[(c*16 + 8*(l>>5) + t)*K + k][32*nb + (l&31)] at t*256 + l*4
__global__ __launch_bounds__(256) void fused_wprep_kernel(const float* __restrict__ w,
                                                          unsigned char* __restrict__ out, int Fin,
                                                          int Fout, int K, int C, int NB, int prec) {
  const int blk = blockIdx.x;  // (k*C + c)*NB + nb
  const int nb = blk % NB, c = (blk / NB) % C, k = blk / (NB * C);
  for (int e = threadIdx.x; e < 512; e += 256) {
    const int l = e >> 3, j = e & 7;
    const int ch = c * FUSED_CH + 8 * (l >> 5) + j, col = 32 * nb + (l & 31);
    const float v = (ch < Fin && col < Fout) ? w[((int64_t)ch * K + k) * Fout + col] : 0.f;
    unsigned char* base = out + (size_t)blk * 2048;
    if (prec == DSPH_PREC_BF16X3) {
      const __bf16 hi = (__bf16)v;
      const __bf16 lo = (__bf16)(v - (float)hi);
      reinterpret_cast<__bf16*>(base)[l * 8 + j] = hi;
      reinterpret_cast<__bf16*>(base + 1024)[l * 8 + j] = lo;
    } else {
      reinterpret_cast<float*>(base)[j * 64 + l] = v;
    }
  }
}

// byte offset of 16-byte slot `slot` (0..3) of region row `row` inside a [rows][16] fp32 plane
__device__ __forceinline__ unsigned plane_byte(unsigned row, unsigned slot) {
  return row * (FUSED_CH * 4) + 16u * (slot ^ ((row >> 2) & 3u));
}

// MFMA operands of one plane T_k for this wave's 32 tile rows: A from the LDS plane (converted
// to split bf16 when PREC says so), B (the weight fragments) from LDS.
template <int NB, int PREC>
struct PlaneFrags {
  float av[8];
  bf16x8 ahi, alo, bhi[NB], blo[NB];
  const unsigned char* wb;
  int lane;

  __device__ __forceinline__ void load(const unsigned char* __restrict__ plane,
                                       const unsigned char* __restrict__ sWblk, int wave, int lane_) {
    lane = lane_;
    wb = sWblk;
    const unsigned r = lane & 31, h = lane >> 5;
    const unsigned row = wave * 32 + r;
    const float4 a0 = *reinterpret_cast<const float4*>(plane + plane_byte(row, 2 * h));
    const float4 a1 = *reinterpret_cast<const float4*>(plane + plane_byte(row, 2 * h + 1));
    av[0] = a0.x; av[1] = a0.y; av[2] = a0.z; av[3] = a0.w;
    av[4] = a1.x; av[5] = a1.y; av[6] = a1.z; av[7] = a1.w;
    if (PREC == DSPH_PREC_BF16X3) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        bhi[b] = *reinterpret_cast<const bf16x8*>(sWblk + b * 2048 + lane * 16);
        blo[b] = *reinterpret_cast<const bf16x8*>(sWblk + b * 2048 + 1024 + lane * 16);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const __bf16 hi = (__bf16)av[j];
        ahi[j] = hi;
        alo[j] = (__bf16)(av[j] - (float)hi);
      }
    }
  }

  // the contraction is cut into PARTS pieces so that each can be issued in front of one pass of
  // the recurrence: the matrix pipe then works in the shadow of that pass's LDS gathers and FMAs
  template <int PART, int PARTS>
  __device__ __forceinline__ void issue(f32x16 (&acc)[NB]) const {
    if (PREC == DSPH_PREC_BF16X3) {
      // 3 products per column block, small terms first; product s goes out in part min(s, PARTS-1)
#pragma unroll
      for (int sidx = 0; sidx < 3; ++sidx) {
        const int part = sidx < PARTS ? sidx : PARTS - 1;
        if (part != PART) continue;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          if (sidx == 0) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, bhi[b], acc[b], 0, 0, 0);
          if (sidx == 1) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, blo[b], acc[b], 0, 0, 0);
          if (sidx == 2) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, bhi[b], acc[b], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int part = (t * PARTS) / 8;
        if (part != PART) continue;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const float bv = *reinterpret_cast<const float*>(wb + b * 2048 + t * 256 + lane * 4);
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv, acc[b], 0, 0, 0);
        }
      }
    }
  }
};

template <int NB, int PREC>
__device__ __forceinline__ void mfma_plane(const unsigned char* __restrict__ plane,
                                           const unsigned char* __restrict__ sWblk, int wave, int lane,
                                           f32x16 (&acc)[NB]) {
  PlaneFrags<NB, PREC> f;
  f.load(plane, sWblk, wave, lane);
  f.template issue<0, 1>(acc);
}

// One sub-pass of a recurrence step: 16-byte slot Q of this lane's RR-th region row.
template <int WT, int RPL, bool HAS_PREV, int RR, int Q>
__device__ __forceinline__ void gather_pass(const unsigned char* __restrict__ pin,
                                            unsigned char* __restrict__ pout, int nrows, int tid,
                                            const float (&val)[RPL][WT], const unsigned (&pre)[RPL][WT],
                                            const unsigned (&own)[RPL]) {
  if (tid + RR * FUSED_THREADS < nrows) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < WT; ++j) {
      const float4 v = *reinterpret_cast<const float4*>(pin + (pre[RR][j] ^ (unsigned)(Q << 4)));
      s.x = fmaf(val[RR][j], v.x, s.x);
      s.y = fmaf(val[RR][j], v.y, s.y);
      s.z = fmaf(val[RR][j], v.z, s.z);
      s.w = fmaf(val[RR][j], v.w, s.w);
    }
    float4* op = reinterpret_cast<float4*>(pout + (own[RR] ^ (unsigned)(Q << 4)));
    if (HAS_PREV) {
      const float4 q = *op;
      s.x = 2.f * s.x - q.x;
      s.y = 2.f * s.y - q.y;
      s.z = 2.f * s.z - q.z;
      s.w = 2.f * s.w - q.w;
    }
    *op = s;
  }
}

template <int WT, int RPL, bool HAS_PREV, int NB, int PREC, int P>
struct StepLoop {
  static __device__ __forceinline__ void run(const PlaneFrags<NB, PREC>& f, f32x16 (&acc)[NB], bool do_m,
                                             bool do_g, const unsigned char* __restrict__ pin,
                                             unsigned char* __restrict__ pout, int nrows, int tid,
                                             const float (&val)[RPL][WT], const unsigned (&pre)[RPL][WT],
                                             const unsigned (&own)[RPL]) {
    constexpr int PARTS = PREC == DSPH_PREC_BF16X3 ? 3 : 4;
    if (P < PARTS && do_m) f.template issue<(P < PARTS ? P : 0), PARTS>(acc);
#ifdef DSPH_PIN_SCHED
    __builtin_amdgcn_sched_barrier(0);
#endif
    if (do_g) gather_pass<WT, RPL, HAS_PREV, P / 4, P % 4>(pin, pout, nrows, tid, val, pre, own);
#ifdef DSPH_PIN_SCHED
    __builtin_amdgcn_sched_barrier(0);
#endif
    StepLoop<WT, RPL, HAS_PREV, NB, PREC, P + 1>::run(f, acc, do_m, do_g, pin, pout, nrows, tid, val, pre, own);
  }
};
template <int WT, int RPL, bool HAS_PREV, int NB, int PREC>
struct StepLoop<WT, RPL, HAS_PREV, NB, PREC, 4 * RPL> {
  static __device__ __forceinline__ void run(const PlaneFrags<NB, PREC>&, f32x16 (&)[NB], bool, bool,
                                             const unsigned char* __restrict__, unsigned char* __restrict__, int,
                                             int, const float (&)[RPL][WT], const unsigned (&)[RPL][WT],
                                             const unsigned (&)[RPL]) {}
};

// Contract plane `pin` (= T_{k-1}) into the accumulators while computing T_k = alpha L~ T_{k-1} - T_{k-2}
// from it into `pout`: the MFMAs of the former are issued between the gather passes of the latter.
template <int WT, int RPL, bool HAS_PREV, int NB, int PREC>
__device__ __forceinline__ void fused_step(const unsigned char* __restrict__ pin, unsigned char* __restrict__ pout,
                                           const unsigned char* __restrict__ wblk, int wave, int lane,
                                           f32x16 (&acc)[NB], bool do_m, bool do_g, int nrows, int tid,
                                           const float (&val)[RPL][WT], const unsigned (&pre)[RPL][WT],
                                           const unsigned (&own)[RPL]) {
  PlaneFrags<NB, PREC> f;
  f.load(pin, wblk, wave, lane);
#ifdef DSPH_PIN_SCHED
  __builtin_amdgcn_sched_barrier(0);
#endif
  StepLoop<WT, RPL, HAS_PREV, NB, PREC, 0>::run(f, acc, do_m, do_g, pin, pout, nrows, tid, val, pre, own);
}

// PR: rows each LDS plane is sized for; RPL: recurrence rows per lane (PR <= 512*RPL + ring D);
// WT: ELL width; NB: 32-column output blocks; PREC: contraction arithmetic.
template <int PR, int WT, int RPL, int NB, int PREC>
__global__ __launch_bounds__(FUSED_THREADS, 2) void cheb_fused_kernel(FusedArgs a) {
  constexpr int PLANE_BYTES = PR * FUSED_CH * 4;
  constexpr int NS = (PR * 4 + FUSED_THREADS - 1) / FUSED_THREADS;  // staging float4 per lane
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
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
  const int items = a.N * a.nchunks;    // (map, slice) pairs per tile

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

  int t = t_begin + slot0;
  if (t < t_end) {
    load_rids(t);
    issue_loads(0);
  }
  for (; t < t_end; t += nslots) {
    const int32_t* __restrict__ re = a.ring_end + (size_t)t * (FUSED_DMAX + 1);
    const int P_t = re[0], E = re[D - 1];
    const int64_t lbase = a.ell_off[t] * WT;
    const int64_t row0 = (int64_t)t * FUSED_P;

    // this lane's recurrence rows: ELL values and swizzled LDS byte addresses stay in registers
    float val[RPL][WT];
    unsigned pre[RPL][WT], own[RPL];
#pragma unroll
    for (int p = 0; p < RPL; ++p) {
      const int i = tid + p * FUSED_THREADS;
      own[p] = plane_byte((unsigned)i, 0);
#pragma unroll
      for (int j = 0; j < WT; ++j) {
        float v = 0.f;
        unsigned c = 0;
        if (i < E) {
          v = a.lvals[lbase + (int64_t)j * E + i];
          c = a.lcols[lbase + (int64_t)j * E + i];
        }
        val[p][j] = v;
        pre[p][j] = plane_byte(c, 0);
      }
    }

    f32x16 acc[NB];
    for (int item = 0; item < items; ++item) {
      const int n = item / a.nchunks, c = item - n * a.nchunks;
      if (c == 0) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int q = 0; q < 16; ++q) acc[b][q] = 0.f;
      }
      __syncthreads();  // the previous slice's last plane is still being read by MFMA
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
      __syncthreads();
      const unsigned char* __restrict__ wblk = sW + (size_t)c * NB * 2048;
      const size_t wstride = (size_t)a.nchunks * NB * 2048;  // per order
      const bool do_g = !(a.dbg & 1), do_m = !(a.dbg & 2);
      // ---- recurrence + contraction, two steps per trip so that the plane roles are static ----
      fused_step<WT, RPL, false, NB, PREC>(planeX, planeY, wblk, wave, lane, acc, do_m, do_g, re[D - 1], tid, val,
                                          pre, own);
      __syncthreads();
      int k = 2;
      for (; k + 1 < a.K; k += 2) {
        fused_step<WT, RPL, true, NB, PREC>(planeY, planeX, wblk + (size_t)(k - 1) * wstride, wave, lane, acc, do_m,
                                           do_g, re[D - k], tid, val, pre, own);
        __syncthreads();
        fused_step<WT, RPL, true, NB, PREC>(planeX, planeY, wblk + (size_t)k * wstride, wave, lane, acc, do_m, do_g,
                                           re[D - k - 1], tid, val, pre, own);
        __syncthreads();
      }
      if (k < a.K) {  // K odd: one more step, the last plane ends up in X
        fused_step<WT, RPL, true, NB, PREC>(planeY, planeX, wblk + (size_t)(k - 1) * wstride, wave, lane, acc, do_m,
                                           do_g, re[D - k], tid, val, pre, own);
        __syncthreads();
        if (do_m) mfma_plane<NB, PREC>(planeX, wblk + (size_t)k * wstride, wave, lane, acc);
      } else {
        if (do_m) mfma_plane<NB, PREC>(planeY, wblk + (size_t)(k - 1) * wstride, wave, lane, acc);
      }

      // ---- epilogue after the last slice of a map: bias, activation, one store of y ---------
      // The accumulator tile (column per lane, rows in registers) goes through LDS so that every
      // store instruction writes whole 256-byte pixel rows (16 bytes per lane) instead of 128-byte
      // fragments: a row-per-lane dword epilogue is store-issue bound (it cost 8 of 28 ms).
      if (c == a.nchunks - 1 && !(a.dbg & 8)) {
        constexpr int T_LD = 32 * NB + 4;  // padded row (floats) of a wave's 32 x (32*NB) tile
        __syncthreads();                   // every wave is done reading the planes
        float* __restrict__ tw = reinterpret_cast<float*>(smem) + wave * (32 * T_LD);
        const int li = lane & 31, h = lane >> 5;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int q = 0; q < 16; ++q)
            tw[((q & 3) + 8 * (q >> 2) + 4 * h) * T_LD + 32 * b + li] = acc[b][q];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        constexpr int LPR = 8 * NB;       // lanes per output row (float4 each)
        constexpr int RPI = 64 / LPR;     // rows per store instruction
        const int cq = (lane % LPR) * 4, rsub = lane / LPR;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.bias) {
          bv.x = cq + 0 < a.Fout ? a.bias[cq + 0] : 0.f;
          bv.y = cq + 1 < a.Fout ? a.bias[cq + 1] : 0.f;
          bv.z = cq + 2 < a.Fout ? a.bias[cq + 2] : 0.f;
          bv.w = cq + 3 < a.Fout ? a.bias[cq + 3] : 0.f;
        }
        const bool vec_ok = (a.Fout % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.y) & 15) == 0);
#pragma unroll
        for (int i = 0; i < 32 / RPI; ++i) {
          const int row = i * RPI + rsub;
          float4 v = *reinterpret_cast<const float4*>(tw + row * T_LD + cq);
          v.x = apply_act(v.x + bv.x, a.act);
          v.y = apply_act(v.y + bv.y, a.act);
          v.z = apply_act(v.z + bv.z, a.act);
          v.w = apply_act(v.w + bv.w, a.act);
          const int grow = wave * 32 + row;
          if (grow < P_t) {
            float* __restrict__ yp = a.y + ((int64_t)n * a.y_rows + row0 + grow) * a.Fout + cq;
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
      }
    }
  }
}

template <int PR, int WT, int RPL, int NB, int PREC>
static int launch_variant(const FusedArgs& args, int grid, size_t lds, hipStream_t stream) {
  auto kern = cheb_fused_kernel<PR, WT, RPL, NB, PREC>;
  DSPH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(FUSED_THREADS), lds, stream, args);
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

int launch_cheb_fused(const dsph_plan* plan, const float* x, const float* w, const float* bias,
                      float* y, int64_t N, int32_t Fin, int32_t Fout, int32_t K, int32_t act,
                      int32_t precision, void* workspace, size_t workspace_bytes,
                      hipStream_t stream) {
  if (!fused_supported(plan, Fin, Fout, K)) {
    set_error("cheb_fused: plan/shape not supported");
    return DSPH_E_UNSUPPORTED;
  }
  const FusedTiles& ft = get_tiles(plan, K - 1);
  const size_t wb = wfrag_bytes(Fin, Fout, K);
  if (!workspace || workspace_bytes < wb) {
    set_error("cheb_fused: workspace %zu < %zu", workspace_bytes, wb);
    return DSPH_E_WORKSPACE;
  }
  if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(workspace) & 15)) {
    set_error("cheb_fused: x and workspace must be 16-byte aligned");
    return DSPH_E_BADARG;
  }
  const int C = (Fin + FUSED_CH - 1) / FUSED_CH, NB = (Fout + 31) / 32;
  hipLaunchKernelGGL(fused_wprep_kernel, dim3(K * C * NB), dim3(256), 0, stream, w,
                     static_cast<unsigned char*>(workspace), (int)Fin, (int)Fout, (int)K, C, NB,
                     (int)precision);
  DSPH_HIP(hipGetLastError());

  FusedArgs args;
  args.x = x;
  args.bias = bias;
  args.y = y;
  args.wfrag = static_cast<const unsigned char*>(workspace);
  args.tile_off = ft.d_tile_off;
  args.ring_end = ft.d_ring_end;
  args.ell_off = ft.d_ell_off;
  args.region = ft.d_region;
  args.lcols = ft.d_lcols;
  args.lvals = ft.d_lvals;
  args.x_rows = plan->n_cols;
  args.y_rows = plan->levels.empty() ? plan->n_rows : plan->levels[0];
  args.N = (int)N;
  args.Fin = Fin;
  args.Fout = Fout;
  args.K = K;
  args.ntiles = ft.ntiles;
  args.nchunks = C;
  args.act = act;
  args.wfrag_bytes = (int)wb;
  const char* dbg = getenv("DSPH_FUSED_DEBUG");
  args.dbg = dbg ? atoi(dbg) : 0;
  const int pr = plane_rows_for(ft.rmax, ft.emax);
  const size_t lds = (size_t)2 * pr * FUSED_CH * 4 + wb;
  const int grid = std::max(8, std::min(plan->fused->num_cu, (ft.ntiles + 7) / 8 * 8));
  if (ft.width == 9) {
    if (pr == 576) return dispatch_nb_prec<576, 9, 1>(args, NB, precision, grid, lds, stream);
    if (pr == 768) return dispatch_nb_prec<768, 9, 2>(args, NB, precision, grid, lds, stream);
    return dispatch_nb_prec<1024, 9, 2>(args, NB, precision, grid, lds, stream);
  }
  if (pr == 576) return dispatch_nb_prec<576, 12, 1>(args, NB, precision, grid, lds, stream);
  if (pr == 768) return dispatch_nb_prec<768, 12, 2>(args, NB, precision, grid, lds, stream);
  return dispatch_nb_prec<1024, 12, 2>(args, NB, precision, grid, lds, stream);
}

}  // namespace dsph
