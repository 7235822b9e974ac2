// One recurrence step through LDS tiles, for graphs too wide for the fused kernels (round 4).
//
// The reference's own models use 20 / 40 / 60 nearest neighbours (healpy_networks.py:38-41; every shipped example: 20): padded ELL
// width 23 and up.  No fused kernel holds a 4-ring halo of such a graph, so they run one step per launch -- and the gather step of
// cheb_step.hip pulls every neighbour row through the L2 again: 23 x |x| of L2 -> CU traffic per step (0.95 ms per step at nside
// 256, 16 channels, batch 8, where the three planes it touches are 1.2 GB).  Here a workgroup owns a tile of 256 consecutive rows
// (a 16 x 16 pixel square in NEST order), stages the rows within ONE hop of it -- the breadth-first region tables the fused path
// builds (cheb_fused.hip, depth 1) -- into LDS, 16 channels at a time, and gathers from there:
//     out[n, m, :] = alpha * sum_j vals[m, j] * in[n, cols[m, j], :] - beta * prev[n, m, :]          (reference utils.py:49-78,
//                                                                                                  gnn_layers.py:138,141)
// with every row of `in` read 1.7 x instead of 23 x (a 21 x 21 region per 16 x 16 tile at 20 neighbours).  A lane owns one (row,
// 4-channel chunk) of the tile for the whole tile: the row's tile-local columns, as swizzled LDS byte addresses, and its values
// stay in registers across all maps and channel slices (WT of each: the template parameter).  The next (map, slice)'s region is
// fetched into registers while the current one is summed; two planes alternate.  The slots of a row are dealt by the residue of
// the neighbour's local index (cheb_fused.hip, get_tiles: slot j of local row i holds a neighbour with index & 3 == (i + j) & 3
// where it can), so that the four rows a 16-lane LDS access covers hit four different bank quarters; the sum therefore runs in
// another ORDER than the gather kernel's -- same terms, results equal to rounding (deterministic from run to run).
// Generic in the graph: rings come from the plan's own pattern; whole graphs only (no halo columns, all rows).
#include <algorithm>

#include "cheb_fused_kernel.h"

namespace dsph {

// TS_RP tile rows per lane: 1 (1,024 threads, sixteen waves: twice the waves to hide the staging loads' latency behind -- the
// kernel waits on memory, not on LDS or the vector pipe -- within 128 registers) or 2 (512 threads, for the widths whose row
// tables do not fit 128 registers)
constexpr int TS_RMAX = 768;      // region rows a plane holds (64 B each): a 27 x 27 pixel region

struct TStepArgs {
  const float* in;
  const float* prev;  // or NULL
  float* out;
  const int32_t* tile_off;
  const int32_t* ring_end;   // [ntiles][FUSED_DMAX + 1]
  const int64_t* ell_off;    // rows
  const int32_t* region;
  const uint16_t* lcols;     // per tile [WT][E] tile-local columns
  const float* lvals;
  int64_t rows;              // rows of every plane (= the plan's rows)
  int ntiles, N, F;
  float alpha, beta;
};

// VEC: the channel count is a multiple of four and the planes are 16-byte aligned (16-byte loads and stores); otherwise a lane
// moves its four channels one by one (the reference's own models have 5 or 2 channels: examples/quick_start.ipynb:118-127)
// YS: the iterations of a tile are split over gridDim.y (small maps); without it the loop starts at zero as it always did (the
// split costs the 24-wide kernel six more spilled registers: not on the large maps' path)
template <int WT, int TS_RP, bool VEC, bool YS = false>
__global__ __launch_bounds__(1024 / TS_RP) void cheb_tstep_kernel(TStepArgs a) {
  constexpr int TS_THREADS = 1024 / TS_RP, TS_ROWS = TS_THREADS / 4;  // rows per pass
  constexpr int TS_SQ = TS_RMAX / TS_ROWS;                            // staged 16-byte pieces per lane
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TS_RMAX * 64];
  const int tid = threadIdx.x, row_l = tid >> 2, slot = tid & 3;
  // A plane row holds 16 channels.  Layers with up to eight channels (the reference's quick-start model: 1 and 5) would leave most
  // of it empty, so several MAPS share a row: with cpm = ceil(F / 4) chunks per map, 4 / cpm maps per iteration (four maps of up
  // to 4 channels, two of up to 8); wider layers: one map, 16 channels of it, per iteration.
  const int cpm = (a.F + 3) / 4, mpi = cpm <= 2 ? 4 / cpm : 1;
  const int nslices = mpi > 1 ? 1 : (a.F + 15) / 16;
  const int ngroups = (a.N + mpi - 1) / mpi;
  const unsigned nblk = gridDim.x;
  // (small maps -- the reference's quick-start model runs at nside 64 .. 8 -- have fewer tiles than the device has CUs: the
  // (map, slice) iterations of a tile are then split over gridDim.y workgroups)
  const int iters_all = ngroups * nslices;
  const int it_begin = YS ? __builtin_amdgcn_readfirstlane((int)((int64_t)iters_all * blockIdx.y / gridDim.y)) : 0;
  const int iters = YS ? __builtin_amdgcn_readfirstlane((int)((int64_t)iters_all * (blockIdx.y + 1) / gridDim.y)) : iters_all;
  if (it_begin >= iters) return;
  for (unsigned tb = xcd_remap(blockIdx.x, nblk); tb < (unsigned)a.ntiles; tb += nblk) {
    const int t = (int)tb;
    const int base = a.tile_off[t];
    const int E = a.ring_end[(size_t)t * (FUSED_DMAX + 1)];       // the tile's own rows (the last tile may be ragged)
    const int R = a.ring_end[(size_t)t * (FUSED_DMAX + 1) + 1];   // rows within one hop
    const int64_t r0 = (int64_t)t * FUSED_P;
    // this lane's rows of the tile-local ELL (row_l, and row_l + 128 with two rows per lane): LDS byte addresses of the neighbours' chunk, and the values
    unsigned pre[TS_RP][WT / 2];  // two 16-bit addresses per register
    float val[TS_RP][WT];
    {
      const uint16_t* lc = a.lcols + a.ell_off[t] * WT;
      const float* lv = a.lvals + a.ell_off[t] * WT;
#pragma unroll
      for (int p = 0; p < TS_RP; ++p) {
        const int row = row_l + TS_ROWS * p;
#pragma unroll
        for (int j = 0; j < WT; ++j) {
          const unsigned c = row < E ? lc[(size_t)j * E + row] : 0u;
          const unsigned addr = plane_byte(c, (unsigned)slot);  // < 768 * 64 = 48 KiB: 16 bits
          if (j & 1) pre[p][j >> 1] |= addr << 16;
          else pre[p][j >> 1] = addr;
          val[p][j] = row < E ? lv[(size_t)j * E + row] : 0.f;
        }
      }
    }
    // staging: region row i = row_l + 128 q, q < TS_SQ, this lane's chunk
    int grow[TS_SQ];
#pragma unroll
    for (int q = 0; q < TS_SQ; ++q) {
      const int i = row_l + TS_ROWS * q;
      grow[q] = i < R ? a.region[base + i] : -1;
    }
    float4 st[TS_SQ];
    auto fetch = [&](int it) __attribute__((always_inline)) {
      const int grp = it / nslices, c = it - grp * nslices;
      const int n = mpi > 1 ? grp * mpi + slot / cpm : grp;
      const int ch = mpi > 1 ? 4 * (slot % cpm) : 16 * c + 4 * slot;
#pragma unroll
      for (int q = 0; q < TS_SQ; ++q) {
        st[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (grow[q] >= 0 && ch < a.F && n < a.N) {
          const float* src = a.in + ((int64_t)n * a.rows + (int64_t)grow[q]) * a.F + ch;
          if (VEC) st[q] = *reinterpret_cast<const float4*>(src);
          else {
            st[q].x = src[0];
            if (ch + 1 < a.F) st[q].y = src[1];
            if (ch + 2 < a.F) st[q].z = src[2];
            if (ch + 3 < a.F) st[q].w = src[3];
          }
        }
      }
    };
    auto stage = [&](unsigned char* plane) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < TS_SQ; ++q) {
        const int i = row_l + TS_ROWS * q;
        if (i < R) *reinterpret_cast<float4*>(plane + plane_byte((unsigned)i, (unsigned)slot)) = st[q];
      }
    };
    fetch(it_begin);
    __syncthreads();  // (the previous tile's last reads of plane 0)
    stage(smem + (unsigned)(it_begin & 1) * (TS_RMAX * 64));
    for (int it = it_begin; it < iters; ++it) {
      if (it + 1 < iters) fetch(it + 1);
      __syncthreads();  // plane `it & 1` is staged; the other one's readers of iteration it - 1 are done
      const int grp = it / nslices, c = it - grp * nslices;
      const int n = mpi > 1 ? grp * mpi + slot / cpm : grp;
      const int ch = mpi > 1 ? 4 * (slot % cpm) : 16 * c + 4 * slot;
      unsigned po = (unsigned)(it & 1) * (TS_RMAX * 64);
      asm volatile("" : "+v"(po));
#pragma unroll
      for (int p = 0; p < TS_RP; ++p) {
        const int row = row_l + TS_ROWS * p;
        if (row < E && ch < a.F && n < a.N) {
          float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
          // eight gathers in flight at a time (all WT at once cost WT x 4 registers of landing space).  The packed addresses are
          // unpacked from a copy hipcc cannot see through: it would otherwise hoist the unpacked, plane-relative addresses of
          // both planes out of the loop over (map, slice) -- 4 WT registers instead of WT / 2
#pragma unroll
          for (int j0 = 0; j0 < WT; j0 += 8) {
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
              unsigned w2 = pre[p][(j0 + j) >> 1];
              asm volatile("" : "+v"(w2));
              v[j] = *reinterpret_cast<const float4*>(smem + (po + (w2 & 0xffffu)));
              v[j + 1] = *reinterpret_cast<const float4*>(smem + (po + (w2 >> 16)));
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              s.x = fmaf(val[p][j0 + j], v[j].x, s.x);
              s.y = fmaf(val[p][j0 + j], v[j].y, s.y);
              s.z = fmaf(val[p][j0 + j], v[j].z, s.z);
              s.w = fmaf(val[p][j0 + j], v[j].w, s.w);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          const int64_t o = ((int64_t)n * a.rows + r0 + row) * a.F + ch;
          float4 r;
          if (a.prev != nullptr) {
            float4 q;
            if (VEC) q = *reinterpret_cast<const float4*>(a.prev + o);
            else {
              q.x = a.prev[o];
              q.y = ch + 1 < a.F ? a.prev[o + 1] : 0.f;
              q.z = ch + 2 < a.F ? a.prev[o + 2] : 0.f;
              q.w = ch + 3 < a.F ? a.prev[o + 3] : 0.f;
            }
            r = make_float4(a.alpha * s.x - a.beta * q.x, a.alpha * s.y - a.beta * q.y, a.alpha * s.z - a.beta * q.z, a.alpha * s.w - a.beta * q.w);
          } else {
            r = make_float4(a.alpha * s.x, a.alpha * s.y, a.alpha * s.z, a.alpha * s.w);
          }
          if (VEC) *reinterpret_cast<float4*>(a.out + o) = r;
          else {
            a.out[o] = r.x;
            if (ch + 1 < a.F) a.out[o + 1] = r.y;
            if (ch + 2 < a.F) a.out[o + 2] = r.z;
            if (ch + 3 < a.F) a.out[o + 3] = r.w;
          }
        }
      }
      if (it + 1 < iters) stage(smem + ((it + 1) & 1) * (TS_RMAX * 64));
    }
  }
}

int tstep_width(int w) {
  if (w <= 16) return 16;
  if (w <= 24) return 24;
  if (w <= 32) return 32;
  return 0;  // (a 48-wide instantiation spills two hundred registers: 40 and 60 neighbours keep the gather kernel of cheb_step.hip)
}

int launch_cheb_tstep(const TStepTables& tb, const float* in, const float* prev, float* out, int64_t rows, int64_t N, int32_t F, float alpha,
                      float beta, int num_cu, hipStream_t stream) {
  TStepArgs a;
  a.in = in;
  a.prev = beta != 0.f ? prev : nullptr;
  a.out = out;
  a.tile_off = tb.tile_off;
  a.ring_end = tb.ring_end;
  a.ell_off = tb.ell_off;
  a.region = tb.region;
  a.lcols = tb.lcols;
  a.lvals = tb.lvals;
  a.rows = rows;
  a.ntiles = tb.ntiles;
  a.N = (int)N;
  a.F = F;
  a.alpha = alpha;
  a.beta = beta;
  // one workgroup per tile up to one per CU; with fewer tiles than CUs the iterations of a tile go to several workgroups
  const int cpm = (F + 3) / 4, mpi = cpm <= 2 ? 4 / cpm : 1;
  const int iters_all = (int)((N + mpi - 1) / mpi) * (mpi > 1 ? 1 : (F + 15) / 16);
  const int gx = std::max(1, std::min(tb.ntiles, num_cu));
  const int gy = std::max(1, std::min(iters_all, num_cu / gx));
  const dim3 grid((unsigned)gx, (unsigned)gy);
  const bool vec = F % 4 == 0 && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(a.prev)) & 15) == 0;
#define DSPH_TS(WT, RP)                                                                                                     \
  do {                                                                                                                      \
    if (gy > 1) {                                                                                                           \
      if (vec) hipLaunchKernelGGL((cheb_tstep_kernel<WT, RP, true, true>), grid, dim3(1024 / RP), 0, stream, a);             \
      else hipLaunchKernelGGL((cheb_tstep_kernel<WT, RP, false, true>), grid, dim3(1024 / RP), 0, stream, a);                \
    } else if (vec) hipLaunchKernelGGL((cheb_tstep_kernel<WT, RP, true, false>), grid, dim3(1024 / RP), 0, stream, a);       \
    else hipLaunchKernelGGL((cheb_tstep_kernel<WT, RP, false, false>), grid, dim3(1024 / RP), 0, stream, a);                 \
  } while (0)
  switch (tb.width) {
    case 16: DSPH_TS(16, 1); break;
    case 24: DSPH_TS(24, 1); break;
    case 32: DSPH_TS(32, 2); break;
    default: set_error("cheb_tstep: no kernel for table width %d", tb.width); return DSPH_E_UNSUPPORTED;
  }
#undef DSPH_TS
  DSPH_HIP(hipGetLastError());
  return DSPH_OK;
}

}  // namespace dsph
