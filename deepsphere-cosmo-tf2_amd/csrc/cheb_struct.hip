// Host side of the structured-tile kernel (cheb_struct_kernel.h): verification of the 2-D stencil structure of
// L~ per row and per tile, the direction-ordered copy of its values, weight preparation and the launch.
//
// Nothing here assumes HEALPix: a row is "regular" when, reading row and column indices as Morton codes
// (x = even bits, y = odd bits), every non-zero of the row lies within +-1 of the row in both coordinates and no
// direction occurs twice; a 256-row tile is "class R" for depth D when its own rows and the cells of its rings
// 1..D-1 exist and are regular and the cells of ring D exist as columns.  HEALPix NEST maps (and any other map
// stored in Z-order) satisfy this away from base-pixel borders; everything else goes to the BFS-tile kernel.
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "cheb_struct_kernel.h"

namespace dsph {

// One row of L~: gdiag[r], gvals8[r][8] and flag[r] (1 = regular).  Rows r >= n_rows (columns that are not rows of this
// plan: halo rows of a shard) get zeros and flag 0.  (__host__ too: the sanitizer build of the plan builders, `make asan`,
// runs these two set-up passes on the host -- tools/asan/)
__host__ __device__ inline void struct_row_body(int64_t r, const int32_t* __restrict__ cols, const float* __restrict__ vals, int W,
                                                int64_t n_rows, float* __restrict__ gvals8, float* __restrict__ gdiag,
                                                unsigned char* __restrict__ flag) {
  // canonical direction index (0 = the row itself, 1..8 = kDirX / kDirY order) by (dy + 1) * 3 + (dx + 1)
  constexpr int kDirIndex[9] = {8, 7, 6, 1, 0, 5, 2, 3, 4};
  float out[9];
#pragma unroll
  for (int d = 0; d < 9; ++d) out[d] = 0.f;
  bool ok = r < n_rows;
  if (ok) {
    const int X = (int)st_compress((unsigned)r), Y = (int)st_compress((unsigned)r >> 1);
    unsigned used = 0;
    for (int j = 0; j < W; ++j) {
      const float v = vals[r * W + j];
      if (v == 0.f) continue;
      const unsigned c = (unsigned)cols[r * W + j];
      const int dx = (int)st_compress(c) - X, dy = (int)st_compress(c >> 1) - Y;
      if (dx < -1 || dx > 1 || dy < -1 || dy > 1) { ok = false; continue; }
      const int d = kDirIndex[(dy + 1) * 3 + (dx + 1)];
      if (used & (1u << d)) ok = false;
      used |= 1u << d;
#pragma unroll
      for (int e = 0; e < 9; ++e)
        if (e == d) out[e] = v;
    }
  }
  gdiag[r] = out[0];
#pragma unroll
  for (int d = 0; d < 8; ++d) gvals8[r * 8 + d] = out[d + 1];
  flag[r] = ok ? 1 : 0;
}
__global__ __launch_bounds__(256) void struct_rows_kernel(const int32_t* __restrict__ cols, const float* __restrict__ vals,
                                                          int W, int64_t n_rows, int64_t n_cols,
                                                          float* __restrict__ gvals8, float* __restrict__ gdiag,
                                                          unsigned char* __restrict__ flag) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n_cols) return;
  struct_row_body(r, cols, vals, W, n_rows, gvals8, gdiag, flag);
}

// One 256-row tile: cls[t] bit 0 = class R for depth D, bit 1 = interior (every region cell is an
// output row of the plan, i.e. no halo row of another rank is read).
__host__ __device__ inline unsigned char struct_tile_body(int t, const unsigned char* __restrict__ flag, int D, int64_t n_rows,
                                                          int64_t n_cols, int64_t out_rows, int dlimit) {
  const int64_t row0 = (int64_t)t * 256;
  bool ok = D >= 1 && D <= dlimit && row0 + 256 <= out_rows && row0 + 256 <= 0xffffffffLL;
  bool interior = true;
  if (ok) {
    const int X0 = (int)st_compress((unsigned)row0), Y0 = (int)st_compress((unsigned)row0 >> 1);
    if (X0 - D < 0 || Y0 - D < 0 || X0 + ST_TILE + D > 65535 || Y0 + ST_TILE + D > 65535) ok = false;
    for (int gy = -D; ok && gy < ST_TILE + D; ++gy)
      for (int gx = -D; gx < ST_TILE + D; ++gx) {
        const int a = -gx > gx - (ST_TILE - 1) ? -gx : gx - (ST_TILE - 1), b = -gy > gy - (ST_TILE - 1) ? -gy : gy - (ST_TILE - 1);
        const int ring = (a > b ? a : b) > 0 ? (a > b ? a : b) : 0;
        const int64_t rid = (int64_t)st_morton((unsigned)(X0 + gx), (unsigned)(Y0 + gy));
        if (rid >= n_cols) { ok = false; break; }
        if (ring < D && (rid >= n_rows || !flag[rid])) { ok = false; break; }
        if (rid >= out_rows) interior = false;
      }
  }
  return (unsigned char)((ok ? 1 : 0) | (interior ? 2 : 0));
}
__global__ __launch_bounds__(256) void struct_tiles_kernel(const unsigned char* __restrict__ flag, int ntiles, int D,
                                                           int64_t n_rows, int64_t n_cols, int64_t out_rows,
                                                           unsigned char* __restrict__ cls, int dlimit) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= ntiles) return;
  cls[t] = struct_tile_body(t, flag, D, n_rows, n_cols, out_rows, dlimit);
}

// Weight fragments of the structured kernel, one 2 KiB block per (slice c, order k, column block nb), a slice
// contiguous (it is streamed into LDS as one piece):
//   bf16x3: lane l, element j <- w[(c*16 + 8*(l>>5) + j)*K + k][32*nb + (l&31)], hi at +0, lo at +1024
//   fp32  : lane l, half e, i <- w[(c*16 + 8*(l>>5) + 4e + i)*K + k][32*nb + (l&31)] at e*1024 + l*16 + 4i
// pack (StructArgs::pack): one slice, two column blocks, block diagonal -- inner index 4 q + c against columns 16 q + o.
__global__ __launch_bounds__(256) void struct_wprep_kernel(const float* __restrict__ w, unsigned char* __restrict__ out,
                                                           int Fin, int Fout, int K, int C, int NB, int prec, int ld, int pack) {
  const int blk = blockIdx.x;  // (c*K + k)*NB + nb
  const int nb = blk % NB, k = (blk / NB) % K, c = blk / (NB * K);
  unsigned char* base = out + (size_t)blk * (prec == DSPH_PREC_BF16X6 ? ST_WBLK3 : 2048);
  for (int e = threadIdx.x; e < 512; e += 256) {
    const int l = e >> 3, j = e & 7;
    const int ch = c * 16 + 8 * (l >> 5) + j, col = 32 * nb + (l & 31);
    float v;
    if (pack) {  // pack = P maps: 16 / P inner indices and 64 / P columns each
      const int ci = 16 / pack, cm = 64 / pack;
      v = (ch / ci == col / cm && ch % ci < Fin && col % cm < Fout) ? w[((int64_t)(ch % ci) * K + k) * ld + col % cm] : 0.f;
    }
    else v = (ch < Fin && col < Fout) ? w[((int64_t)ch * K + k) * ld + col] : 0.f;
    if (prec == DSPH_PREC_BF16X6) {  // exact three-way split by truncation (st_contract): hi | mid | lo, 1 KiB each
      const unsigned u = __float_as_uint(v);
      const float r = v - __uint_as_float(u & 0xffff0000u);
      const unsigned s = __float_as_uint(r);
      const float q = r - __uint_as_float(s & 0xffff0000u);
      reinterpret_cast<unsigned short*>(base)[l * 8 + j] = (unsigned short)(u >> 16);
      reinterpret_cast<unsigned short*>(base + 1024)[l * 8 + j] = (unsigned short)(s >> 16);
      reinterpret_cast<unsigned short*>(base + 2048)[l * 8 + j] = (unsigned short)(__float_as_uint(q) >> 16);
    } else if (prec == DSPH_PREC_BF16X3) {
      const __bf16 hi = (__bf16)v;
      const __bf16 lo = (__bf16)(v - (float)hi);
      reinterpret_cast<__bf16*>(base)[l * 8 + j] = hi;
      reinterpret_cast<__bf16*>(base + 1024)[l * 8 + j] = lo;
    } else {
      reinterpret_cast<float*>(base + (j >> 2) * 1024)[l * 4 + (j & 3)] = v;
    }
  }
}

int struct_build_rows(const dsph_plan* plan, float** gvals8, float** gdiag, unsigned char** flag) {
  const int64_t n = plan->n_cols;
  DSPH_HIP(hipMalloc((void**)gvals8, (size_t)n * 8 * sizeof(float)));
  DSPH_HIP(hipMalloc((void**)gdiag, (size_t)n * sizeof(float)));
  DSPH_HIP(hipMalloc((void**)flag, (size_t)n));
#ifdef DSPH_HOST_EMU  // (`make asan`: "device" memory is host memory there, tools/asan/hip_stub.cpp)
  for (int64_t r = 0; r < n; ++r) struct_row_body(r, plan->d_cols, plan->d_vals, (int)plan->width, plan->n_rows, *gvals8, *gdiag, *flag);
#else
  hipLaunchKernelGGL(struct_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, plan->d_cols, plan->d_vals,
                     (int)plan->width, plan->n_rows, plan->n_cols, *gvals8, *gdiag, *flag);
#endif
  DSPH_HIP(hipGetLastError());
  DSPH_HIP(hipDeviceSynchronize());
  return DSPH_OK;
}

__host__ __device__ inline void struct_patch_body(int64_t i, const int32_t* __restrict__ rows, const float* __restrict__ vals,
                                                  float* __restrict__ gvals8, float* __restrict__ gdiag) {
  const int64_t r = rows[i];
  gdiag[r] = vals[9 * i];
#pragma unroll
  for (int d = 0; d < 8; ++d) gvals8[r * 8 + d] = vals[9 * i + 1 + d];
}
__global__ __launch_bounds__(256) void struct_patch_kernel(const int32_t* __restrict__ rows, const float* __restrict__ vals, int64_t n,
                                                           float* __restrict__ gvals8, float* __restrict__ gdiag) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) struct_patch_body(i, rows, vals, gvals8, gdiag);
}
int struct_patch_rows(const dsph_plan* plan, float* gvals8, float* gdiag, const int32_t* rows, const float* vals, int64_t n) {
  (void)plan;
  if (n <= 0) return DSPH_OK;
  int32_t* d_rows = nullptr;
  float* d_vals = nullptr;
  hipError_t e = hipMalloc((void**)&d_rows, (size_t)n * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&d_vals, (size_t)n * 36);
  if (e == hipSuccess) e = hipMemcpy(d_rows, rows, (size_t)n * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_vals, vals, (size_t)n * 36, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
#ifdef DSPH_HOST_EMU
    for (int64_t i = 0; i < n; ++i) struct_patch_body(i, d_rows, d_vals, gvals8, gdiag);
#else
    hipLaunchKernelGGL(struct_patch_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_rows, d_vals, n, gvals8, gdiag);
#endif
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (d_rows) (void)hipFree(d_rows);
  if (d_vals) (void)hipFree(d_vals);
  if (e != hipSuccess) return hip_fail(e, "struct_patch_rows");
  return DSPH_OK;
}

// cls (host, ntiles bytes) <- classification of every 256-row tile for depth D
int struct_classify_tiles(const dsph_plan* plan, const unsigned char* d_flag, int ntiles, int D, int64_t out_rows,
                          unsigned char* h_cls, int dlimit) {
  unsigned char* d_cls = nullptr;
  DSPH_HIP(hipMalloc((void**)&d_cls, (size_t)ntiles));
#ifdef DSPH_HOST_EMU
  for (int t = 0; t < ntiles; ++t) d_cls[t] = struct_tile_body(t, d_flag, D, plan->n_rows, plan->n_cols, out_rows, dlimit);
#else
  hipLaunchKernelGGL(struct_tiles_kernel, dim3((ntiles + 255) / 256), dim3(256), 0, 0, d_flag, ntiles, D, plan->n_rows,
                     plan->n_cols, out_rows, d_cls, dlimit);
#endif
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpy(h_cls, d_cls, (size_t)ntiles, hipMemcpyDeviceToHost);
  (void)hipFree(d_cls);
  if (e != hipSuccess) return hip_fail(e, "struct_classify_tiles");
  return DSPH_OK;
}

// (the kernel addresses x by 32-bit byte offsets inside a map: struct_map_ok())
// The activations the structured-tile kernel does not apply itself (it fuses bias and ReLU only): one elementwise
// pass over a (rows x cols) block of y with row stride ld.
__global__ __launch_bounds__(256) void struct_act_kernel(float* __restrict__ y, int64_t rows, int cols, int ld, int act) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= rows * cols) return;
  const int64_t r = e / cols;
  const int c = (int)(e - r * cols);
  y[r * ld + c] = apply_act(y[r * ld + c], act);
}

int launch_struct_act(float* y, int64_t rows, int32_t cols, int32_t ld, int32_t act, hipStream_t stream) {
  const int64_t total = rows * cols;
  if (total <= 0) return DSPH_OK;
  hipLaunchKernelGGL(struct_act_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, y, rows, (int)cols, (int)ld, (int)act);
  DSPH_HIP(hipGetLastError());
  return DSPH_OK;
}

// The same pass over the rows of a list of 256-row tiles, for every map: what one part of a two-part launch finishes (each part
// finalises only the rows it wrote, so repeating a part or running it alone never touches the other part's rows).
__global__ __launch_bounds__(256) void struct_act_tiles_kernel(float* __restrict__ y, const int32_t* __restrict__ tiles, int64_t y_rows,
                                                               int cols, int ld, int act) {
  const int64_t r0 = (int64_t)tiles[blockIdx.x] * 256;
  const int64_t nr = y_rows - r0 < 256 ? y_rows - r0 : 256;
  float* base = y + ((int64_t)blockIdx.y * y_rows + r0) * ld;
  for (int64_t e = threadIdx.x; e < nr * cols; e += 256) {
    const int64_t r = e / cols;
    const int c = (int)(e - r * cols);
    base[r * ld + c] = apply_act(base[r * ld + c], act);
  }
}

int launch_struct_act_tiles(float* y, const int32_t* d_tiles, int ntiles, int64_t N, int64_t y_rows, int32_t cols, int32_t ld,
                            int32_t act, hipStream_t stream) {
  if (ntiles <= 0 || N <= 0) return DSPH_OK;
  for (int64_t n0 = 0; n0 < N; n0 += 65535) {
    const unsigned nn = (unsigned)(N - n0 < 65535 ? N - n0 : 65535);
    hipLaunchKernelGGL(struct_act_tiles_kernel, dim3((unsigned)ntiles, nn), dim3(256), 0, stream, y + n0 * y_rows * ld, d_tiles, y_rows,
                       (int)cols, (int)ld, (int)act);
    DSPH_HIP(hipGetLastError());
  }
  return DSPH_OK;
}

// Skip connection of a residual block in one pass (dsph_residual_epilogue): four floats per thread where the pointers allow.
template <bool BEFORE>
__global__ __launch_bounds__(256) void residual_epilogue_kernel(float* __restrict__ y, const float* __restrict__ skip, int64_t n,
                                                                float alpha, int act, int vec) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  if (vec) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
      float4 a = reinterpret_cast<float4*>(y)[i];
      const float4 b = reinterpret_cast<const float4*>(skip)[i];
      if (BEFORE) {
        a.x = apply_act(a.x, act) + alpha * b.x; a.y = apply_act(a.y, act) + alpha * b.y;
        a.z = apply_act(a.z, act) + alpha * b.z; a.w = apply_act(a.w, act) + alpha * b.w;
      } else {
        a.x = apply_act(a.x + alpha * b.x, act); a.y = apply_act(a.y + alpha * b.y, act);
        a.z = apply_act(a.z + alpha * b.z, act); a.w = apply_act(a.w + alpha * b.w, act);
      }
      reinterpret_cast<float4*>(y)[i] = a;
    }
    for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
      y[i] = BEFORE ? apply_act(y[i], act) + alpha * skip[i] : apply_act(y[i] + alpha * skip[i], act);
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
      y[i] = BEFORE ? apply_act(y[i], act) + alpha * skip[i] : apply_act(y[i] + alpha * skip[i], act);
  }
}

int launch_residual_epilogue(float* y, const float* skip, int64_t n, float alpha, int32_t act, bool before, hipStream_t stream) {
  if (n <= 0) return DSPH_OK;
  const int vec = ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(skip)) & 15) == 0;
  const int64_t work = vec ? (n + 3) / 4 : n;
  const unsigned grid = (unsigned)std::min<int64_t>((work + 255) / 256, 256 * 32);
  if (before) hipLaunchKernelGGL(residual_epilogue_kernel<true>, dim3(grid), dim3(256), 0, stream, y, skip, n, alpha, (int)act, vec);
  else hipLaunchKernelGGL(residual_epilogue_kernel<false>, dim3(grid), dim3(256), 0, stream, y, skip, n, alpha, (int)act, vec);
  DSPH_HIP(hipGetLastError());
  return DSPH_OK;
}

bool struct_shape_ok(int32_t Fin, int32_t Fout, int32_t K) {
  const int NB = (Fout + 31) / 32;
  return K >= 2 && K - 1 <= ST_DMAX && Fin >= 4 && Fin % 4 == 0 && Fout >= 1 && Fout <= 64 &&
         K * NB * 2048 <= ST_WSLICE_BYTES;
}

// layers the kernel runs with four maps per item (StructArgs::pack): two column blocks whatever the layer's width
int struct_packs(int32_t Fin, int32_t Fout) { return Fin == 4 && Fout <= 16 ? 4 : (Fin == 8 && Fout <= 32 ? 2 : 0); }

size_t struct_wfrag_bytes(int32_t Fin, int32_t Fout, int32_t K) {  // (sized for the largest block form, DSPH_PREC_BF16X6)
  const int C = (Fin + 15) / 16, NB = struct_packs(Fin, Fout) ? 2 : (Fout + 31) / 32;
  return (size_t)C * K * NB * ST_WBLK3;
}

int launch_cheb_struct(const StructLaunch& s, hipStream_t stream) {
  // (packed: the per-lane offset of a map inside its group is 32 bits wide)
  // (not for a single map: the two column blocks would be pure overhead -- BASELINE configs[0])
  const int pack = (s.allow_pack && s.N >= 2 && s.x_rows * (int64_t)s.Fin * 4 * 3 < (1ll << 32)) ? struct_packs(s.Fin, s.Fout) : 0;
  const int C = (s.Fin + 15) / 16, NB = pack ? 2 : (s.Fout + 31) / 32;
  // the six-term split needs 3 KiB weight blocks: double-buffered while a slice has at most six of them, replaced in place
  // for K = 5 with 64 columns (cheb_struct_kernel.h); the one shape in between (K = 4, 64 columns) runs exact fp32
  int prec = s.precision;
  if (prec == DSPH_PREC_BF16X6 && !(s.K * NB <= 6 || (s.K == 5 && NB == 2))) prec = DSPH_PREC_FP32;
  if (s.prep_weights) {
    hipLaunchKernelGGL(struct_wprep_kernel, dim3(C * s.K * NB), dim3(256), 0, stream, s.w, s.wfrag, (int)s.Fin_w, (int)s.Fout,
                       (int)s.K, C, NB, prec, (int)s.ld, pack);
    DSPH_HIP(hipGetLastError());
  }
  StructArgs a;
  a.x = s.x;
  a.bias = s.bias;
  a.y = s.y;
  a.wfrag = s.wfrag;
  a.tiles = s.tiles;
  a.gvals8 = s.gvals8;
  a.gdiag = s.gdiag;
  a.tabrow = s.tabrow;
  a.tabvals = s.tabvals;
  a.x_rows = s.x_rows;
  a.y_rows = s.y_rows;
  a.ntiles = s.ntiles;
  a.N = pack ? (int)((s.N + pack - 1) / pack) : (int)s.N;  // (packed: groups of P maps)
  a.n_maps = (int)s.N;
  a.pack = pack;
  a.Fin = s.Fin;
  a.Fout = s.Fout;
  a.K = s.K;
  a.C = C;
  a.act = s.act;
  a.ld = s.ld;
  a.pool = s.pool;
  a.ypool = s.ypool;
  a.ypool_rows = s.ypool_rows;
  const int grid = std::max(8, std::min(s.num_cu, (s.ntiles + 7) / 8 * 8));
#ifdef DSPH_STAMPS
  static unsigned long long* d_stamps = nullptr;
  constexpr size_t NST = 8 * 8 * 32;
  if (!d_stamps) DSPH_HIP(hipMalloc(&d_stamps, NST * 8));
  DSPH_HIP(hipMemsetAsync(d_stamps, 0, NST * 8, stream));
  a.stamps = d_stamps;
#endif
  void (*kern)(StructArgs) = nullptr;
  const bool tab = s.tabrow != nullptr;
  const bool px = a.pack != 0 || a.pool != 0;
#define DSPH_ST_PICK3(N_, P, CH, TB) (px ? cheb_struct_kernel<N_, P, CH, TB, true> : cheb_struct_kernel<N_, P, CH, TB, false>)
#define DSPH_ST_PICK2(P, CH, TB) (NB == 1 ? DSPH_ST_PICK3(1, P, CH, TB) : DSPH_ST_PICK3(2, P, CH, TB))
#define DSPH_ST_PICK(P, CH) (tab ? DSPH_ST_PICK2(P, CH, true) : DSPH_ST_PICK2(P, CH, false))
  if (prec == DSPH_PREC_BF16X3) kern = s.cheb ? DSPH_ST_PICK(DSPH_PREC_BF16X3, true) : DSPH_ST_PICK(DSPH_PREC_BF16X3, false);
  else if (prec == DSPH_PREC_BF16X6) kern = s.cheb ? DSPH_ST_PICK(DSPH_PREC_BF16X6, true) : DSPH_ST_PICK(DSPH_PREC_BF16X6, false);
  else kern = s.cheb ? DSPH_ST_PICK(DSPH_PREC_FP32, true) : DSPH_ST_PICK(DSPH_PREC_FP32, false);
#undef DSPH_ST_PICK
#undef DSPH_ST_PICK2
#undef DSPH_ST_PICK3
  // one workgroup per CU; where the tiles do not fill the device the maps (or groups of maps) of the batch are split over the y dimension
  const int gy = std::max(1, std::min<int>(a.N, s.num_cu / grid));
  hipLaunchKernelGGL(kern, dim3(grid, gy), dim3(ST_THREADS), 0, stream, a);
  DSPH_HIP(hipGetLastError());
#ifdef DSPH_STAMPS
  if (getenv("DSPH_STAMPS_DUMP")) {
    std::vector<unsigned long long> h(NST);
    if (hipStreamSynchronize(stream) == hipSuccess && hipMemcpy(h.data(), d_stamps, NST * 8, hipMemcpyDeviceToHost) == hipSuccess)
      for (int w = 0; w < 8; ++w)
        for (int it = 0; it < 8; ++it) {
          fprintf(stderr, "STSTAMP wave %d item %d:", w, it + 4);
          const unsigned long long* r = &h[((size_t)w * 8 + it) * 32];
          for (int i = 1; i < 18; ++i) fprintf(stderr, " %lld", r[i] && r[i - 1] ? (long long)(r[i] - r[i - 1]) : -1LL);
          fprintf(stderr, " | k2: A %lld B %lld C %lld", (long long)(r[18] - r[6]), (long long)(r[19] - r[18]), (long long)(r[7] - r[19]));
          fprintf(stderr, " | t0 %llu\n", r[0]);
        }
  }
#endif
  return DSPH_OK;
}

}  // namespace dsph
