// One Chebyshev recurrence step as a padded-ELL gather SpMM on (N, rows, F) planes:
//     out[n,m,:] = alpha * sum_j vals[m,j] * in[n, cols[m,j], :] - beta * prev[n,m,:]
// Replaces one utils.split_sparse_dense_matmul call (reference utils.py:49-78) plus the
// `2 * ... - x0` temporaries of Chebyshev.call (gnn_layers.py:138,141).  The reference first
// re-lays x out as M x Fin*N (gnn_layers.py:131-132); here the caller's (N, M, F) layout is
// read directly: a pixel's F channels are one contiguous 4F-byte row, so every neighbour
// gather is a run of full cache lines when F >= 32.
//
// HBM-bound: 4F bytes in (+ the neighbour rows, which hit L2 because rows are dealt to XCDs in
// contiguous ranges), 4F bytes of prev, 4F bytes out per (n, pixel); the ELL row (8W bytes)
// is re-read per map from L1/L2.  No atomics: each output element has one writer and a fixed
// summation order (j ascending), so results are bitwise reproducible.
#include "dsphere_common.h"

namespace dsph {

template <int VEC>
struct VecT;
template <>
struct VecT<4> {
  using type = float4;
};
template <>
struct VecT<1> {
  using type = float;
};

__device__ __forceinline__ void fma_vec(float4& a, float s, const float4& b) {
  a.x = fmaf(s, b.x, a.x);
  a.y = fmaf(s, b.y, a.y);
  a.z = fmaf(s, b.z, a.z);
  a.w = fmaf(s, b.w, a.w);
}
__device__ __forceinline__ void fma_vec(float& a, float s, const float& b) { a = fmaf(s, b, a); }
__device__ __forceinline__ float4 axpby(float al, const float4& a, float be, const float4& p) {
  return make_float4(al * a.x - be * p.x, al * a.y - be * p.y, al * a.z - be * p.z, al * a.w - be * p.w);
}
__device__ __forceinline__ float axpby(float al, const float& a, float be, const float& p) {
  return al * a - be * p;
}
__device__ __forceinline__ float4 scale(float al, const float4& a) {
  return make_float4(al * a.x, al * a.y, al * a.z, al * a.w);
}
__device__ __forceinline__ float scale(float al, const float& a) { return al * a; }
__device__ __forceinline__ void zero(float4& a) { a = make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void zero(float& a) { a = 0.f; }

// One thread per (row m, channel vector q); the batch is looped inside so the ELL row stays hot.
template <int VEC, typename IDX>
__global__ __launch_bounds__(256) void cheb_step_kernel(
    const int32_t* __restrict__ cols, const float* __restrict__ vals, int W,
    const float* __restrict__ in, int64_t in_rows, const float* __restrict__ prev,
    int64_t prev_rows, float* __restrict__ out, int64_t out_rows, int N, int fv, float alpha,
    float beta, int64_t rows, unsigned nblk) {
  using V = typename VecT<VEC>::type;
  const unsigned bid = xcd_remap(blockIdx.x, nblk);
  const IDX e = (IDX)bid * 256 + (IDX)threadIdx.x;
  if ((int64_t)e >= rows * (int64_t)fv) return;
  const IDX m = e / (IDX)fv;
  const int q = (int)(e - m * (IDX)fv);
  const int32_t* __restrict__ c = cols + (int64_t)m * W;
  const float* __restrict__ v = vals + (int64_t)m * W;
  const V* __restrict__ inv = reinterpret_cast<const V*>(in);
  const V* __restrict__ pv = reinterpret_cast<const V*>(prev);
  V* __restrict__ ov = reinterpret_cast<V*>(out);
  for (int n = 0; n < N; ++n) {
    const int64_t ibase = (int64_t)n * in_rows * fv + q;
    V acc;
    zero(acc);
#pragma unroll 4
    for (int j = 0; j < W; ++j) {
      const V xv = inv[ibase + (int64_t)c[j] * fv];
      fma_vec(acc, v[j], xv);
    }
    V r;
    if (pv != nullptr) {
      const V p = pv[((int64_t)n * prev_rows + (int64_t)m) * fv + q];
      r = axpby(alpha, acc, beta, p);
    } else {
      r = scale(alpha, acc);
    }
    ov[((int64_t)n * out_rows + (int64_t)m) * fv + q] = r;
  }
}

int launch_cheb_step(const dsph_plan* plan, const float* in, int64_t in_rows, const float* prev,
                     int64_t prev_rows, float* out, int64_t out_rows, int64_t N, int32_t F,
                     float alpha, float beta, int64_t rows, hipStream_t stream) {
  if (rows <= 0 || N <= 0) return DSPH_OK;
  if (beta != 0.f && prev == nullptr) {
    set_error("cheb_step: beta != 0 needs prev");
    return DSPH_E_BADARG;
  }
  // graphs wider than the fused kernels' templates (the reference's 20 / 40 neighbours): the step through LDS tiles, every row of
  // `in` read once per tile region instead of once per neighbour (cheb_tstep.hip) -- whole graphs, whole planes
  if (plan->fused != nullptr && plan->opt.tstep && rows == plan->n_rows && in_rows == plan->n_rows && out_rows == plan->n_rows &&
      (prev == nullptr || beta == 0.f || prev_rows == plan->n_rows)) {
    TStepTables tb;
    if (fused_tstep_tables(plan, &tb))
      return launch_cheb_tstep(tb, in, prev, out, rows, N, F, alpha, beta, fused_num_cu(plan), stream);
  }
  const bool vec4 = (F % 4 == 0) && ((reinterpret_cast<uintptr_t>(in) & 15) == 0) &&
                    ((reinterpret_cast<uintptr_t>(out) & 15) == 0) &&
                    (prev == nullptr || (reinterpret_cast<uintptr_t>(prev) & 15) == 0);
  const int fv = vec4 ? F / 4 : F;
  const int64_t total = rows * (int64_t)fv;
  const int64_t nblk64 = (total + 255) / 256;
  if (nblk64 > 0x7fffffffLL) {
    set_error("cheb_step: grid too large (%lld blocks)", (long long)nblk64);
    return DSPH_E_UNSUPPORTED;
  }
  const unsigned nblk = (unsigned)nblk64;
  const bool small = total < (1LL << 31) - 256;
  const float* pr = (beta != 0.f) ? prev : nullptr;
  if (beta != 0.f && prev == nullptr) {
    set_error("cheb_step: beta != 0 needs prev");
    return DSPH_E_BADARG;
  }
#define DSPH_LAUNCH_STEP(VEC, IDX)                                                              \
  hipLaunchKernelGGL((cheb_step_kernel<VEC, IDX>), dim3(nblk), dim3(256), 0, stream,            \
                     plan->d_cols, plan->d_vals, (int)plan->width, in, in_rows, pr, prev_rows,  \
                     out, out_rows, (int)N, fv, alpha, beta, rows, nblk)
  if (vec4) {
    if (small) DSPH_LAUNCH_STEP(4, uint32_t);
    else DSPH_LAUNCH_STEP(4, uint64_t);
  } else {
    if (small) DSPH_LAUNCH_STEP(1, uint32_t);
    else DSPH_LAUNCH_STEP(1, uint64_t);
  }
#undef DSPH_LAUNCH_STEP
  DSPH_HIP(hipGetLastError());
  return DSPH_OK;
}

// ---- boundary-row gather / scatter for the halo exchange of the sharded path ----------------
__global__ __launch_bounds__(256) void rows_pack_kernel(const float* __restrict__ src,
                                                        int64_t src_rows,
                                                        const int32_t* __restrict__ idx,
                                                        int64_t n_idx, float* __restrict__ buf,
                                                        int N, int F, int unpack) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n_idx * (int64_t)F) return;
  const int64_t i = e / F;
  const int f = (int)(e - i * F);
  const int64_t r = idx[i];
  for (int n = 0; n < N; ++n) {
    const int64_t a = ((int64_t)n * src_rows + r) * F + f;  // full-array side
    const int64_t b = ((int64_t)n * n_idx + i) * F + f;     // packed side
    if (unpack) const_cast<float*>(src)[a] = buf[b];
    else buf[b] = src[a];
  }
}

int launch_rows_pack(const float* src, int64_t src_rows, const int32_t* idx, int64_t n_idx,
                     float* buf, int64_t N, int32_t F, bool unpack, hipStream_t stream) {
  if (n_idx <= 0 || N <= 0) return DSPH_OK;
  const int64_t total = n_idx * (int64_t)F;
  const int64_t nblk = (total + 255) / 256;
  hipLaunchKernelGGL(rows_pack_kernel, dim3((unsigned)nblk), dim3(256), 0, stream, src, src_rows,
                     idx, n_idx, buf, (int)N, (int)F, unpack ? 1 : 0);
  DSPH_HIP(hipGetLastError());
  return DSPH_OK;
}

}  // namespace dsph
