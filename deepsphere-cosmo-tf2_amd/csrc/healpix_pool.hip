// NEST pooling over the 4^p children of a HEALPix pixel (the step either side of the convolution in every
// reference model).  Replaces the Keras MaxPool1D / AveragePooling1D of healpy_layers.HealpyPool (reference
// healpy_layers.py:48-63: pool_size = strides = 4^p, channels last): in NEST order the children of coarse pixel m are the
// consecutive rows 4^p m .. 4^p (m + 1) - 1, so a pooled row reads one contiguous run of 4^p * F floats.  HBM-bound byte
// work: a lane owns four channels of one output row (16-byte loads, consecutive lanes = consecutive channel quads, then
// consecutive rows), nothing is staged.  Backward: mean -> dy / 4^p to every child; max -> dy to the FIRST child that holds
// the maximum (the children are compared in row order).
#include <algorithm>

#include "dsphere_common.h"

namespace dsph {

template <bool MAXP, int V>  // V = 4: F % 4 == 0 and 16-byte aligned pointers; V = 1 otherwise
__global__ __launch_bounds__(256) void healpix_pool_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t rows_out, int F,
                                                           int group) {
  const int Q = F / V;
  const int64_t total = rows_out * Q;
  const float inv = 1.f / (float)group;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t r = e / Q;
    const int q = (int)(e - r * Q);
    const float* src = x + (r * group) * F + (int64_t)q * V;
    float acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = MAXP ? -__builtin_huge_valf() : 0.f;
    for (int i = 0; i < group; ++i) {
      float t[V];
      if (V == 4) {
        const float4 f = *reinterpret_cast<const float4*>(src + (int64_t)i * F);
        t[0] = f.x; t[1 % V] = f.y; t[2 % V] = f.z; t[3 % V] = f.w;
      } else {
        t[0] = src[(int64_t)i * F];
      }
#pragma unroll
      for (int v = 0; v < V; ++v) acc[v] = MAXP ? fmaxf(acc[v], t[v]) : acc[v] + t[v];
    }
    if (!MAXP) {
#pragma unroll
      for (int v = 0; v < V; ++v) acc[v] *= inv;
    }
    float* dst = y + r * F + (int64_t)q * V;
    if (V == 4) *reinterpret_cast<float4*>(dst) = make_float4(acc[0], acc[1 % V], acc[2 % V], acc[3 % V]);
    else dst[0] = acc[0];
  }
}

// dx of the pooling: one lane per (output row, channel), writes its 4^p children
template <bool MAXP>
__global__ __launch_bounds__(256) void healpix_pool_backward_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                    float* __restrict__ dx, int64_t rows_out, int F, int group) {
  const int64_t total = rows_out * F;
  const float inv = 1.f / (float)group;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t r = e / F;
    const int f = (int)(e - r * F);
    const float g = dy[e];
    const int64_t base = (r * group) * F + f;
    if (!MAXP) {
      for (int i = 0; i < group; ++i) dx[base + (int64_t)i * F] = g * inv;
    } else {
      float m = -__builtin_huge_valf();
      int arg = 0;
      for (int i = 0; i < group; ++i) {
        const float t = x[base + (int64_t)i * F];
        if (t > m) { m = t; arg = i; }
      }
      for (int i = 0; i < group; ++i) dx[base + (int64_t)i * F] = i == arg ? g : 0.f;
    }
  }
}

static unsigned pool_grid(int64_t total) { return (unsigned)std::min<int64_t>((total + 255) / 256, 1 << 20); }

int launch_healpix_pool(const float* x, float* y, int64_t rows_out, int32_t F, int32_t group, bool maxp, hipStream_t stream) {
  if (rows_out <= 0) return DSPH_OK;
  const bool vec = F % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
  const unsigned grid = pool_grid(rows_out * (vec ? F / 4 : F));
  if (vec) {
    if (maxp) hipLaunchKernelGGL((healpix_pool_kernel<true, 4>), dim3(grid), dim3(256), 0, stream, x, y, rows_out, (int)F, (int)group);
    else hipLaunchKernelGGL((healpix_pool_kernel<false, 4>), dim3(grid), dim3(256), 0, stream, x, y, rows_out, (int)F, (int)group);
  } else {
    if (maxp) hipLaunchKernelGGL((healpix_pool_kernel<true, 1>), dim3(grid), dim3(256), 0, stream, x, y, rows_out, (int)F, (int)group);
    else hipLaunchKernelGGL((healpix_pool_kernel<false, 1>), dim3(grid), dim3(256), 0, stream, x, y, rows_out, (int)F, (int)group);
  }
  DSPH_HIP(hipGetLastError());
  return DSPH_OK;
}

int launch_healpix_pool_backward(const float* x, const float* dy, float* dx, int64_t rows_out, int32_t F, int32_t group, bool maxp,
                                 hipStream_t stream) {
  if (rows_out <= 0) return DSPH_OK;
  const unsigned grid = pool_grid(rows_out * F);
  if (maxp) hipLaunchKernelGGL(healpix_pool_backward_kernel<true>, dim3(grid), dim3(256), 0, stream, x, dy, dx, rows_out, (int)F, (int)group);
  else hipLaunchKernelGGL(healpix_pool_backward_kernel<false>, dim3(grid), dim3(256), 0, stream, x, dy, dx, rows_out, (int)F, (int)group);
  DSPH_HIP(hipGetLastError());
  return DSPH_OK;
}

}  // namespace dsph
