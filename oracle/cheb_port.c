/* cheb_port.c -- TEST INFRASTRUCTURE (the timed CPU baseline of bench.py; never imported by the product).
 *
 * Multi-threaded fp32 stand-ins for the two TensorFlow kernels and the re-layouts that Chebyshev.call strings
 * together (/root/reference/src/deepsphere/gnn_layers.py:131-150): TensorFlow is installed on neither box, and its
 * CPU kernels (Eigen thread pool) are multi-threaded, so a fair "reference path on the host cores" needs threads too.
 * The op SEQUENCE and every materialisation are the reference's; only the loops inside an op are OpenMP.
 *   gcc -O3 -march=native -fopenmp -shared -fPIC oracle/cheb_port.c -o oracle/_build/libcheb_port.so
 */
#include <stdint.h>
#include <string.h>

/* x0 = reshape(transpose(x, [1, 2, 0]), [M, Fin*N])   (gnn_layers.py:131-132): x (N, M, Fin) -> x0[m][f*N + n] */
void port_relayout_in(const float* x, float* x0, int64_t N, int64_t M, int64_t Fin, int threads) {
#pragma omp parallel for num_threads(threads) schedule(static)
  for (int64_t m = 0; m < M; ++m)
    for (int64_t f = 0; f < Fin; ++f)
      for (int64_t n = 0; n < N; ++n) x0[m * Fin * N + f * N + n] = x[(n * M + m) * Fin + f];
}

/* out = alpha * (L @ in) - beta * prev   with L in padded ELL [M][W] and in/out/prev dense [M][C]
 * (tf.sparse.sparse_dense_matmul, utils.py:73-76, plus the `2 * ... - x0` temporaries of gnn_layers.py:141,
 * fused into one pass here -- which favours the baseline) */
void port_spmm(const int32_t* cols, const float* vals, int64_t M, int64_t W, const float* in, const float* prev,
               float* out, int64_t C, float alpha, float beta, int threads) {
#pragma omp parallel for num_threads(threads) schedule(static)
  for (int64_t m = 0; m < M; ++m) {
    float* o = out + m * C;
    for (int64_t c = 0; c < C; ++c) o[c] = 0.f;
    for (int64_t j = 0; j < W; ++j) {
      const float v = vals[m * W + j];
      if (v == 0.f) continue;
      const float* r = in + (int64_t)cols[m * W + j] * C;
      for (int64_t c = 0; c < C; ++c) o[c] += v * r[c];
    }
    if (beta != 0.f || alpha != 1.f) {
      const float* p = prev ? prev + m * C : 0;
      for (int64_t c = 0; c < C; ++c) o[c] = alpha * o[c] - (p ? beta * p[c] : 0.f);
    }
  }
}

/* X = reshape(transpose(reshape(stack(planes), [K, M, Fin, N]), [3, 1, 2, 0]), [N*M, Fin*K])   (gnn_layers.py:144-147)
 * planes: K pointers to [M][Fin*N]; X[(n*M + m)][f*K + k] */
void port_relayout_out(const float* const* planes, float* X, int64_t K, int64_t N, int64_t M, int64_t Fin, int threads) {
#pragma omp parallel for num_threads(threads) schedule(static)
  for (int64_t m = 0; m < M; ++m)
    for (int64_t n = 0; n < N; ++n) {
      float* row = X + (n * M + m) * Fin * K;
      for (int64_t f = 0; f < Fin; ++f)
        for (int64_t k = 0; k < K; ++k) row[f * K + k] = planes[k][m * Fin * N + f * N + n];
    }
}
