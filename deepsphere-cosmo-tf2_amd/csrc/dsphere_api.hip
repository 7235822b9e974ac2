// C ABI of include/dsphere.h: plan management and the host-side sequencing of the kernels.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <new>

#include "dsphere_common.h"

namespace dsph {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
  set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
  return DSPH_E_HIP;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static int64_t step_rows(const dsph_plan* p, int K, int k) {
  // rows on which recurrence step k (1..K-1) must be valid
  if (p->levels.empty()) return p->n_rows;
  const int lvl = K - 1 - k;
  const int last = (int)p->levels.size() - 1;
  return p->levels[lvl < last ? lvl : last];
}

static int64_t out_rows(const dsph_plan* p) { return p->levels.empty() ? p->n_rows : p->levels[0]; }

}  // namespace dsph

using namespace dsph;

extern "C" {

int dsph_abi_version(void) { return DSPH_ABI_VERSION; }

const char* dsph_last_error(void) { return g_err; }

int dsph_plan_create(dsph_plan** out, int64_t n_rows, int64_t n_cols, int32_t ell_width,
                     const int32_t* cols, const float* vals, int device) {
  if (!out) { set_error("plan_create: out is NULL"); return DSPH_E_BADARG; }
  *out = nullptr;
  if (n_rows <= 0 || n_cols < n_rows || ell_width <= 0 || !cols || !vals) {
    set_error("plan_create: bad shape n_rows=%lld n_cols=%lld width=%d or NULL arrays",
              (long long)n_rows, (long long)n_cols, (int)ell_width);
    return DSPH_E_BADARG;
  }
  if (n_cols > 0x7fffffffLL) {
    set_error("plan_create: n_cols=%lld exceeds int32 column indices", (long long)n_cols);
    return DSPH_E_UNSUPPORTED;
  }
  const int64_t nnz = n_rows * (int64_t)ell_width;
  for (int64_t i = 0; i < nnz; ++i) {
    if (cols[i] < 0 || (int64_t)cols[i] >= n_cols) {
      set_error("plan_create: column %d at row %lld slot %lld outside [0, %lld)", (int)cols[i],
                (long long)(i / ell_width), (long long)(i % ell_width), (long long)n_cols);
      return DSPH_E_BADARG;
    }
  }
  int ndev = 0;
  DSPH_HIP(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) {
    set_error("plan_create: device %d not in [0, %d)", device, ndev);
    return DSPH_E_BADARG;
  }
  DeviceGuard guard(device);
  if (!guard.ok) { set_error("plan_create: cannot select device %d", device); return DSPH_E_HIP; }
  dsph_plan* p = new (std::nothrow) dsph_plan();
  if (!p) { set_error("plan_create: out of host memory"); return DSPH_E_BADARG; }
  p->device = device;
  p->n_rows = n_rows;
  p->n_cols = n_cols;
  p->width = ell_width;
  hipError_t e = hipMalloc((void**)&p->d_cols, (size_t)nnz * sizeof(int32_t));
  if (e == hipSuccess) e = hipMalloc((void**)&p->d_vals, (size_t)nnz * sizeof(float));
  if (e == hipSuccess) e = hipMemcpy(p->d_cols, cols, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(p->d_vals, vals, (size_t)nnz * sizeof(float), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    dsph_plan_destroy(p);
    return hip_fail(e, "plan_create upload");
  }
  p->fused = fused_plan_build(p, cols, vals);  // may be NULL: graph not tileable
  *out = p;
  return DSPH_OK;
}

void dsph_plan_destroy(dsph_plan* p) {
  if (!p) return;
  DeviceGuard guard(p->device);
  if (p->fused) fused_plan_destroy(p->fused);
  if (p->d_cols) (void)hipFree(p->d_cols);
  if (p->d_vals) (void)hipFree(p->d_vals);
  delete p;
}

int dsph_plan_set_levels(dsph_plan* p, int32_t n_levels, const int64_t* rows_at_level) {
  if (!p || n_levels <= 0 || !rows_at_level) { set_error("set_levels: bad arguments"); return DSPH_E_BADARG; }
  int64_t prev = 0;
  for (int i = 0; i < n_levels; ++i) {
    if (rows_at_level[i] < prev || rows_at_level[i] > p->n_rows || rows_at_level[i] <= 0) {
      set_error("set_levels: level %d = %lld not in non-decreasing (0, %lld]", i,
                (long long)rows_at_level[i], (long long)p->n_rows);
      return DSPH_E_BADARG;
    }
    prev = rows_at_level[i];
  }
  if (p->fused && fused_host_released(p->fused)) {  // the tile tables cannot be rebuilt for the new output rows
    set_error("set_levels: the plan's host copy of L~ was released (DSPH_PREPARE_RELEASE_HOST); set the levels before preparing");
    return DSPH_E_UNSUPPORTED;
  }
  p->levels.assign(rows_at_level, rows_at_level + n_levels);
  if (p->fused) {
    DeviceGuard guard(p->device);
    fused_plan_invalidate(p->fused);
  }
  return DSPH_OK;
}

int dsph_plan_set_option(dsph_plan* p, int32_t option, int64_t value) {
  if (!p) { set_error("plan_set_option: plan is NULL"); return DSPH_E_BADARG; }
  PlanOptions o = p->opt;
  bool tables = true;  // the option changes what the tile tables hold
  switch (option) {
    case DSPH_OPT_STRIPS:
      if (value < 0 || value > 2) { set_error("plan_set_option: DSPH_OPT_STRIPS takes 0 (cost rule), 1 (always), 2 (never)"); return DSPH_E_BADARG; }
      o.strips = (int)value;
      break;
    case DSPH_OPT_STRUCT: o.use_struct = value != 0; break;
    case DSPH_OPT_TABLES: o.use_tables = value != 0; break;
    case DSPH_OPT_FORK: o.fork = value != 0; tables = false; break;
    case DSPH_OPT_STRIP_SEG:
      if (value < 0 || value > (1 << 20)) { set_error("plan_set_option: DSPH_OPT_STRIP_SEG takes rows, 0 = automatic"); return DSPH_E_BADARG; }
      o.strip_seg = (int)value;
      break;
    case DSPH_OPT_STRIP_MINROWS:
      if (value < 4 || value > (1 << 20)) { set_error("plan_set_option: DSPH_OPT_STRIP_MINROWS takes tiles, at least 4"); return DSPH_E_BADARG; }
      o.strip_min_rows = (int)value;
      break;
    case DSPH_OPT_STRIP_GENERIC: o.strip_generic = value != 0; tables = false; break;
    case DSPH_OPT_SPLIT:
      if (value < 0 || value > 2) { set_error("plan_set_option: DSPH_OPT_SPLIT takes 0 (automatic), 1 (always), 2 (never)"); return DSPH_E_BADARG; }
      o.split_order = (int)value;
      tables = false;
      break;
    case DSPH_OPT_TSTEP: o.tstep = value != 0; tables = false; break;
    case DSPH_OPT_PACK: o.pack = value != 0; tables = false; break;
    case DSPH_OPT_STRIP_FORM:
      if (value < 0 || value > 1) { set_error("plan_set_option: DSPH_OPT_STRIP_FORM takes 0 (quad strips) or 1 (strip pairs)"); return DSPH_E_BADARG; }
      o.strip_form = (int)value;
      break;
    case DSPH_OPT_F16_XEXP:
      if (value < -100 || value > 100) { set_error("plan_set_option: DSPH_OPT_F16_XEXP takes a binary exponent in [-100, 100]"); return DSPH_E_BADARG; }
      o.f16_xexp = (int)value;
      tables = false;
      break;
    default: set_error("plan_set_option: unknown option %d", (int)option); return DSPH_E_BADARG;
  }
  if (tables && p->fused) {
    if (fused_host_released(p->fused)) {
      set_error("plan_set_option: the plan's host copy of L~ was released (DSPH_PREPARE_RELEASE_HOST); set options before preparing");
      return DSPH_E_UNSUPPORTED;
    }
    DeviceGuard guard(p->device);
    fused_plan_invalidate(p->fused);  // like dsph_plan_set_levels: the tables are rebuilt on the next prepare / forward
  }
  p->opt = o;
  return DSPH_OK;
}

int dsph_plan_strip_pairs(const dsph_plan* p, int32_t K, int32_t* out, int64_t capacity, int64_t* n_pairs) {
  if (!p || !n_pairs || capacity < 0 || (capacity > 0 && !out)) { set_error("plan_strip_pairs: bad arguments"); return DSPH_E_BADARG; }
  const int64_t n = fused_strip_pairs(p, K, out, capacity);
  if (n < 0) { *n_pairs = 0; set_error("plan_strip_pairs: the plan has no fused tables for K = %d", (int)K); return DSPH_E_UNSUPPORTED; }
  *n_pairs = n;
  return DSPH_OK;
}

int dsph_plan_strip_rows(const dsph_plan* p, int32_t K, int64_t strip, int64_t n, const int32_t* xy, int64_t* rows) {
  if (!p || n < 0 || (n > 0 && (!xy || !rows))) { set_error("plan_strip_rows: bad arguments"); return DSPH_E_BADARG; }
  DeviceGuard guard(p->device);
  if (fused_strip_rows(p, K, strip, n, xy, rows) < 0) { set_error("plan_strip_rows: no strip record %lld for K = %d", (long long)strip, (int)K); return DSPH_E_UNSUPPORTED; }
  return DSPH_OK;
}

static bool use_split(const dsph_plan* p, int32_t Fin, int32_t Fout, int32_t K, int32_t algo, int32_t part);

int dsph_plan_strip_split(const dsph_plan* p, int64_t N, int32_t* grid, int32_t* pieces, int32_t* wg_per_piece, int64_t* tape_rows) {
  if (!p || !grid || !pieces || !wg_per_piece || !tape_rows) { set_error("plan_strip_split: NULL argument"); return DSPH_E_BADARG; }
  *grid = *pieces = *wg_per_piece = 0;
  *tape_rows = 0;
  DeviceGuard guard(p->device);
  if (!fused_strip_split(p, N, grid, pieces, wg_per_piece, tape_rows)) { set_error("plan_strip_split: the plan holds no quad strips"); return DSPH_E_UNSUPPORTED; }
  return DSPH_OK;
}

int dsph_plan_prepare_layer(dsph_plan* p, int32_t K, int32_t Fin, int32_t Fout, int32_t flags) {
  if (!p || K <= 0 || Fin <= 0 || Fout <= 0 || (flags & ~(DSPH_PREPARE_BACKWARD | DSPH_PREPARE_RELEASE_HOST))) {
    set_error("plan_prepare: bad arguments (plan %p, K %d, Fin %d, Fout %d, flags %d)", (void*)p, K, Fin, Fout, flags);
    return DSPH_E_BADARG;
  }
  DeviceGuard guard(p->device);
  if (!guard.ok) { set_error("plan_prepare: cannot select device %d", p->device); return DSPH_E_HIP; }
  // a layer with more than five terms may run as a chain of passes (cheb_split.hip): then it is their tables that are
  // built, not the breadth-first tables of depth K - 1 -- by the predicate the forward itself uses (use_split), so that a
  // prepared forward of this (K, Fin, Fout) allocates nothing whichever route it takes.
  if (use_split(p, Fin, Fout, K, DSPH_ALGO_AUTO, DSPH_PART_ALL)) return split_prepare(p, K, Fin, Fout, flags);
  return fused_prepare(p, K, Fin, flags);
}

// (the output width unknown: judged as Fout = Fin; a layer of another width whose route differs builds its tables in its
// first forward -- dsph_plan_prepare_layer takes the width)
int dsph_plan_prepare(dsph_plan* p, int32_t K, int32_t Fin, int32_t flags) { return dsph_plan_prepare_layer(p, K, Fin, Fin, flags); }

// A plan with halo columns and no shrinking schedule cannot run more than one recurrence step by itself: step 2
// would gather halo entries of T_1 that no step wrote (the fused kernels reject it; the unfused path must too).
static bool needs_levels(const dsph_plan* p, int32_t K) { return p->n_cols > p->n_rows && K > 2 && p->levels.empty(); }

int64_t dsph_plan_rows(const dsph_plan* p) { return p ? p->n_rows : 0; }
int64_t dsph_plan_cols(const dsph_plan* p) { return p ? p->n_cols : 0; }
int32_t dsph_plan_ell_width(const dsph_plan* p) { return p ? p->width : 0; }
int64_t dsph_plan_out_rows(const dsph_plan* p, int32_t K) { (void)K; return p ? out_rows(p) : 0; }

int dsph_plan_fused_ok(const dsph_plan* p, int32_t Fin, int32_t Fout, int32_t K) {
  // (tables are built under the plan's device: get_tiles)
  return (p && (fused_supported(p, Fin, Fout, K) || use_split(p, Fin, Fout, K, DSPH_ALGO_AUTO, DSPH_PART_ALL))) ? 1 : 0;
}

int dsph_plan_uses_chain(const dsph_plan* p, int32_t Fin, int32_t Fout, int32_t K) {
  return (p && Fin >= 1 && Fout >= 1 && use_split(p, Fin, Fout, K, DSPH_ALGO_AUTO, DSPH_PART_ALL)) ? 1 : 0;
}

int dsph_plan_tile_counts(const dsph_plan* p, int32_t K, int64_t* n_struct, int64_t* n_bfs) {
  if (!p || !n_struct || !n_bfs) { set_error("plan_tile_counts: NULL argument"); return DSPH_E_BADARG; }
  *n_struct = *n_bfs = 0;
  if (!fused_tile_counts(p, K, n_struct, n_bfs)) { set_error("plan_tile_counts: the fused kernels cannot run this plan with K = %d", K); return DSPH_E_UNSUPPORTED; }
  return DSPH_OK;
}

int dsph_plan_strip_tiles(const dsph_plan* p, int64_t N, int32_t Fin, int32_t Fout, int32_t K, int32_t precision, int64_t* n_tiles) {
  if (!p || !n_tiles) { set_error("plan_strip_tiles: NULL argument"); return DSPH_E_BADARG; }
  *n_tiles = fused_supported(p, Fin, Fout, K) ? fused_strip_tiles(p, N, Fin, Fout, K, precision) : 0;
  return DSPH_OK;
}

// K > 5 by the product identity (cheb_split.hip) instead of the breadth-first-table kernel (K <= 9) or the unfused kernels:
// DSPH_OPT_SPLIT = always whenever the plan allows it; by default beyond the fused kernels' own reach (K >= 10, where the
// alternative is the unfused path's K planes through HBM) and for 6 <= K <= 9 where the measured rule says so (use_split).
static bool use_split(const dsph_plan* p, int32_t Fin, int32_t Fout, int32_t K, int32_t algo, int32_t part) {
  if (K <= 5 || algo == DSPH_ALGO_UNFUSED || part != DSPH_PART_ALL || p->opt.split_order == 2) return false;
  if (p->opt.split_order == 0 && K - 1 <= fused_dmax() && fused_weights_resident(p, Fin, Fout, K)) return false;
  // K = 10 (round 6): one pass over 9-ring regions -- two planes of 1,168 rows fill the LDS, the weights are never resident --
  // beats the chain of three passes at the layers' default arithmetic wherever it exists (tools/k10_routes.py: 16 -> 32 at
  // nside 256 2.10 against 3.71 ms, 64 -> 64 4.64 / 8.00, 32 -> 32 at nside 512 4.03 / 4.90, 8 -> 8 at nside 128 0.38 / 0.58)
  if (p->opt.split_order == 0 && K - 1 == fused_dmax() && fused_supported(p, Fin, Fout, K)) return false;
  return split_applicable(p, Fin, Fout, K);
}

static int resolve_algo(const dsph_plan* p, int32_t Fin, int32_t Fout, int32_t K, int32_t algo) {
  if (algo == DSPH_ALGO_UNFUSED) return DSPH_ALGO_UNFUSED;
  if (fused_supported(p, Fin, Fout, K)) return DSPH_ALGO_FUSED;
  return algo == DSPH_ALGO_FUSED ? -1 : DSPH_ALGO_UNFUSED;
}

int dsph_plan_pool_fusable(const dsph_plan* p, int64_t N, int32_t Fin, int32_t Fout, int32_t K, int32_t act) {
  if (!p || N <= 0 || Fin <= 0 || Fout <= 0 || K <= 0) return 0;
  DeviceGuard guard(p->device);
  return (!use_split(p, Fin, Fout, K, DSPH_ALGO_AUTO, DSPH_PART_ALL) && fused_pool_ok(p, N, Fin, Fout, K, act)) ? 1 : 0;
}

int dsph_poly_forward_pool(const dsph_plan* p, const float* x, const float* w, const float* bias, float* y_scratch,
                           float* y_pooled, int64_t N, int32_t Fin, int32_t Fout, int32_t K, int32_t basis, int32_t act,
                           int32_t precision, int32_t pool_type, int32_t flags, void* workspace, size_t workspace_bytes,
                           void* hip_stream) {
  if (flags & ~DSPH_FWD_KEEP_WEIGHTS) { set_error("poly_forward_pool: unknown flags %d", flags); return DSPH_E_BADARG; }
  if (basis != DSPH_BASIS_CHEBYSHEV && basis != DSPH_BASIS_MONOMIAL) { set_error("poly_forward_pool: unknown basis %d", basis); return DSPH_E_BADARG; }
  if (pool_type != DSPH_POOL_MAX && pool_type != DSPH_POOL_AVG) { set_error("poly_forward_pool: unknown pool type %d", pool_type); return DSPH_E_BADARG; }
  if (!p || !x || !w || !y_pooled || N <= 0 || Fin <= 0 || Fout <= 0 || K <= 0) {
    set_error("poly_forward_pool: bad arguments (NULL pointer or non-positive size)");
    return DSPH_E_BADARG;
  }
  if (precision < DSPH_PREC_FP32 || precision > DSPH_PREC_F16X3) { set_error("poly_forward_pool: unknown precision %d", precision); return DSPH_E_BADARG; }
  if (precision == DSPH_PREC_F16X3) precision = DSPH_PREC_BF16X6;  // (the pooled store is not the quad strips')
  if (!dsph_plan_pool_fusable(p, N, Fin, Fout, K, act)) {
    set_error("poly_forward_pool: no fused pooling for this plan / shape (dsph_plan_pool_fusable)");
    return DSPH_E_UNSUPPORTED;
  }
  const size_t need = dsph_workspace_bytes(p, N, Fin, Fout, K, precision, DSPH_ALGO_FUSED);
  if (need > 0 && (!workspace || workspace_bytes < need)) {
    set_error("poly_forward_pool: workspace %zu bytes, need %zu", workspace_bytes, need);
    return DSPH_E_WORKSPACE;
  }
  DeviceGuard guard(p->device);
  const FusedPool fp{y_pooled, pool_type == DSPH_POOL_MAX ? 1 : 2};
  return launch_cheb_fused(p, x, w, bias, y_scratch, N, Fin, Fout, K, act, precision, basis == DSPH_BASIS_CHEBYSHEV ? 2.f : 1.f,
                           basis == DSPH_BASIS_CHEBYSHEV ? 1.f : 0.f, workspace, workspace_bytes, (hipStream_t)hip_stream,
                           DSPH_PART_ALL, (flags & DSPH_FWD_KEEP_WEIGHTS) != 0, &fp);
}

size_t dsph_workspace_bytes(const dsph_plan* p, int64_t N, int32_t Fin, int32_t Fout, int32_t K,
                            int32_t precision, int32_t algo) {
  if (!p || N <= 0 || Fin <= 0 || K <= 0) return 0;
  if (use_split(p, Fin, Fout, K, algo, DSPH_PART_ALL)) return split_workspace_bytes(p, N, Fin, Fout, K, precision);
  const int a = resolve_algo(p, Fin, Fout, K, algo);
  if (a == DSPH_ALGO_FUSED) return fused_workspace_bytes(p, N, Fin, Fout, K, precision);
  // unfused: planes 1..K-1, each (N, n_cols, Fin) fp32, 256-byte aligned
  const size_t plane = align_up((size_t)N * (size_t)p->n_cols * (size_t)Fin * sizeof(float), 256);
  return plane * (size_t)(K > 1 ? K - 1 : 0);
}

int dsph_cheb_step(const dsph_plan* p, const float* in, const float* prev, float* out, int64_t N,
                   int32_t F, float alpha, float beta, int64_t rows, void* hip_stream) {
  if (!p || !in || !out || N < 0 || F <= 0) { set_error("cheb_step: bad arguments"); return DSPH_E_BADARG; }
  if (rows <= 0) rows = p->n_rows;
  if (rows > p->n_rows) { set_error("cheb_step: rows %lld > plan rows %lld", (long long)rows, (long long)p->n_rows); return DSPH_E_BADARG; }
  DeviceGuard guard(p->device);
  return launch_cheb_step(p, in, p->n_cols, prev, p->n_cols, out, p->n_cols, N, F, alpha, beta,
                          rows, (hipStream_t)hip_stream);
}

int dsph_cheb_contract(const float* const* planes, int64_t plane_rows, const float* w,
                       const float* bias, float* y, int64_t N, int64_t rows, int32_t Fin,
                       int32_t Fout, int32_t K, int32_t act, int32_t precision, int device,
                       void* hip_stream) {
  if (!planes || !w || !y || N < 0 || rows < 0 || rows > plane_rows || Fin <= 0 || Fout <= 0 || K <= 0) {
    set_error("cheb_contract: bad arguments");
    return DSPH_E_BADARG;
  }
  for (int k = 0; k < K; ++k)
    if (!planes[k]) { set_error("cheb_contract: plane %d is NULL", k); return DSPH_E_BADARG; }
  DeviceGuard guard(device);
  return launch_cheb_contract(planes, plane_rows, w, bias, y, N, rows, Fin, Fout, K, act,
                              precision == DSPH_PREC_F16X3 ? DSPH_PREC_BF16X6 : precision, (hipStream_t)hip_stream);
}

int dsph_cheb_forward(const dsph_plan* p, const float* x, const float* w, const float* bias,
                      float* y, int64_t N, int32_t Fin, int32_t Fout, int32_t K, int32_t act,
                      int32_t precision, int32_t algo, void* workspace, size_t workspace_bytes,
                      void* hip_stream) {
  return dsph_poly_forward(p, x, w, bias, y, N, Fin, Fout, K, DSPH_BASIS_CHEBYSHEV, act, precision, algo,
                           workspace, workspace_bytes, hip_stream);
}

int dsph_poly_forward(const dsph_plan* p, const float* x, const float* w, const float* bias,
                      float* y, int64_t N, int32_t Fin, int32_t Fout, int32_t K, int32_t basis,
                      int32_t act, int32_t precision, int32_t algo, void* workspace,
                      size_t workspace_bytes, void* hip_stream) {
  return dsph_poly_forward_part(p, x, w, bias, y, N, Fin, Fout, K, basis, act, precision, algo, DSPH_PART_ALL,
                                workspace, workspace_bytes, hip_stream);
}

int dsph_poly_forward_part(const dsph_plan* p, const float* x, const float* w, const float* bias,
                           float* y, int64_t N, int32_t Fin, int32_t Fout, int32_t K, int32_t basis,
                           int32_t act, int32_t precision, int32_t algo, int32_t part, void* workspace,
                           size_t workspace_bytes, void* hip_stream) {
  return dsph_poly_forward_ex(p, x, w, bias, y, N, Fin, Fout, K, basis, act, precision, algo, part, 0, workspace, workspace_bytes,
                              hip_stream);
}

int dsph_poly_forward_ex(const dsph_plan* p, const float* x, const float* w, const float* bias,
                         float* y, int64_t N, int32_t Fin, int32_t Fout, int32_t K, int32_t basis,
                         int32_t act, int32_t precision, int32_t algo, int32_t part, int32_t flags, void* workspace,
                         size_t workspace_bytes, void* hip_stream) {
  if (flags & ~DSPH_FWD_KEEP_WEIGHTS) { set_error("poly_forward: unknown flags %d", flags); return DSPH_E_BADARG; }
  const bool keep_weights = (flags & DSPH_FWD_KEEP_WEIGHTS) != 0;
  if (part != DSPH_PART_ALL && part != DSPH_PART_INTERIOR && part != DSPH_PART_BOUNDARY) {
    set_error("poly_forward: unknown part %d", part);
    return DSPH_E_BADARG;
  }
  if (basis != DSPH_BASIS_CHEBYSHEV && basis != DSPH_BASIS_MONOMIAL) {
    set_error("poly_forward: unknown basis %d", basis);
    return DSPH_E_BADARG;
  }
  // coefficients of step k >= 2:  T_k = alpha * L~ T_{k-1} - beta * T_{k-2}
  const float alpha_rest = basis == DSPH_BASIS_CHEBYSHEV ? 2.f : 1.f;
  const float beta_rest = basis == DSPH_BASIS_CHEBYSHEV ? 1.f : 0.f;
  if (!p || !x || !w || !y || N < 0 || Fin <= 0 || Fout <= 0 || K <= 0) {
    set_error("cheb_forward: bad arguments (NULL pointer or non-positive size)");
    return DSPH_E_BADARG;
  }
  if (act < DSPH_ACT_NONE || act > DSPH_ACT_TANH) { set_error("cheb_forward: unknown activation %d", act); return DSPH_E_BADARG; }
  if (precision < DSPH_PREC_FP32 || precision > DSPH_PREC_F16X3) { set_error("cheb_forward: unknown precision %d", precision); return DSPH_E_BADARG; }
  if (!p->levels.empty() && (int)p->levels.size() < K - 1) {
    set_error("cheb_forward: plan has %d levels, K = %d needs %d", (int)p->levels.size(), K, K - 1);
    return DSPH_E_BADARG;
  }
  if (needs_levels(p, K)) {
    set_error("cheb_forward: the plan has %lld halo columns and no levels (dsph_plan_set_levels): K = %d would read halo rows of T_1 that nobody writes",
              (long long)(p->n_cols - p->n_rows), K);
    return DSPH_E_BADARG;
  }
  if (N == 0) return DSPH_OK;
  if (use_split(p, Fin, Fout, K, algo, part)) {
    DeviceGuard guard(p->device);
    return launch_split_forward(p, x, w, bias, y, N, Fin, Fout, K, basis, act, precision, workspace, workspace_bytes,
                                (hipStream_t)hip_stream, keep_weights);
  }
  const int a = resolve_algo(p, Fin, Fout, K, algo);
  if (a < 0) { set_error("cheb_forward: fused kernel cannot run this plan/shape (Fin=%d Fout=%d K=%d)", Fin, Fout, K); return DSPH_E_UNSUPPORTED; }
  const size_t need = dsph_workspace_bytes(p, N, Fin, Fout, K, precision, a);
  if (need > 0 && (!workspace || workspace_bytes < need)) {
    set_error("cheb_forward: workspace %zu bytes, need %zu", workspace_bytes, need);
    return DSPH_E_WORKSPACE;
  }
  hipStream_t stream = (hipStream_t)hip_stream;
  DeviceGuard guard(p->device);
  if (a == DSPH_ALGO_FUSED)
    return launch_cheb_fused(p, x, w, bias, y, N, Fin, Fout, K, act, precision, alpha_rest, beta_rest,
                             workspace, workspace_bytes, stream, part, keep_weights);
  if (part != DSPH_PART_ALL) {
    set_error("poly_forward: interior / boundary parts exist for the fused kernel only");
    return DSPH_E_UNSUPPORTED;
  }

  // ---- unfused: K-1 SpMM launches into workspace planes, then one contraction ---------------
  if (K > 64) { set_error("cheb_forward: K = %d exceeds 64", K); return DSPH_E_UNSUPPORTED; }
  const size_t plane = align_up((size_t)N * (size_t)p->n_cols * (size_t)Fin * sizeof(float), 256);
  const float* planes[64];
  planes[0] = x;
  for (int k = 1; k < K; ++k)
    planes[k] = reinterpret_cast<const float*>(static_cast<char*>(workspace) + plane * (size_t)(k - 1));
  for (int k = 1; k < K; ++k) {
    float* outp = const_cast<float*>(planes[k]);
    const int64_t rows = step_rows(p, K, k);
    int rc;
    if (k == 1)
      rc = launch_cheb_step(p, planes[0], p->n_cols, nullptr, p->n_cols, outp, p->n_cols, N, Fin, 1.f, 0.f, rows, stream);
    else
      rc = launch_cheb_step(p, planes[k - 1], p->n_cols, beta_rest != 0.f ? planes[k - 2] : nullptr, p->n_cols, outp,
                            p->n_cols, N, Fin, alpha_rest, beta_rest, rows, stream);
    if (rc != DSPH_OK) return rc;
  }
  return launch_cheb_contract(planes, p->n_cols, w, bias, y, N, out_rows(p), Fin, Fout, K, act,
                              precision == DSPH_PREC_F16X3 ? DSPH_PREC_BF16X6 : precision, stream);
}

int dsph_cheb_planes(const dsph_plan* p, const float* x, float* planes, int64_t N, int32_t Fin, int32_t K,
                     int32_t basis, int32_t algo, void* hip_stream) {
  if (basis != DSPH_BASIS_CHEBYSHEV && basis != DSPH_BASIS_MONOMIAL) {
    set_error("cheb_planes: unknown basis %d", basis);
    return DSPH_E_BADARG;
  }
  if (!p || !x || N < 0 || Fin <= 0 || K <= 0 || (K > 1 && !planes)) {
    set_error("cheb_planes: bad arguments (NULL pointer or non-positive size)");
    return DSPH_E_BADARG;
  }
  if (!p->levels.empty() && (int)p->levels.size() < K - 1) {
    set_error("cheb_planes: plan has %d levels, K = %d needs %d", (int)p->levels.size(), K, K - 1);
    return DSPH_E_BADARG;
  }
  if (needs_levels(p, K)) {
    set_error("cheb_planes: the plan has halo columns and no levels: K = %d is not computable", K);
    return DSPH_E_BADARG;
  }
  if (N == 0 || K == 1) return DSPH_OK;
  const float alpha_rest = basis == DSPH_BASIS_CHEBYSHEV ? 2.f : 1.f;
  const float beta_rest = basis == DSPH_BASIS_CHEBYSHEV ? 1.f : 0.f;
  hipStream_t stream = (hipStream_t)hip_stream;
  DeviceGuard guard(p->device);
  const bool can_fuse = fused_planes_supported(p, Fin, K);
  if (algo == DSPH_ALGO_FUSED && !can_fuse) {
    set_error("cheb_planes: fused kernel cannot run this plan/shape (Fin=%d K=%d)", Fin, K);
    return DSPH_E_UNSUPPORTED;
  }
  if (can_fuse && algo != DSPH_ALGO_UNFUSED)
    return launch_cheb_fused_planes(p, x, planes, N, Fin, K, alpha_rest, beta_rest, stream);
  // one gather launch per step (any L, any Fin)
  const size_t plane = (size_t)N * (size_t)p->n_cols * (size_t)Fin;
  for (int k = 1; k < K; ++k) {
    float* outp = planes + plane * (size_t)(k - 1);
    const float* in = k == 1 ? x : planes + plane * (size_t)(k - 2);
    const float* prev = (k >= 2 && beta_rest != 0.f) ? (k == 2 ? x : planes + plane * (size_t)(k - 3)) : nullptr;
    const int64_t rows = step_rows(p, K, k);
    const int rc = k == 1 ? launch_cheb_step(p, in, p->n_cols, nullptr, p->n_cols, outp, p->n_cols, N, Fin, 1.f, 0.f, rows, stream)
                          : launch_cheb_step(p, in, p->n_cols, prev, p->n_cols, outp, p->n_cols, N, Fin, alpha_rest,
                                             beta_rest, rows, stream);
    if (rc != DSPH_OK) return rc;
  }
  return DSPH_OK;
}

static size_t planes_bytes(const dsph_plan* p, int64_t N, int32_t Fin, int32_t K) {
  return align_up((size_t)(K - 1) * (size_t)N * (size_t)p->n_cols * (size_t)Fin * sizeof(float), 256);
}

size_t dsph_backward_weights_workspace_bytes(const dsph_plan* p, int64_t N, int32_t Fin, int32_t Fout, int32_t K,
                                             int32_t algo) {
  if (!p || N <= 0 || Fin <= 0 || Fout <= 0 || K <= 0) return 0;
  // (channel counts that are no multiple of four -- a first layer's single channel -- run the fused kernel on a zero-padded
  // copy of x behind the slabs: the split-over-pixels kernel of the unfused route moves two pixel rows per MFMA whatever the
  // channel count, 13.5 ms for the 5 x 16 numbers of a 1 -> 16 layer at nside 512)
  const int32_t Fp = (Fin + 3) & ~3;
  if (algo != DSPH_ALGO_UNFUSED && fused_wgrad_supported(p, Fp, Fout, K))
    return align_up(fused_wgrad_workspace_bytes(p, Fp, Fout, K), 256) + (Fp != Fin ? (size_t)N * (size_t)p->n_cols * (size_t)Fp * 4 : 0) +
           (qwgrad_shape_ok(Fin, 64, K) && Fout % 64 == 0 ? align_up(fused_qwgrad_workspace_bytes(p), 256) : 0);  // (the quad strips' slabs)
  if (algo == DSPH_ALGO_FUSED) return 0;
  return planes_bytes(p, N, Fin, K) + wgrad_workspace_bytes(N, out_rows(p), Fin, Fout, K);
}

int dsph_cheb_backward_weights(const dsph_plan* p, const float* x, const float* dy, float* dw, int64_t N,
                               int32_t Fin, int32_t Fout, int32_t K, int32_t basis, int32_t precision, int32_t algo,
                               void* workspace, size_t workspace_bytes, void* hip_stream) {
  if (precision == DSPH_PREC_BF16X6 || precision == DSPH_PREC_F16X3) precision = DSPH_PREC_FP32;  // (only the forward kernels have those forms)
  if (precision != DSPH_PREC_FP32 && precision != DSPH_PREC_BF16X3) {
    set_error("backward_weights: unknown precision %d", precision);
    return DSPH_E_BADARG;
  }
  if (basis != DSPH_BASIS_CHEBYSHEV && basis != DSPH_BASIS_MONOMIAL) {
    set_error("backward_weights: unknown basis %d", basis);
    return DSPH_E_BADARG;
  }
  if (!p || !x || !dy || !dw || N <= 0 || Fin <= 0 || Fout <= 0 || K <= 0) {
    set_error("backward_weights: bad arguments (NULL pointer or non-positive size)");
    return DSPH_E_BADARG;
  }
  if (!p->levels.empty() && (int)p->levels.size() < K - 1) {
    set_error("backward_weights: plan has %d levels, K = %d needs %d", (int)p->levels.size(), K, K - 1);
    return DSPH_E_BADARG;
  }
  if (needs_levels(p, K)) {
    set_error("backward_weights: the plan has halo columns and no levels: K = %d is not computable", K);
    return DSPH_E_BADARG;
  }
  const float alpha_rest = basis == DSPH_BASIS_CHEBYSHEV ? 2.f : 1.f;
  const float beta_rest = basis == DSPH_BASIS_CHEBYSHEV ? 1.f : 0.f;
  hipStream_t stream = (hipStream_t)hip_stream;
  DeviceGuard guard0(p->device);  // the tables behind fused_wgrad_supported live on the plan's device
  const int32_t Fp = (Fin + 3) & ~3;
  const bool can_fuse = fused_wgrad_supported(p, Fp, Fout, K);
  if (algo == DSPH_ALGO_FUSED && !can_fuse) {
    set_error("backward_weights: fused kernel cannot run this plan/shape (Fin=%d Fout=%d K=%d)", Fin, Fout, K);
    return DSPH_E_UNSUPPORTED;
  }
  const size_t need = dsph_backward_weights_workspace_bytes(p, N, Fin, Fout, K, algo);
  if (need > 0 && (!workspace || workspace_bytes < need)) {
    set_error("backward_weights: workspace %zu bytes, need %zu", workspace_bytes, need);
    return DSPH_E_WORKSPACE;
  }
  if (can_fuse && algo != DSPH_ALGO_UNFUSED) {
    DeviceGuard guard(p->device);
    const size_t slab_bytes = align_up(fused_wgrad_workspace_bytes(p, Fp, Fout, K), 256);
    if (Fp != Fin) {
      float* xp = reinterpret_cast<float*>(static_cast<char*>(workspace) + slab_bytes);
      const int rc = launch_fused_pad(x, xp, N * p->n_cols, Fin, Fp, stream);
      if (rc != DSPH_OK) return rc;
      x = xp;
    }
    // K = 5, 64 -> 64 j in the three-term arithmetic on a map the forward runs on the quad strips: their pixels on the
    // quad-strip weight-gradient kernel (cheb_qwgrad.hip), the other tiles on the BFS-tile kernel
    if (Fp == Fin && fused_qwgrad_applies(p, N, Fin, Fout, K, precision))
      return launch_cheb_fused_qwgrad(p, x, dy, dw, N, Fin, Fout, K, alpha_rest, beta_rest, workspace, slab_bytes, stream);
    return launch_cheb_fused_wgrad(p, x, dy, dw, N, Fp, Fout, K, precision, alpha_rest, beta_rest, workspace, slab_bytes, stream, Fin);
  }
  // any L, any shape: K-1 gather launches into workspace planes, then the split-over-pixels MFMA kernel
  if (K > 64) { set_error("backward_weights: K = %d exceeds 64", K); return DSPH_E_UNSUPPORTED; }
  float* planes = static_cast<float*>(workspace);
  int rc = dsph_cheb_planes(p, x, planes, N, Fin, K, basis, DSPH_ALGO_UNFUSED, hip_stream);
  if (rc != DSPH_OK) return rc;
  const size_t plane = (size_t)N * (size_t)p->n_cols * (size_t)Fin;
  const float* pl[64];
  pl[0] = x;
  for (int k = 1; k < K; ++k) pl[k] = planes + plane * (size_t)(k - 1);
  DeviceGuard guard(p->device);
  const size_t pb = planes_bytes(p, N, Fin, K);
  return launch_cheb_wgrad(pl, p->n_cols, dy, dw, N, out_rows(p), Fin, Fout, K, static_cast<char*>(workspace) + pb,
                           workspace_bytes - pb, stream);
}

size_t dsph_wgrad_workspace_bytes(int64_t N, int64_t rows, int32_t Fin, int32_t Fout, int32_t K) {
  if (N <= 0 || rows <= 0 || Fin <= 0 || Fout <= 0 || K <= 0) return 0;
  return wgrad_workspace_bytes(N, rows, Fin, Fout, K);
}

int dsph_cheb_wgrad(const float* const* planes, int64_t plane_rows, const float* dy, float* dw,
                    int64_t N, int64_t rows, int32_t Fin, int32_t Fout, int32_t K, void* workspace,
                    size_t workspace_bytes, int device, void* hip_stream) {
  if (!planes || !dy || !dw || N <= 0 || rows <= 0 || rows > plane_rows || Fin <= 0 || Fout <= 0 || K <= 0) {
    set_error("cheb_wgrad: bad arguments");
    return DSPH_E_BADARG;
  }
  for (int k = 0; k < K && k < 64; ++k)
    if (!planes[k]) { set_error("cheb_wgrad: plane %d is NULL", k); return DSPH_E_BADARG; }
  DeviceGuard guard(device);
  return launch_cheb_wgrad(planes, plane_rows, dy, dw, N, rows, Fin, Fout, K, workspace, workspace_bytes,
                           (hipStream_t)hip_stream);
}

int dsph_rows_pack(const float* src, int64_t src_rows, const int32_t* idx, int64_t n_idx,
                   float* buf, int64_t N, int32_t F, int device, void* hip_stream) {
  if (!src || !idx || !buf || n_idx < 0 || N < 0 || F <= 0) { set_error("rows_pack: bad arguments"); return DSPH_E_BADARG; }
  DeviceGuard guard(device);
  return launch_rows_pack(src, src_rows, idx, n_idx, buf, N, F, false, (hipStream_t)hip_stream);
}

int dsph_rows_unpack(float* dst, int64_t dst_rows, const int32_t* idx, int64_t n_idx,
                     const float* buf, int64_t N, int32_t F, int device, void* hip_stream) {
  if (!dst || !idx || !buf || n_idx < 0 || N < 0 || F <= 0) { set_error("rows_unpack: bad arguments"); return DSPH_E_BADARG; }
  DeviceGuard guard(device);
  return launch_rows_pack(dst, dst_rows, idx, n_idx, const_cast<float*>(buf), N, F, true, (hipStream_t)hip_stream);
}

static int pool_args_ok(const void* a, const void* b, int64_t N, int64_t rows_out, int32_t F, int32_t group, int32_t type) {
  if (!a || !b || N < 0 || rows_out < 0 || F <= 0 || group <= 0 || (type != DSPH_POOL_MAX && type != DSPH_POOL_AVG)) {
    set_error("healpix_pool: bad arguments (NULL pointer, negative size or unknown pooling type %d)", (int)type);
    return DSPH_E_BADARG;
  }
  if (group > (1 << 20)) { set_error("healpix_pool: group %d too large", (int)group); return DSPH_E_UNSUPPORTED; }
  return DSPH_OK;
}

int dsph_healpix_pool(const float* x, float* y, int64_t N, int64_t rows_out, int32_t F, int32_t group, int32_t type, int device,
                      void* hip_stream) {
  const int rc = pool_args_ok(x, y, N, rows_out, F, group, type);
  if (rc != DSPH_OK) return rc;
  DeviceGuard guard(device);
  return launch_healpix_pool(x, y, N * rows_out, F, group, type == DSPH_POOL_MAX, (hipStream_t)hip_stream);
}

int dsph_healpix_pool_backward(const float* x, const float* dy, float* dx, int64_t N, int64_t rows_out, int32_t F, int32_t group,
                               int32_t type, int device, void* hip_stream) {
  const int rc = pool_args_ok(dy, dx, N, rows_out, F, group, type);
  if (rc != DSPH_OK) return rc;
  if (type == DSPH_POOL_MAX && !x) { set_error("healpix_pool_backward: max pooling needs the forward input"); return DSPH_E_BADARG; }
  DeviceGuard guard(device);
  return launch_healpix_pool_backward(x, dy, dx, N * rows_out, F, group, type == DSPH_POOL_MAX, (hipStream_t)hip_stream);
}

int dsph_residual_epilogue(float* y, const float* skip, int64_t n, float alpha, int32_t act, int32_t act_before,
                           int device, void* hip_stream) {
  if (!y || !skip || n < 0) { set_error("residual_epilogue: bad arguments"); return DSPH_E_BADARG; }
  if (act < DSPH_ACT_NONE || act > DSPH_ACT_TANH) { set_error("residual_epilogue: unknown activation %d", act); return DSPH_E_BADARG; }
  DeviceGuard guard(device);
  return launch_residual_epilogue(y, skip, n, alpha, act, act_before != 0, (hipStream_t)hip_stream);
}

}  // extern "C"
