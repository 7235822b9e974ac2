// Internal declarations shared by the HIP translation units behind include/dsphere.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "dsphere.h"

namespace dsph {

// thread-local error text returned by dsph_last_error()
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

#define DSPH_HIP(call)                                           \
  do {                                                           \
    hipError_t e__ = (call);                                     \
    if (e__ != hipSuccess) return ::dsph::hip_fail(e__, #call);  \
  } while (0)

// RAII device switch: plans may live on a device other than the caller's current one.
struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
    if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

struct FusedPlan;  // tile decomposition for the single-launch kernel (cheb_fused.hip)

// Per-plan choices of dsph_plan_set_option (include/dsphere.h: DSPH_OPT_*).  They belong to the plan, not to the process:
// the library reads no environment variable on the product path.
struct PlanOptions {
  int strips = 0;            // DSPH_OPT_STRIPS: 0 cost rule per call, 1 always (when the shape is the kernel's), 2 never
  bool use_struct = true;    // DSPH_OPT_STRUCT: structured-tile kernel for the tiles that verify as a 2-D stencil
  bool use_tables = true;    // DSPH_OPT_TABLES: class-T tiles (per-tile tables) on the structured kernel
  bool fork = true;          // DSPH_OPT_FORK: BFS-tile launch on the plan's side stream beside the structured ones
  int strip_seg = 0;         // DSPH_OPT_STRIP_SEG: rows per strip segment (0: chosen by the makespan rule)
  int strip_min_rows = 4;    // DSPH_OPT_STRIP_MINROWS: least height of a strip rectangle, in tiles
  bool strip_generic = false;  // DSPH_OPT_STRIP_GENERIC: compiler-scheduled strip kernel instead of the hand-ordered one
  int split_order = 0;       // DSPH_OPT_SPLIT: K > 5 by the product identity (0 auto, 1 always when possible, 2 never)
  bool tstep = true;         // DSPH_OPT_TSTEP: wide graphs step through LDS tiles (cheb_tstep.hip) instead of the gather kernel
  int strip_form = 0;        // DSPH_OPT_STRIP_FORM: 0 quad strips (cheb_qstrip_kernel.h), 1 strip pairs (cheb_strip_kernel.h)
  bool pack = true;          // DSPH_OPT_PACK: narrow layers run several maps per item / wave when the batch has more than one
  int f16_xexp = 0;          // DSPH_OPT_F16_XEXP: x enters the f16 split of DSPH_PREC_F16X3 times 2^f16_xexp (y times 2^-f16_xexp)
};

}  // namespace dsph

// The opaque plan of the C ABI: the rescaled Laplacian in padded ELL form, resident in HBM.
struct dsph_plan {
  int device = 0;
  int64_t n_rows = 0;   // rows of L~ (one ELL row each)
  int64_t n_cols = 0;   // length of the vectors L~ multiplies (>= n_rows; tail = halo rows)
  int32_t width = 0;    // ELL width W
  int32_t* d_cols = nullptr;  // [n_rows][W] row-major (wave-uniform row reads)
  float* d_vals = nullptr;    // [n_rows][W]
  std::vector<int64_t> levels;  // optional shrinking schedule, see dsph_plan_set_levels
  dsph::PlanOptions opt;
  dsph::FusedPlan* fused = nullptr;
};

namespace dsph {

// out[n,m,:] = alpha * sum_j vals[m,j] * in[n, cols[m,j], :] - beta * prev[n,m,:],  m < rows
int launch_cheb_step(const dsph_plan* plan, const float* in, int64_t in_rows, const float* prev,
                     int64_t prev_rows, float* out, int64_t out_rows, int64_t N, int32_t F,
                     float alpha, float beta, int64_t rows, hipStream_t stream);

int launch_cheb_contract(const float* const* planes, int64_t plane_rows, const float* w,
                         const float* bias, float* y, int64_t N, int64_t rows, int32_t Fin,
                         int32_t Fout, int32_t K, int32_t act, int32_t precision,
                         hipStream_t stream);

bool launch_cheb_tcontract(const float* const* planes, int64_t plane_rows, const float* w, const float* bias, float* y, int64_t N,
                           int64_t rows, int32_t Fin, int32_t Fout, int32_t K, int32_t act, int32_t precision, int num_cu,
                           hipStream_t stream, int* rc);

size_t wgrad_workspace_bytes(int64_t N, int64_t rows, int32_t Fin, int32_t Fout, int32_t K);
int launch_cheb_wgrad(const float* const* planes, int64_t plane_rows, const float* dy, float* dw,
                      int64_t N, int64_t rows, int32_t Fin, int32_t Fout, int32_t K, void* workspace,
                      size_t workspace_bytes, hipStream_t stream);

int launch_rows_pack(const float* src, int64_t src_rows, const int32_t* idx, int64_t n_idx,
                     float* buf, int64_t N, int32_t F, bool unpack, hipStream_t stream);

// fused path (cheb_fused.hip)
FusedPlan* fused_plan_build(const dsph_plan* plan, const int32_t* h_cols, const float* h_vals);
void fused_plan_destroy(FusedPlan* fp);
void fused_plan_invalidate(FusedPlan* fp);
bool fused_host_released(FusedPlan* fp);
int fused_prepare(const dsph_plan* plan, int32_t K, int32_t Fin, int32_t flags);
bool fused_supported(const dsph_plan* plan, int32_t Fin, int32_t Fout, int32_t K);
bool fused_weights_resident(const dsph_plan* plan, int32_t Fin, int32_t Fout, int32_t K);
bool fused_tile_counts(const dsph_plan* plan, int32_t K, int64_t* n_struct, int64_t* n_bfs);
int64_t fused_strip_tiles(const dsph_plan* plan, int64_t N, int32_t Fin, int32_t Fout, int32_t K, int32_t precision);
int64_t fused_strip_pairs(const dsph_plan* plan, int32_t K, int32_t* out, int64_t cap);
int64_t fused_strip_rows(const dsph_plan* plan, int32_t K, int64_t strip, int64_t n, const int32_t* xy, int64_t* rows);
bool fused_strip_split(const dsph_plan* plan, int64_t N, int32_t* grid, int32_t* pieces, int32_t* wg_per_piece, int64_t* tape_rows);
// conv + HealpyPool(p = 1) in one forward: the pooled map (N, n_rows / 4, Fout) and the kind of reduction (1 max, 2 mean)
struct FusedPool {
  float* y;
  int32_t type;
};
bool fused_pool_ok(const dsph_plan* plan, int64_t N, int32_t Fin, int32_t Fout, int32_t K, int32_t act);
int launch_cheb_fused(const dsph_plan* plan, const float* x, const float* w, const float* bias,
                      float* y, int64_t N, int32_t Fin, int32_t Fout, int32_t K, int32_t act,
                      int32_t precision, float alpha_rest, float beta_rest, void* workspace,
                      size_t workspace_bytes, hipStream_t stream, int32_t part = 0, bool keep_weights = false,
                      const FusedPool* pool = nullptr);  // pool: y is then scratch (full resolution), pool->y the result
bool fused_planes_supported(const dsph_plan* plan, int32_t Fin, int32_t K);
int launch_cheb_fused_planes(const dsph_plan* plan, const float* x, float* planes_out, int64_t N, int32_t Fin,
                             int32_t K, float alpha_rest, float beta_rest, hipStream_t stream);
bool fused_wgrad_supported(const dsph_plan* plan, int32_t Fin, int32_t Fout, int32_t K);
size_t fused_wgrad_workspace_bytes(const dsph_plan* plan, int32_t Fin, int32_t Fout, int32_t K);
int launch_fused_pad(const float* x, float* xp, int64_t rows, int32_t Fin, int32_t Fp, hipStream_t stream);  // zero-padded copy, Fp = 4 ceil(Fin / 4)
// the strips' share of dW on the quad-strip weight-gradient kernel, the other tiles' on the BFS-tile kernel (cheb_fused.hip)
bool fused_qwgrad_applies(const dsph_plan* plan, int64_t N, int32_t Fin, int32_t Fout, int32_t K, int32_t precision);
size_t fused_qwgrad_workspace_bytes(const dsph_plan* plan);
int launch_cheb_fused_qwgrad(const dsph_plan* plan, const float* x, const float* dy, float* dw, int64_t N, int32_t Fin, int32_t Fout,
                             int32_t K, float alpha_rest, float beta_rest, void* workspace, size_t bfs_slab_bytes, hipStream_t stream);
int launch_cheb_fused_wgrad(const dsph_plan* plan, const float* x, const float* dy, float* dw, int64_t N,
                            int32_t Fin, int32_t Fout, int32_t K, int32_t precision, float alpha_rest, float beta_rest,
                            void* workspace, size_t workspace_bytes, hipStream_t stream, int32_t Fin_w = 0);  // Fin_w: real channels (rows of dw) when x is a zero-padded copy
size_t fused_workspace_bytes(const dsph_plan* plan, int64_t N, int32_t Fin, int32_t Fout, int32_t K,
                             int32_t precision);
// which packed weight images a workspace block holds (DSPH_FWD_KEEP_WEIGHTS; FusedPlan::images)
enum : uint32_t { IMG_BFS = 1, IMG_STRUCT = 2, IMG_STRIP = 4, IMG_ISTRIP = 8, IMG_QSTRIP = 16, IMG_SPLIT = 32, IMG_Q8 = 64 };
void fused_images_begin(const dsph_plan* plan, const void* ws, uint64_t key, bool keep);
bool fused_images_claim(const dsph_plan* plan, const void* ws, uint32_t bit);
void fused_images_forget(const dsph_plan* plan, const void* ws);
uint64_t fused_images_key(int32_t Fin, int32_t Fin_w, int32_t Fout, int32_t K, int32_t ld, int32_t precision, bool cheb, bool many, bool pack);

// K > 5 as a chain of passes with K <= 5 on the fused kernels (cheb_split.hip)
int fused_dmax();
bool split_applicable(const dsph_plan* plan, int32_t Fin, int32_t Fout, int32_t K);
size_t split_workspace_bytes(const dsph_plan* plan, int64_t N, int32_t Fin, int32_t Fout, int32_t K, int32_t precision);
int split_prepare(const dsph_plan* plan, int32_t K, int32_t Fin, int32_t Fout, int32_t flags);
int launch_split_forward(const dsph_plan* plan, const float* x, const float* w, const float* bias, float* y, int64_t N, int32_t Fin,
                         int32_t Fout, int32_t K, int32_t basis, int32_t act, int32_t precision, void* workspace,
                         size_t workspace_bytes, hipStream_t stream, bool keep_weights = false);

// one recurrence step through LDS tiles, for graphs wider than the fused kernels' templates (cheb_tstep.hip)
struct TStepTables {
  const int32_t* tile_off; const int32_t* ring_end; const int64_t* ell_off; const int32_t* region;
  const uint16_t* lcols; const float* lvals;
  int ntiles = 0, width = 0;
};
int tstep_width(int ell_width);  // table width the tiled step is instantiated for (0: none)
bool fused_tstep_tables(const dsph_plan* plan, TStepTables* out);  // depth-1 tables of a wide whole graph, built on first use
int launch_cheb_tstep(const TStepTables& tb, const float* in, const float* prev, float* out, int64_t rows, int64_t N, int32_t F, float alpha,
                      float beta, int num_cu, hipStream_t stream);
int fused_num_cu(const dsph_plan* plan);

// structured-tile kernel (cheb_struct.hip)
struct StructLaunch {
  const float* x; const float* w; const float* bias; float* y;
  unsigned char* wfrag;      // workspace: struct_wfrag_bytes()
  const int32_t* tiles;      // device list of class-R tiles
  const float* gvals8; const float* gdiag;
  int64_t x_rows, y_rows, N;
  int32_t ntiles, Fin, Fout, K, act, precision, ld, num_cu;
  int32_t Fin_w;             // channels of w (< Fin when x was zero-padded to a multiple of four channels)
  bool cheb;
  const int32_t* tabrow = nullptr;  // class-T tiles: [ntiles][ST_CELLS] rows and [ntiles][ST_CELLS][ST_TABV] values
  const float* tabvals = nullptr;   // (null: class-R tiles, rows by Morton arithmetic, values from gvals8 / gdiag)
  bool prep_weights = true;         // pack the weight fragments first (false: an earlier launch of this forward did)
  bool allow_pack = true;           // DSPH_OPT_PACK of the plan
  float* ypool = nullptr;           // pool != 0: the 2 x 2 pooled output (n_rows / 4 rows per map); y is not written
  int64_t ypool_rows = 0;
  int32_t pool = 0;                 // 0 none, 1 max, 2 mean
};
int struct_build_rows(const dsph_plan* plan, float** gvals8, float** gdiag, unsigned char** flag);
// gdiag[rows[i]] <- vals[9 i], gvals8[rows[i]][d] <- vals[9 i + 1 + d] (host arrays; duplicates carry equal values): the rows of
// class-T tiles that the table-addressed quad strips evaluate, by the directions of the tile's verified embedding
int struct_patch_rows(const dsph_plan* plan, float* gvals8, float* gdiag, const int32_t* rows, const float* vals, int64_t n);
int struct_classify_tiles(const dsph_plan* plan, const unsigned char* d_flag, int ntiles, int D, int64_t out_rows,
                          unsigned char* h_cls, int dlimit = 4 /* ST_DMAX; 7 for the K = 8 quad strips' rectangles */);
bool struct_shape_ok(int32_t Fin, int32_t Fout, int32_t K);
size_t struct_wfrag_bytes(int32_t Fin, int32_t Fout, int32_t K);
int launch_cheb_struct(const StructLaunch& s, hipStream_t stream);
int launch_struct_act(float* y, int64_t rows, int32_t cols, int32_t ld, int32_t act, hipStream_t stream);
int launch_struct_act_tiles(float* y, const int32_t* d_tiles, int ntiles, int64_t N, int64_t y_rows, int32_t cols, int32_t ld,
                            int32_t act, hipStream_t stream);
int launch_residual_epilogue(float* y, const float* skip, int64_t n, float alpha, int32_t act, bool before, hipStream_t stream);

// strip kernel (cheb_strip.hip): rectangles of class-R tiles, streamed in 32-column strips
struct StripPair;
struct StripLaunch {
  const float* x; const float* w; const float* bias; float* y;
  unsigned char* wimg;       // workspace: strip_wimg_bytes()
  const StripPair* pairs;    // device list of strip pairs
  const float* gvals8; const float* gdiag;
  int64_t x_rows, y_rows, N;
  int32_t npairs, Fin, Fout, K, act, precision, ld, num_cu;
  bool cheb;
  bool prep_weights = true;
  bool generic = false;      // DSPH_OPT_STRIP_GENERIC: the compiler-scheduled template also at K = 5
};
bool strip_shape_ok(int32_t Fin, int32_t Fout, int32_t K);
size_t strip_wimg_bytes(int32_t Fin, int32_t Fout, int32_t K);
int launch_cheb_strip(const StripLaunch& s, hipStream_t stream);

// quad-strip kernel (cheb_qstrip.hip, round 5): the same rectangles in 64-column strips, four pixels per lane
struct QStrip;
struct QStripLaunch {
  const float* x; const float* w; const float* bias; float* y;
  unsigned char* wimg;       // workspace: qstrip_wimg_bytes()
  const QStrip* strips;      // device list of strips (uncut along y: the kernel cuts the tape of their rows by workgroup)
  const int32_t* tab = nullptr;  // device: the rectangles' tables of tile bases (QStrip::tab, ::tws)
  const int32_t* prefix;     // device [nstrips + 1]: rows of the strips before each one
  int64_t tape_rows;         // prefix[nstrips]
  const float* gvals8; const float* gdiag;
  int64_t x_rows, y_rows, N;
  int32_t nstrips, Fin, Fout, act, ld, num_cu;
  bool cheb;
  bool f16 = false;          // DSPH_PREC_F16X3: the three-term split on f16 pairs instead of bf16 pairs
  int f16_xexp = 0;          // f16: x is split as x 2^f16_xexp, the store takes the factor out again (DSPH_OPT_F16_XEXP)
  bool prep_weights = true;
};
bool qstrip_shape_ok(int32_t Fin, int32_t Fout, int32_t K);
size_t qstrip_wimg_bytes();
int64_t qstrip_split(int num_cu, int64_t tape_rows, int64_t N, int64_t mean_height, int* grid, int* pieces, int* wg_per_piece);
int launch_cheb_qstrip(const QStripLaunch& s, hipStream_t stream);

// K = 8, 32 -> 32 quad strips (cheb_qstrip8.hip, round 6)
struct QStrip8Launch {
  const float* x; const float* w; const float* bias; float* y;
  unsigned char* wimg;       // workspace: qstrip8_wimg_bytes()
  const QStrip* strips; const int32_t* tab; const int32_t* prefix;
  int64_t tape_rows;
  const float* gvals8; const float* gdiag;
  int64_t x_rows, y_rows, N;
  int32_t nstrips, act, ld, ld_w, num_cu;
  bool f16 = false;          // DSPH_PREC_F16X3
  int f16_xexp = 0;
  bool prep_weights = true;
};
bool qstrip8_shape_ok(int32_t Fin, int32_t Fout, int32_t K);
size_t qstrip8_wimg_bytes();
int64_t qstrip8_split(int num_cu, int64_t tape_rows, int64_t N, int64_t mean_height, int* grid, int* pieces, int* wg_per_piece);
int launch_cheb_qstrip8(const QStrip8Launch& s, hipStream_t stream);

// quad-strip weight gradient (cheb_qwgrad.hip, round 5): dW of a K = 5, 64 -> 64 layer on the strips of the quad-strip kernel
struct QWgradLaunch {
  const float* x; const float* dy;
  float* dw;                 // [64 * 5][lddw]
  float* slabs;              // workspace: qwgrad_slab_bytes(num_cu)
  const QStrip* strips; const int32_t* prefix;
  const int32_t* tab = nullptr;  // the rectangles' tables of tile bases
  int64_t tape_rows;
  const float* gvals8; const float* gdiag;
  int64_t x_rows, dy_rows, N;
  int32_t nstrips, lddy, lddw, num_cu;
  bool cheb;
  bool accumulate = false;   // dw += (the other tiles' share is already there)
};
bool qwgrad_shape_ok(int32_t Fin, int32_t Fout, int32_t K);
size_t qwgrad_slab_bytes(int num_cu);
int launch_cheb_qwgrad(const QWgradLaunch& s, hipStream_t stream);

// NEST pooling (healpix_pool.hip)
int launch_healpix_pool(const float* x, float* y, int64_t rows_out, int32_t F, int32_t group, bool maxp, hipStream_t stream);
int launch_healpix_pool_backward(const float* x, const float* dy, float* dx, int64_t rows_out, int32_t F, int32_t group, bool maxp,
                                 hipStream_t stream);

// input-side strip kernel (cheb_istrip.hip): layers with at most 16 input channels, one wave per strip
struct IStripLaunch {
  const float* x; const float* w; const float* bias; float* y;
  unsigned char* wimg;       // workspace: istrip_wimg_bytes() per 32-column block
  const StripPair* pairs;    // device list; every pair is two single strips
  const float* gvals8; const float* gdiag;
  int64_t x_rows, y_rows, N;
  int32_t npairs, Fin, Fin_w, Fout, K, act, precision, ld, num_cu;
  int32_t nseg = 1;          // row segments per strip (istrip_segments)
  float* ypool = nullptr;    // pool != 0 (one or two input channels only): the 2 x 2 pooled output, ypool_rows rows per map; y is not written
  int64_t ypool_rows = 0;
  int32_t pool = 0;          // 0 none, 1 max, 2 mean
  bool cheb;
  bool prep_weights = true;
};
int istrip_segments(const std::vector<int32_t>& heights, const std::vector<unsigned char>& second, int64_t N, int num_cu, int D,
                    bool narrow);
bool istrip_narrow(int32_t Fin_w);
bool istrip_pairs(int32_t Fin_w, int32_t Fout);
bool istrip_shape_ok(int32_t Fin, int32_t K);
size_t istrip_wimg_bytes(int32_t K, int32_t precision);
int launch_cheb_istrip(const IStripLaunch& s, hipStream_t stream);

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case DSPH_ACT_RELU: return v > 0.f ? v : 0.f;
    case DSPH_ACT_ELU: return v > 0.f ? v : expm1f(v);
    case DSPH_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    case DSPH_ACT_TANH: return tanhf(v);
    default: return v;
  }
}

// Blocks are dealt round-robin over the 8 XCDs (each with a private L2).  Give every XCD one
// contiguous range of tiles so that neighbouring pixel rows -- which gather each other's data --
// share an L2.  Bijective for any grid size.  Speed only; never correctness.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
  const unsigned q = nblk >> 3, r = nblk & 7u, xcd = bid & 7u, i = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}

}  // namespace dsph
