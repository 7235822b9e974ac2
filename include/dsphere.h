/* dsphere.h -- C ABI of the MI355X-native Chebyshev graph-convolution forward.
 *
 * Drop-in boundary for ONE path of deepsphere/deepsphere-cosmo-tf2: the forward of
 * `deepsphere.gnn_layers.Chebyshev` (reference src/deepsphere/gnn_layers.py:106-161) with
 * its one-time Laplacian upload (gnn_layers.py:64-72).  The reference has no FFI of its
 * own for this path -- it calls TensorFlow ops from Python -- so every entry point below
 * names the Python/TF call sites it replaces.  Plain pointers and sizes only; no torch
 * types.  All device pointers are HIP device memory on the plan's device; the caller
 * (PyTorch) owns x, w, bias, y and the workspace.  Functions return 0 on success or a
 * negative DSPH_E_* code; the message is available from dsph_last_error() (thread-local).
 *
 * Allocation and synchronisation: dsph_plan_create, dsph_plan_prepare, dsph_plan_set_levels and
 * dsph_plan_destroy allocate / free device memory and synchronise.  The compute entry points
 * (dsph_cheb_*, dsph_poly_*, dsph_rows_*) only enqueue kernels on the caller's stream -- EXCEPT that
 * the first fused call for a (plan, K) that has not been through dsph_plan_prepare builds the tile
 * tables on the spot (device allocation + synchronous copies).  Call dsph_plan_prepare once per K
 * before timing, and before capturing a forward into a hipGraph; after it, a forward neither
 * allocates nor synchronises (tests/test_gpu_parity.py::test_prepared_forward_*).
 */
#ifndef DSPHERE_H
#define DSPHERE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSPH_ABI_VERSION 3  /* 3: DSPH_OPT_F16_XEXP, dsph_plan_strip_rows, dsph_plan_uses_chain (round 6); 2: dsph_plan_prepare_layer; the entry points added
                              * since version 1 (set_option, forward_ex, forward_pool, healpix_pool, strip_pairs) are part of it */

/* error codes */
#define DSPH_OK 0
#define DSPH_E_BADARG (-1)      /* null pointer, negative size, inconsistent shapes */
#define DSPH_E_HIP (-2)         /* a HIP runtime call failed */
#define DSPH_E_UNSUPPORTED (-3) /* shape outside what the kernels implement */
#define DSPH_E_WORKSPACE (-4)   /* workspace too small */

/* fused epilogue activations (reference: tf.keras.activations looked up by name,
 * gnn_layers.py:55-60; anything else is applied by the host layer after the call) */
#define DSPH_ACT_NONE 0
#define DSPH_ACT_RELU 1
#define DSPH_ACT_ELU 2
#define DSPH_ACT_SIGMOID 3
#define DSPH_ACT_TANH 4

/* arithmetic of the dense [K*Fin]x[Fout] contraction (the recurrence is always fp32) */
#define DSPH_PREC_FP32 0   /* v_mfma_f32_32x32x2_f32: bitwise an fp32 fma chain        */
#define DSPH_PREC_BF16X3 1 /* hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16, fp32 accumulate */
/* fp32-equivalent on the bf16 matrix pipe: both operands split three ways (8 + 8 + 8 mantissa bits), the six products
 * down to 2^-16 kept (hh, hm, mh, mm, hl, lh), fp32 accumulate: what is dropped is 2^-23 of a product, the size of an
 * fp32 rounding.  A third of the matrix-pipe time of DSPH_PREC_FP32.  The structured-tile kernel implements it; every
 * other kernel (BFS tiles, unfused, weight gradient) runs DSPH_PREC_FP32 when asked for it. */
#define DSPH_PREC_BF16X6 2
/* fp32-equivalent at the three-term split's price, where the quad-strip kernels run (csrc/cheb_qstrip_kernel.h: K = 5, 64 input
 * and 64 output channels per column block; csrc/cheb_qstrip8_kernel.h: K = 8, 32 -> 32; the rectangles of a HEALPix map): both operands split hi + lo into f16 (11 + 11
 * mantissa bits), hi.hi + hi.lo + lo.hi on v_mfma_f32_16x16x32_f16, fp32 accumulate: what is dropped is 2^-22 of a product.
 * The weights go in times a power of two of the library's choosing (taken out again in the store); x goes in times 2^e, e =
 * DSPH_OPT_F16_XEXP of the plan (default 0), taken out again in the store as well.  The RANGE CONDITION is the caller's to meet:
 *   upper: |x| 2^e < 65,504 -- an input beyond it does not fit an f16 and comes out as Inf / NaN rows of y (loud);
 *   lower: an f16 pair keeps 22 bits only where both halves are normal numbers, |x| 2^e >= 2^-3; below that the lo half goes
 *          subnormal and the absolute error of an input element stays at 2^-25 (2^-e of it in x's units): with unit-variance x and e
 *          = 0 the result is 3e-7 of max|y| from float64, with x of scale 1e-3 and e = 0 it is 2e-5 -- WORSE than DSPH_PREC_BF16X3
 *          (silent).  Choose e so that max|x| 2^e lies in [2^13, 2^15): every element within 2^-16 of the maximum then keeps its
 *          22 bits and the smaller ones err by 2^-38 of the maximum (deepsphere/gnn_layers.py does this from `x_absmax`, or
 *          from a reduction over x when the caller names no scale).
 * Every tile, shape and kernel the quad strips do not take runs DSPH_PREC_BF16X6 (same accuracy, no range condition). */
#define DSPH_PREC_F16X3 3

/* polynomial basis of the recurrence: T_1 = L~ x in both;
 *   Chebyshev: T_k = 2 L~ T_{k-1} - T_{k-2}   (reference gnn_layers.Chebyshev, gnn_layers.py:137-143)
 *   monomial:  T_k = L~ T_{k-1}                (reference gnn_layers.Monomial,  gnn_layers.py:283-286) */
#define DSPH_BASIS_CHEBYSHEV 0
#define DSPH_BASIS_MONOMIAL 1

/* which implementation dsph_cheb_forward runs */
#define DSPH_ALGO_AUTO 0
#define DSPH_ALGO_UNFUSED 1 /* K-1 ELL SpMM launches into workspace planes + one contraction launch */
#define DSPH_ALGO_FUSED 2   /* one launch: tile + halo in LDS, planes never leave the CU */

typedef struct dsph_plan dsph_plan;

/* Upload the rescaled Laplacian L~ once.
 * Replaces: the tf.constant triple built in Chebyshev.__init__ (gnn_layers.py:68-72) and the
 * per-call tf.SparseTensor + tf.sparse.reorder (gnn_layers.py:114-115).
 *   n_rows      rows of L~ held by this plan (outputs are produced for rows [0, n_rows))
 *   n_cols      length of the vectors L~ is applied to; n_cols >= n_rows.  Equal for a whole
 *               graph; larger for a shard whose trailing n_cols - n_rows entries are halo rows
 *               received from other shards.
 *   ell_width   padded row width W (max non-zeros per row, diagonal included)
 *   cols, vals  HOST arrays [n_rows][ell_width], row-major; padding entries carry val 0 and any
 *               valid column (conventionally the row itself)
 *   device      HIP device ordinal
 */
int dsph_plan_create(dsph_plan** out, int64_t n_rows, int64_t n_cols, int32_t ell_width,
                     const int32_t* cols, const float* vals, int device);

void dsph_plan_destroy(dsph_plan* plan);

/* Optional shrinking schedule for a plan whose rows are ordered by graph distance from the
 * rows it owns (sharded, deep-halo mode): rows_at_level[j] = number of leading rows within j
 * hops of the owned rows, j = 0..n_levels-1, non-decreasing, rows_at_level[0] = owned rows,
 * rows_at_level[n_levels-1] <= n_rows.  With a schedule, recurrence step k of a K-term forward
 * is evaluated on rows_at_level[K-1-k] rows (so a K-term forward needs n_levels >= K-1) and y
 * is written for rows_at_level[0] rows.  Without one every step covers n_rows rows.
 * Host array, copied. */
int dsph_plan_set_levels(dsph_plan* plan, int32_t n_levels, const int64_t* rows_at_level);

/* Build everything the fused kernels need from this plan for a K-term layer with Fin input channels, now
 * instead of inside the first forward: the direction-ordered copy of L~ and the per-row / per-tile verification
 * of the 2-D stencil structure (structured-tile kernel), and the breadth-first ring tables of the remaining
 * tiles (BFS-tile kernel).  Allocates device memory, launches set-up kernels and synchronises.
 *   flags  DSPH_PREPARE_BACKWARD      also the tables of dsph_cheb_planes / dsph_cheb_backward_weights, and the look at the
 *                                     matrix that dsph_cheb_backward_weights otherwise takes on its first call (is it
 *                                     symmetric: one host pass over the ELL arrays, about a second at 12.6 M rows)
 *          DSPH_PREPARE_RELEASE_HOST  afterwards drop the plan's host copy of the ELL arrays (kept by
 *                                     dsph_plan_create to build tables for further K); preparing another K
 *                                     later then fails with DSPH_E_UNSUPPORTED and forwards with that K take
 *                                     the unfused path under DSPH_ALGO_AUTO; dsph_plan_set_levels then returns
 *                                     DSPH_E_UNSUPPORTED (it would have to rebuild the tables), and the tables of
 *                                     dsph_cheb_planes / dsph_cheb_backward_weights exist only if DSPH_PREPARE_BACKWARD
 *                                     was passed in the same call
 * Returns DSPH_OK also when the fused kernels cannot run the plan (dsph_plan_fused_ok says which). */
#define DSPH_PREPARE_BACKWARD 1
#define DSPH_PREPARE_RELEASE_HOST 2
int dsph_plan_prepare(dsph_plan* plan, int32_t K, int32_t Fin, int32_t flags);
/* The same for a layer whose output width is known: a K > 5 layer runs either on the breadth-first tables of depth K - 1 or as
 * a chain of K <= 5 passes (DSPH_OPT_SPLIT), and which of the two depends on Fin AND Fout; this call builds exactly the tables
 * the forward of (K, Fin, Fout) will use, by the forward's own rule.  dsph_plan_prepare(plan, K, Fin, flags) is this call with
 * Fout = Fin. */
int dsph_plan_prepare_layer(dsph_plan* plan, int32_t K, int32_t Fin, int32_t Fout, int32_t flags);

/* Per-plan choices.  The library reads no environment variable: what used to be process-global switches of the fused path
 * are options of the plan, set before the tables of a K are built (dsph_plan_prepare or the first forward) -- setting one that
 * changes the tables drops the cached ones, like dsph_plan_set_levels, and after DSPH_PREPARE_RELEASE_HOST returns
 * DSPH_E_UNSUPPORTED.  Not thread-safe against forwards in flight on the same plan.
 *   DSPH_OPT_STRIPS         0 (default): whether the strip kernel takes its rectangles is decided per call by a cost rule that
 *                           depends on the batch N and the device's CU count (csrc/cheb_fused.hip, strips_apply) -- so the
 *                           same map on the same plan can be summed in Clenshaw order at one batch size and in forward order
 *                           at another, equal to rounding, not bit for bit; 1: always (when the shape is the kernel's);
 *                           2: never.  1 and 2 make the choice independent of batch, device and sharding.
 *   DSPH_OPT_STRUCT         1 (default) / 0: the structured-tile kernel for tiles that verify as a 2-D stencil; 0 sends
 *                           every tile to the breadth-first-table kernel
 *   DSPH_OPT_TABLES         1 (default) / 0: per-tile tables (base-pixel borders, halo rows) on the structured kernel
 *   DSPH_OPT_FORK           1 (default) / 0: the BFS-tile launch of a large forward on a plan-owned side stream
 *   DSPH_OPT_STRIP_SEG      rows per strip segment, 0 (default) = chosen by the makespan rule          (tuning)
 *   DSPH_OPT_STRIP_MINROWS  least height of a strip rectangle in tiles, default 4                      (tuning)
 *   DSPH_OPT_STRIP_GENERIC  0 (default) / 1: the compiler-scheduled strip kernel at K = 5              (diagnosis)
 *   DSPH_OPT_STRIP_FORM     which strip kernel takes the rectangles of the 64 -> 64, K = 5, three-term shape: 0 (default) the
 *                           quad strips of round 5 (64-column strips, four pixels per lane, csrc/cheb_qstrip_kernel.h),
 *                           1 the strip pairs of round 3 (32-column strips, csrc/cheb_strip_kernel.h).  Same rectangles, same
 *                           arithmetic, another order of summation inside a row: equal to rounding, not bit for bit.
 *   DSPH_OPT_SPLIT          K > 5: 0 (default) the faster of the two routes by rule, 1 always the product identity
 *                           T_{4+j} = 2 T_4 T_j - T_{|4-j|} (passes of K <= 5 on the fast kernels), 2 never
 *   DSPH_OPT_TSTEP          1 (default) / 0: graphs wider than the fused kernels take (ELL width 13 .. 32: the reference's 20
 *                           neighbours) run every recurrence step through LDS tiles (csrc/cheb_tstep.hip); 0: the gather kernel.
 *                           Same bits either way.
 *   DSPH_OPT_PACK           1 (default) / 0: a layer with at most four input channels and at most 16 output columns (the first
 *                           layers of a network; likewise eight channels and 32 columns, two maps) runs its batch four maps to
 *                           an item on the tile kernels when the batch has more
 *                           than one map (the input-side strips put two maps on a wave whatever the batch: map n always rides
 *                           half n & 1, its bits do not depend on the batch).  A map then equals its single-map result
 *                           to rounding, not bit for bit (another order of exact-zero products, and at K = 4 the fp32-equivalent
 *                           arithmetic in its exact-fp32 form); 0 keeps every map's bits independent of the batch. */
#define DSPH_OPT_STRIPS 1
#define DSPH_OPT_STRUCT 2
#define DSPH_OPT_TABLES 3
#define DSPH_OPT_FORK 4
#define DSPH_OPT_STRIP_SEG 5
#define DSPH_OPT_STRIP_MINROWS 6
#define DSPH_OPT_STRIP_GENERIC 7
#define DSPH_OPT_SPLIT 8
#define DSPH_OPT_TSTEP 9
#define DSPH_OPT_PACK 10
#define DSPH_OPT_STRIP_FORM 11
/*   DSPH_OPT_F16_XEXP       binary exponent e in [-100, 100], default 0: under DSPH_PREC_F16X3 the quad strips split x 2^e into
 *                           f16 pairs and store y 2^-e (exact both ways); see DSPH_PREC_F16X3 for how to choose it.  Read when a
 *                           forward is enqueued: it may change between forwards (not while one is being enqueued on this plan). */
#define DSPH_OPT_F16_XEXP 12
int dsph_plan_set_option(dsph_plan* plan, int32_t option, int64_t value);

int64_t dsph_plan_rows(const dsph_plan* plan);
int64_t dsph_plan_cols(const dsph_plan* plan);
int32_t dsph_plan_ell_width(const dsph_plan* plan);
int64_t dsph_plan_out_rows(const dsph_plan* plan, int32_t K); /* rows y is produced for */
/* 1 if the fused kernels can run this (plan, shape) -- in one forward, or, for K > 5 where the plan's options allow it, as the
 * chain of <= 5-term passes; 0 otherwise (the unfused kernels then serve dsph_cheb_forward under DSPH_ALGO_AUTO) */
int dsph_plan_fused_ok(const dsph_plan* plan, int32_t Fin, int32_t Fout, int32_t K);
/* 1 if a forward of this (plan, shape) runs as the chain of <= 5-term passes (K > 5: the product identity, csrc/cheb_split.hip) --
 * every pass of the chain rounds its input to bf16 hi + lo again under DSPH_PREC_BF16X3, so a caller that wants 1e-5 asks for the
 * six-term split there; 0 if one pass of the fused kernels (K <= 10 on the 8-neighbour grid) or the unfused kernels serve it.
 * (Round 6: K = 10 -- the reference tutorials' order -- runs in ONE pass of the breadth-first tile kernel over 9-ring regions
 * where the plan's regions fit its 1,168-row planes, i.e. on the 8-neighbour grid stencil; on wider graphs it stays the chain.) */
int dsph_plan_uses_chain(const dsph_plan* plan, int32_t Fin, int32_t Fout, int32_t K);

/* How the fused forward with K terms splits the plan's 256-row tiles between its two kernels: *n_struct tiles whose
 * (K-1)-ring region was verified to be a square of a 2-D 9-point stencil (structured-tile kernel,
 * csrc/cheb_struct_kernel.h: either the rows of the region are a plain Z-order continuation of the tile's, or the
 * region was laid out from the graph and is addressed through per-tile tables -- base-pixel borders of the sphere, halo
 * rows of a sharded plan) and *n_bfs tiles handled through breadth-first ring tables (csrc/cheb_fused_kernel.h).
 * DSPH_E_UNSUPPORTED when neither kernel can run the plan (the unfused path then serves dsph_cheb_forward). */
int dsph_plan_tile_counts(const dsph_plan* plan, int32_t K, int64_t* n_struct, int64_t* n_bfs);

/* How many of those structured tiles a forward of this shape hands to the strip kernel instead (csrc/cheb_strip_kernel.h:
 * rectangles of at least 3 x 4 tiles whose regions are plain Z-order squares, streamed row by row in 32-column strips with
 * the recurrence in registers -- Clenshaw's backward form, fed by the MFMA).  0 when the strip kernel does not take the shape
 * (it implements K = 5, Fin = Fout = 64 per 64-column block, DSPH_PREC_BF16X3), the plan holds no such rectangle, or the
 * strips would not pay for a batch of N maps (its work items are (pair of strips, map): small or ragged plans need a batch
 * to fill the device; the rule is in csrc/cheb_fused.hip, strips_apply); the other tiles are served as dsph_plan_tile_counts
 * says.  Same summation per output whichever kernel writes it?  No: the two
 * forms round differently (both within the tolerance of their precision), so outputs of tiles that change hands between
 * two plans of the same graph agree to rounding, not bit for bit. */
int dsph_plan_strip_tiles(const dsph_plan* plan, int64_t N, int32_t Fin, int32_t Fout, int32_t K, int32_t precision,
                          int64_t* n_tiles);

/* The strip kernel's work list for K terms, for tests and tools: `out` receives up to `capacity` records of twelve int32 --
 * x0[2], w[2] (first output column and output width of the two strips of the pair), xs[2] (column of lane 0), y0, y1 (output
 * rows [y0, y1)), xlo, xhi, ylo, yhi (the clamp rectangle) -- in the virtual Z-order plane of the row index (x = even bits,
 * y = odd bits of the row number); *n_pairs the number of pairs the plan holds (also when capacity is smaller). */
int dsph_plan_strip_pairs(const dsph_plan* plan, int32_t K, int32_t* out, int64_t capacity, int64_t* n_pairs);
/* (With the quad strips -- DSPH_OPT_STRIP_FORM 0, K = 5 -- a record is ONE 64-column strip, uncut along y: x0[0], w[0], xs[0] and
 * w[1] = 0.  The kernel cuts the rows at run time: the rows of all strips laid end to end form one tape per map, cut into
 * `pieces` equal pieces, `wg_per_piece` workgroups per piece each taking every wg_per_piece-th map of the batch; a strip is cut
 * wherever a piece ends.  This call reports that split for a batch of N maps -- for tests and tools that want to look at the seams.) */
int dsph_plan_strip_split(const dsph_plan* plan, int64_t N, int32_t* grid, int32_t* pieces, int32_t* wg_per_piece, int64_t* tape_rows);
/* (Round 6: the rectangles of the quad strips are found on the LOGICAL tile grid -- tiles on a border between two base pixels, or
 * between two superpixels of a compacted partial-sky map, belong to them wherever the neighbouring tiles continue the pixel grid by
 * a pure translation -- and a strip's record is in the coordinates of its rectangle's plane: the rectangle's first pixel is
 * (16, 16), one ring of tiles around it is addressable.  Which ROW of the map a pixel of that plane is, the strip's table says:
 * this call looks up n pixels xy[2 i], xy[2 i + 1] of record `strip` (clamped to the rectangle and its halo as the kernel clamps
 * them; K = 5: y to [ylo - 1, yhi + 6] -- that kernel reads a row as it comes, a run of steps touches these rows of the ring
 * tiles, past the halo for nothing) into rows[i].  For the strip pairs (DSPH_OPT_STRIP_FORM 1) the plane is the virtual Z-order plane of the row index itself.) */
int dsph_plan_strip_rows(const dsph_plan* plan, int32_t K, int64_t strip, int64_t n, const int32_t* xy, int64_t* rows);

/* Bytes of scratch dsph_cheb_forward needs for this call shape (0 is possible). */
size_t dsph_workspace_bytes(const dsph_plan* plan, int64_t N, int32_t Fin, int32_t Fout,
                            int32_t K, int32_t precision, int32_t algo);

/* The whole forward:  y[n,m,o] = act( sum_f sum_k (T_k(L~) x[n,:,f])[m] * w[f*K + k, o] + bias[o] )
 * Replaces: Chebyshev.call (gnn_layers.py:131-150, 155-159) including the K-1 calls of
 * utils.split_sparse_dense_matmul (utils.py:49-78) and tf.matmul (gnn_layers.py:149).
 *   x     device (N, n_cols, Fin) fp32, channels fastest, NEST pixel order as given by the caller
 *   w     device [Fin*K, Fout] fp32, row index f*K + k  (the reference's `kernel` variable)
 *   bias  device [Fout] fp32 or NULL                    (the reference's `bias`, shape [1,1,Fout])
 *   y     device (N, out_rows, Fout) fp32
 * Batch normalisation (gnn_layers.py:152-153) sits between the contraction and the bias in the
 * reference.  With the moving statistics (inference; center=False, scale=False) it is a per-channel scale and shift,
 * s[o] = 1 / sqrt(var[o] + eps): the caller passes w[:, o] * s[o] as `w` and bias[o] - mean[o] * s[o] as `bias` and gets
 * act(BN(conv) + bias) from this one call (deepsphere/gnn_layers.py does, keyed on the versions of the statistics); with
 * batch statistics (training) a layer calls this with bias=NULL, act=NONE and finishes in the host framework.
 * Asynchronous on `hip_stream` (a hipStream_t; NULL = the default stream).  (Part of a large fused forward may run on a
 * stream the plan owns, forked from and joined back into `hip_stream` inside the call: invisible to the caller, capturable.) */
int dsph_cheb_forward(const dsph_plan* plan, const float* x, const float* w, const float* bias,
                      float* y, int64_t N, int32_t Fin, int32_t Fout, int32_t K, int32_t act,
                      int32_t precision, int32_t algo, void* workspace, size_t workspace_bytes,
                      void* hip_stream);

/* The same forward for either polynomial basis (dsph_cheb_forward is this with
 * DSPH_BASIS_CHEBYSHEV).  With DSPH_BASIS_MONOMIAL it replaces Monomial.call
 * (gnn_layers.py:262-309): same layouts, same weight row order f*K + k. */
int dsph_poly_forward(const dsph_plan* plan, const float* x, const float* w, const float* bias,
                      float* y, int64_t N, int32_t Fin, int32_t Fout, int32_t K, int32_t basis,
                      int32_t act, int32_t precision, int32_t algo, void* workspace,
                      size_t workspace_bytes, void* hip_stream);

/* The same forward restricted to a subset of the output tiles (fused kernel only; DSPH_E_UNSUPPORTED otherwise):
 * DSPH_PART_INTERIOR = the 256-row tiles whose whole (K-1)-hop region lies inside the plan's output rows, i.e. on
 * a sharded plan the tiles that read no halo row of another rank; DSPH_PART_BOUNDARY = the others.  Launching
 * INTERIOR while the halo exchange is in flight and BOUNDARY after it hides the exchange behind the interior
 * (deepsphere/sharding.py).  INTERIOR + BOUNDARY write exactly the rows DSPH_PART_ALL writes, with the same bits.
 * Each part finalises exactly the rows of its own tiles (with an activation other than NONE / RELU: the kernels write the
 * pre-activation and one elementwise pass over those tiles' rows follows inside the same call), so the two parts may be
 * issued in either order, alone, or repeated. */
#define DSPH_PART_ALL 0
#define DSPH_PART_INTERIOR 1
#define DSPH_PART_BOUNDARY 2
int dsph_poly_forward_part(const dsph_plan* plan, const float* x, const float* w, const float* bias,
                           float* y, int64_t N, int32_t Fin, int32_t Fout, int32_t K, int32_t basis,
                           int32_t act, int32_t precision, int32_t algo, int32_t part, void* workspace,
                           size_t workspace_bytes, void* hip_stream);

/* The same with flags:
 *   DSPH_FWD_KEEP_WEIGHTS  `workspace` still holds the weight images that the previous call on this workspace packed from the
 *                          same w (values, not only pointer), Fin, Fout, K, basis, precision and batch class (N == 1 or N > 1:
 *                          layers with at most four input channels and 16 output columns pack four maps into one item of the
 *                          tile kernels when there is more than one, with a block-diagonal image): the fused kernels use them as
 *                          they are and the call launches no weight-preparation kernel (three small launches per forward
 *                          otherwise; on a small map they are a third of the forward).  The caller vouches for it -- a layer
 *                          in inference does, by the version counter of its kernel tensor (deepsphere/gnn_layers.py).  The
 *                          unfused path packs nothing and ignores the flag. */
#define DSPH_FWD_KEEP_WEIGHTS 1
int dsph_poly_forward_ex(const dsph_plan* plan, const float* x, const float* w, const float* bias,
                         float* y, int64_t N, int32_t Fin, int32_t Fout, int32_t K, int32_t basis,
                         int32_t act, int32_t precision, int32_t algo, int32_t part, int32_t flags, void* workspace,
                         size_t workspace_bytes, void* hip_stream);

/* The forward followed by HealpyPool(p = 1) (reference healpy_layers.py:20-63: MaxPool1D / AveragePooling1D with pool size 4 over
 * the NEST-ordered pixels, after gnn_layers.py:106-161) in one call: y_pooled (N, n_rows / 4, Fout) = pool(act(conv(x) + bias)).
 * The input-side strip kernels and both tile kernels reduce the four children in their store step and write the pooled map
 * only -- the full-resolution output of a 1 -> 16 layer is 16 times its input and, with the pooling's read of it, most of that
 * layer's time; `y_scratch` is not used any more (every kernel pools in
 * its store) and may be NULL.  Same values as the two calls (bit for bit for the maximum; the mean adds the
 * children in the same order).  pool_type: DSPH_POOL_MAX | DSPH_POOL_AVG.  Availability: dsph_plan_pool_fusable (whole
 * unsharded maps of whole tiles, a width that is a multiple of four, activation none or ReLU, not the 64 -> 64 shape where the
 * Clenshaw strip kernel takes tiles); DSPH_E_UNSUPPORTED otherwise -- run
 * dsph_poly_forward and dsph_healpix_pool then.  Workspace and flags as dsph_poly_forward_ex. */
int dsph_plan_pool_fusable(const dsph_plan* plan, int64_t N, int32_t Fin, int32_t Fout, int32_t K, int32_t act);
int dsph_poly_forward_pool(const dsph_plan* plan, const float* x, const float* w, const float* bias, float* y_scratch,
                           float* y_pooled, int64_t N, int32_t Fin, int32_t Fout, int32_t K, int32_t basis, int32_t act,
                           int32_t precision, int32_t pool_type, int32_t flags, void* workspace, size_t workspace_bytes,
                           void* hip_stream);

/* One recurrence step on (N, n_cols, F) planes:  out = alpha * (L~ @ in) - beta * prev
 * for rows [0, rows) of every map (rows <= n_rows; rows <= 0 means n_rows); prev may be NULL
 * when beta == 0.  Replaces one utils.split_sparse_dense_matmul call plus the `2*... - x0`
 * temporaries (gnn_layers.py:138,141).  Exposed for the one-ring-halo-per-step sharded mode,
 * where the host exchanges boundary rows between steps, and for tests. */
int dsph_cheb_step(const dsph_plan* plan, const float* in, const float* prev, float* out,
                   int64_t N, int32_t F, float alpha, float beta, int64_t rows, void* hip_stream);
/* Note on plans with halo columns (n_cols > n_rows): a step writes rows [0, rows) only, so a SECOND step would
 * gather halo entries nobody produced.  The multi-step entry points (dsph_cheb_forward / dsph_poly_forward*,
 * dsph_cheb_planes, dsph_cheb_backward_weights) therefore return DSPH_E_BADARG for K > 2 on such a plan unless a
 * shrinking schedule was installed with dsph_plan_set_levels; drive dsph_cheb_step yourself, exchanging the halo
 * between steps, for the one-ring-per-step mode. */

/* The dense contraction alone on K planes of shape (N, plane_rows, Fin), plane k at
 * planes[k] (device pointers in a HOST array of K entries):
 *   y[n,m,o] = act( sum_k sum_f planes[k][n,m,f] * w[f*K + k, o] + bias[o] ),  m < rows.
 * Replaces tf.stack/reshape/transpose/matmul (gnn_layers.py:144-150). */
int dsph_cheb_contract(const float* const* planes, int64_t plane_rows, const float* w,
                       const float* bias, float* y, int64_t N, int64_t rows, int32_t Fin,
                       int32_t Fout, int32_t K, int32_t act, int32_t precision, int device,
                       void* hip_stream);

/* The planes T_1 x .. T_{K-1} x alone (T_0 x = x), without the contraction: `planes` receives K-1
 * arrays of x's shape (N, n_cols, Fin) back to back, valid on the plan's output rows.  This is the left
 * operand of the weight gradient, rebuilt in the backward pass: the same recurrence as
 * gnn_layers.py:134-143, run by the fused tile kernel (no MFMA, the tile rows of every plane are
 * stored from LDS) when the plan/shape allows, else by K-1 dsph_cheb_step launches.
 * `algo`: DSPH_ALGO_AUTO | UNFUSED | FUSED as in dsph_cheb_forward. */
int dsph_cheb_planes(const dsph_plan* plan, const float* x, float* planes, int64_t N, int32_t Fin,
                     int32_t K, int32_t basis, int32_t algo, void* hip_stream);

/* Weight gradient of the layer (training):
 *     dw[f*K + k, o] = sum_{n, m < out_rows} (T_k(L~) x)[n,m,f] * dy[n,m,o]
 * from the layer input x (N, n_cols, Fin) and the upstream gradient dy (N, out_rows, Fout).  The reference
 * has no counterpart of its own: TensorFlow differentiates the op sequence of gnn_layers.py:131-150.
 * When the fused tile kernel applies (dsph_plan_fused_ok) the planes never leave the LDS: each tile's T_k
 * is contracted over its pixels against dy -- `precision` DSPH_PREC_FP32: exact-fp32 MFMAs; DSPH_PREC_BF16X3: both
 * operands split hi + lo, three bf16 MFMAs per product term, fp32 accumulate (fused path only) -- and the
 * per-workgroup partial sums are reduced in a fixed order (deterministic).  Otherwise: dsph_cheb_planes into `workspace`, then
 * dsph_cheb_wgrad.  `algo` as in dsph_cheb_forward.
 * K = 5, Fin = 64, Fout a multiple of 64, DSPH_PREC_BF16X3, on a whole-graph plan whose forward of that shape runs on the quad
 * strips and whose matrix equals its transpose (checked entry for entry, to fp32 rounding): the strips' pixels go through the
 * quad-strip weight-gradient kernel (csrc/cheb_qwgrad_kernel.h: the recurrence to order two on x AND on dy, orders 3 and 4 from
 * the product rule of the polynomials), the other tiles through the tile kernel; same result to the arithmetic's rounding
 * (measured 4 - 7e-6 of max |dw| against float64), 2.2 x faster at BASELINE configs[2].  A non-symmetric matrix, another shape
 * or DSPH_PREC_FP32 keep every tile on the tile kernel.
 * In the bf16 arithmetic every second workgroup of both kernels contracts against -dy and its partial sum is subtracted: the
 * matrix pipe's fp32 accumulation of long sums is biased towards minus infinity (1 - 2e-5 of max |dw| at configs[2]) and the
 * mirror cancels it. */
size_t dsph_backward_weights_workspace_bytes(const dsph_plan* plan, int64_t N, int32_t Fin, int32_t Fout,
                                             int32_t K, int32_t algo);
int dsph_cheb_backward_weights(const dsph_plan* plan, const float* x, const float* dy, float* dw, int64_t N,
                               int32_t Fin, int32_t Fout, int32_t K, int32_t basis, int32_t precision, int32_t algo,
                               void* workspace, size_t workspace_bytes, void* hip_stream);

/* Weight gradient (training):  dw[f*K + k, o] = sum_{n, m < rows} planes[k][n,m,f] * dy[n,m,o]
 * for K planes T_k x of shape (N, plane_rows, Fin) (HOST array of K device pointers, e.g. x and the
 * outputs of dsph_cheb_step) and an upstream gradient dy (N, rows, Fout).  The reference has no
 * counterpart of its own: TensorFlow differentiates tf.matmul (gnn_layers.py:149).  Deterministic
 * (fixed-order reduction of per-workgroup partial sums held in `workspace`). */
size_t dsph_wgrad_workspace_bytes(int64_t N, int64_t rows, int32_t Fin, int32_t Fout, int32_t K);
int dsph_cheb_wgrad(const float* const* planes, int64_t plane_rows, const float* dy, float* dw,
                    int64_t N, int64_t rows, int32_t Fin, int32_t Fout, int32_t K, void* workspace,
                    size_t workspace_bytes, int device, void* hip_stream);

/* Gather / scatter of boundary rows for the halo exchange of the sharded path:
 *   pack:   buf[n, i, :] = src[n, idx[i], :]       src (N, src_rows, F), buf (N, n_idx, F)
 *   unpack: dst[n, idx[i], :] = buf[n, i, :]
 * idx is a DEVICE int32 array. */
int dsph_rows_pack(const float* src, int64_t src_rows, const int32_t* idx, int64_t n_idx,
                   float* buf, int64_t N, int32_t F, int device, void* hip_stream);
int dsph_rows_unpack(float* dst, int64_t dst_rows, const int32_t* idx, int64_t n_idx,
                     const float* buf, int64_t N, int32_t F, int device, void* hip_stream);

/* Skip connection of a residual block (reference gnn_layers.GCNN_ResidualLayer.call, gnn_layers.py:407-413: tf.add and
 * the activation, three elementwise ops over the whole map) in ONE pass, in place over n contiguous floats:
 *   act_before == 0:  y = act(y + alpha * skip)          act_before != 0:  y = act(y) + alpha * skip
 * act is a DSPH_ACT_* code (DSPH_ACT_NONE with alpha = 1 is the reference's activation=None case, x + input). */
int dsph_residual_epilogue(float* y, const float* skip, int64_t n, float alpha, int32_t act, int32_t act_before,
                           int device, void* hip_stream);

/* Pooling over the 4^p NEST children of a HEALPix pixel, the step between two convolutions of every reference model.
 * Replaces: the Keras MaxPool1D / AveragePooling1D inside healpy_layers.HealpyPool.call (healpy_layers.py:48-63,77-85;
 * pool_size = strides = 4^p, padding "valid", channels last).
 *   x  device (N, rows_out * group, F) fp32      y  device (N, rows_out, F) fp32      group = 4^p
 *   y[n, m, f] = max | mean over i < group of x[n, group * m + i, f]
 * The backward (the reference gets it from TensorFlow's autodiff): dx[n, group * m + i, f] = dy[n, m, f] / group for the mean;
 * for the maximum dy[n, m, f] at the first child that holds it and 0 at the others (x: the forward input; may be NULL for the mean). */
#define DSPH_POOL_MAX 0
#define DSPH_POOL_AVG 1
int dsph_healpix_pool(const float* x, float* y, int64_t N, int64_t rows_out, int32_t F, int32_t group, int32_t type, int device,
                      void* hip_stream);
int dsph_healpix_pool_backward(const float* x, const float* dy, float* dx, int64_t N, int64_t rows_out, int32_t F, int32_t group,
                               int32_t type, int device, void* hip_stream);

const char* dsph_last_error(void);
int dsph_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* DSPHERE_H */
