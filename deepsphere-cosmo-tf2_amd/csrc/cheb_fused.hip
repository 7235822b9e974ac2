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
//                                 order) first, then ring 1, ... ring D, each ring ordered so that
//                                 (local index & 3) == (row id & 3) where the ring's mix allows
//   ring_end[t][r]                number of region entries within r hops (r = 0..D)
//   lcols/lvals                   tile-local ELL of the rows within D-1 hops, column = index into
//                                 the region list (uint16), stored [slot][row] per tile so that
//                                 lane i reads row i coalesced
// LDS: two planes [Rmax][16] fp32 (16-byte slots XOR-swizzled so that 16 lanes reading the same
// slot of 16 different rows hit 16 different bank groups) + the weight fragments of all
// (order, slice) pairs in MFMA operand order.
// Registers: each lane owns one (or two) region rows for the whole tile: their ELL values and
// pre-swizzled LDS addresses stay in VGPRs across all slices and all maps of the batch.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "cheb_fused_kernel.h"
#include "cheb_istrip_kernel.h"
#include "cheb_qstrip8_kernel.h"
#include "cheb_qstrip_kernel.h"
#include "cheb_strip_kernel.h"
#include "cheb_struct_kernel.h"

namespace dsph {


struct FusedTiles {
  int D = 0;
  int width = 0;     // ELL width of the tile-local table (template width, >= plan width)
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
  // tiles whose whole region lies in the plan's output rows ("interior": on a sharded plan they need no halo row
  // of another rank) and the rest ("boundary"); positions [0, n_interior) and [n_interior, ntiles) of d_part
  int32_t* d_part = nullptr;
  int n_interior = 0;
  int n_part = 0;      // entries of d_part: all tiles of a full table, the class-G tiles otherwise
  // class-R tiles (cheb_struct_kernel.h), interior ones first; empty in a full table
  int32_t* d_rlist = nullptr;
  int n_r = 0, n_r_interior = 0;
  // class-T tiles (structured kernel with per-tile tables, embed_tile below), interior ones first
  int32_t* d_tlist = nullptr;
  int32_t* d_tabrow = nullptr;  // [n_t][ST_CELLS] row of every plane cell
  float* d_tabvals = nullptr;   // [n_t][ST_CELLS][ST_TABV] diagonal + eight directions of L~ per cell, in the tile's frame
  int n_t = 0, n_t_interior = 0;
  // strip kernel (cheb_strip_kernel.h): rectangles of interior class-R tiles cut into strip pairs, and the class-R tiles
  // they leave over (interior ones first).  Built next to d_rlist; which of the two sets a forward uses depends on its shape.
  StripPair* d_pairs = nullptr;
  int n_pairs = 0;
  int64_t n_strip_tiles = 0;
  std::vector<int32_t> strip_steps;        // per pair, in list order: rows + run-in = strip steps of one map
  std::vector<StripPair> h_pairs;          // host copy of d_pairs (dsph_plan_strip_pairs: what the seam tests read)
  mutable std::map<int64_t, int64_t> strip_span;  // batch N -> steps of the busiest workgroup (strip_makespan; under FusedPlan::mu)
  bool strip_forced = false;               // DSPH_OPT_STRIPS = 1 when the tables were built: the cost gate is off
  // quad-strip kernel (cheb_qstrip_kernel.h): the same rectangles, merged side by side and cut into 64-column strips
  QStrip* d_qstrips = nullptr;             // uncut along y: the kernel cuts the tape of their rows evenly over its workgroups
  int32_t* d_qprefix = nullptr;            // [n_qstrips + 1] rows before each strip
  int n_qstrips = 0;
  int64_t qtape_rows = 0;
  std::vector<QStrip> h_qstrips;
  // ... addressed through tables of tile bases (round 6, build_qtstrips): the rectangles of the quad strips are found on the
  // LOGICAL tile grid -- class-R tiles and the class-T tiles whose eight neighbour tiles continue the pixel grid by a pure
  // translation (base-pixel borders between an equatorial and a polar face, superpixel borders of a compacted partial-sky
  // map) -- and a strip looks its rows up as tab[tile row][tile column] + morton(x & 15, y & 15).
  int32_t* d_qtab = nullptr;               // tile-base tables of all rectangles, back to back (row numbers)
  std::vector<int32_t> h_qtab;
  int64_t n_qstrip_tiles = 0;              // tiles the quad strips take
  // K = 8 quad strips (cheb_qstrip8_kernel.h; the tables of depth 7 only): rectangles of the tiles whose 7-ring region is a regular
  // square of the Morton plane, and the tiles they leave to the breadth-first tile kernel (interior first, the order of d_part)
  QStrip* d_q8strips = nullptr;
  int32_t* d_q8prefix = nullptr;
  int32_t* d_q8tab = nullptr;
  int32_t* d_q8rest = nullptr;
  int n_q8strips = 0, n_q8rest = 0, n_q8rest_interior = 0;  // (the rest list: interior tiles first, like d_part)
  int64_t q8tape_rows = 0, n_q8_tiles = 0;
  std::vector<QStrip> h_q8strips;
  std::vector<int32_t> h_q8tab;
  // what the quad strips leave to the structured kernel: class-R tiles (interior first) and class-T tiles with their tables
  int32_t* d_qrrest = nullptr;
  int n_qrrest = 0, n_qrrest_interior = 0;
  int32_t* d_qtlist = nullptr;
  int32_t* d_qtabrow = nullptr;
  float* d_qtabvals = nullptr;
  int n_qt = 0, n_qt_interior = 0;
  // input-side strip kernel (cheb_istrip_kernel.h): the same rectangles, uncut along y (the kernel cuts every strip into the
  // number of row segments that istrip_segments picks for the batch)
  StripPair* d_ipairs = nullptr;
  int n_ipairs = 0;
  std::vector<int32_t> ipair_h;             // rows of every pair
  std::vector<unsigned char> ipair_second;  // whether its second strip exists
  mutable std::map<int64_t, int> iseg;      // 2 * batch N + narrow -> row segments per strip (under FusedPlan::mu)
  // every tile of the plan, the interior ones first (d_all[0 .. n_all_interior)): what a two-part launch with a deferred
  // activation finishes per part (launch_struct_act_tiles)
  int32_t* d_all = nullptr;
  int n_all = 0, n_all_interior = 0;
  int32_t* d_rrest = nullptr;
  int n_rrest = 0, n_rrest_interior = 0;
  // every tile the strips leave over, whatever its class (class R rest, T, G): what the BFS-tile kernel's weight-gradient mode
  // runs next to the quad-strip weight gradient (cheb_qwgrad.hip)
  int32_t* d_nonq = nullptr;
  int n_nonq = 0;
};

struct FusedPlan {
  std::vector<int32_t> h_cols;  // host copy of the ELL (needed to build tiles for a new K)
  std::vector<float> h_vals;
  std::mutex mu;
  std::map<int, FusedTiles> by_depth;  // key: 2 * depth + (1: BFS tables of every tile, 0: of the class-G tiles only)
  int num_cu = 256;
  // direction-ordered copy of L~ and the per-row regularity flags (cheb_struct.hip), built on first use
  float* d_gvals8 = nullptr;
  float* d_gdiag = nullptr;
  unsigned char* d_rowflag = nullptr;
  bool rows_tried = false;
  bool host_released = false;  // DSPH_PREPARE_RELEASE_HOST: no tables for further depths
  int symmetric = -1;          // L~ == L~^T entry for entry (fused_symmetric; -1: not looked at yet)
  bool wide = false;           // ELL wider than the fused kernels' templates: only the depth-1 tables of the tiled step exist
  // The BFS-tile launch of a forward writes tiles of y that the structured launches do not touch: it runs on this side stream,
  // forked from and joined back into the caller's stream by the two events (a few dozen face-corner tiles would otherwise
  // hold the whole device for the latency of one tile: 14 of the 44 us of BASELINE configs[0], 57 of 544 us of configs[1]).
  // Created by dsph_plan_prepare on plans whose tables call for a fork (side_stream_ready) -- a prepared forward allocates
  // nothing; fork_mu keeps two host threads that share a plan from interleaving their record / wait pairs.  Capturable: the side stream joins the capture through the fork event and leaves it at the join.
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool side_failed = false;
  std::mutex fork_mu;
  // DSPH_FWD_KEEP_WEIGHTS: the caller vouches for the weight VALUES; WHICH images a workspace block holds is the library's
  // own business -- it depends on per-call choices (strips or tiles by the batch, maps packed or not) the caller cannot see.
  // Per workspace block: the shape key the images were packed for and the set of images packed since (IMG_* bits).  A kept
  // call packs what the set lacks; a call that does not keep starts an empty set.
  struct Images { uint64_t key = 0; uint32_t mask = 0; };
  std::mutex img_mu;
  std::unordered_map<const void*, Images> images;
};

void fused_images_begin(const dsph_plan* plan, const void* ws, uint64_t key, bool keep) {
  FusedPlan* fp = plan->fused;
  if (!fp || !ws) return;
  std::lock_guard<std::mutex> lock(fp->img_mu);
  if (fp->images.size() > 16384 && fp->images.find(ws) == fp->images.end()) fp->images.clear();  // (forgetting only costs a re-pack)
  FusedPlan::Images& im = fp->images[ws];
  if (!keep || im.key != key) { im.key = key; im.mask = 0; }
}
// true: image `bit` of this block has to be packed now (and counts as packed from here on)
bool fused_images_claim(const dsph_plan* plan, const void* ws, uint32_t bit) {
  FusedPlan* fp = plan->fused;
  if (!fp || !ws) return true;
  std::lock_guard<std::mutex> lock(fp->img_mu);
  auto it = fp->images.find(ws);
  if (it == fp->images.end()) return true;
  if (it->second.mask & bit) return false;
  it->second.mask |= bit;
  return true;
}
void fused_images_forget(const dsph_plan* plan, const void* ws) {
  FusedPlan* fp = plan->fused;
  if (!fp || !ws) return;
  std::lock_guard<std::mutex> lock(fp->img_mu);
  fp->images.erase(ws);
}
uint64_t fused_images_key(int32_t Fin, int32_t Fin_w, int32_t Fout, int32_t K, int32_t ld, int32_t precision, bool cheb, bool many, bool pack) {
  // (values beyond the fields' widths alias at worst to "another key": a re-pack)
  return ((uint64_t)(Fin & 0xfff)) | ((uint64_t)(Fin_w & 0xfff) << 12) | ((uint64_t)(Fout & 0xff) << 24) | ((uint64_t)(K & 0x3f) << 32) |
         ((uint64_t)(ld & 0xffff) << 38) | ((uint64_t)(precision & 3) << 54) | ((uint64_t)cheb << 56) | ((uint64_t)many << 57) |
         ((uint64_t)pack << 58) | (1ull << 63);
}

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
  if (ft.d_part) (void)hipFree(ft.d_part);
  if (ft.d_rlist) (void)hipFree(ft.d_rlist);
  if (ft.d_tlist) (void)hipFree(ft.d_tlist);
  if (ft.d_tabrow) (void)hipFree(ft.d_tabrow);
  if (ft.d_tabvals) (void)hipFree(ft.d_tabvals);
  if (ft.d_pairs) (void)hipFree(ft.d_pairs);
  if (ft.d_qstrips) (void)hipFree(ft.d_qstrips);
  if (ft.d_qprefix) (void)hipFree(ft.d_qprefix);
  if (ft.d_qtab) (void)hipFree(ft.d_qtab);
  if (ft.d_q8strips) (void)hipFree(ft.d_q8strips);
  if (ft.d_q8prefix) (void)hipFree(ft.d_q8prefix);
  if (ft.d_q8tab) (void)hipFree(ft.d_q8tab);
  if (ft.d_q8rest) (void)hipFree(ft.d_q8rest);
  if (ft.d_qrrest) (void)hipFree(ft.d_qrrest);
  if (ft.d_qtlist) (void)hipFree(ft.d_qtlist);
  if (ft.d_qtabrow) (void)hipFree(ft.d_qtabrow);
  if (ft.d_qtabvals) (void)hipFree(ft.d_qtabvals);
  if (ft.d_rrest) (void)hipFree(ft.d_rrest);
  if (ft.d_nonq) (void)hipFree(ft.d_nonq);
  if (ft.d_all) (void)hipFree(ft.d_all);
  if (ft.d_ipairs) (void)hipFree(ft.d_ipairs);
  ft = FusedTiles();
}

FusedPlan* fused_plan_build(const dsph_plan* plan, const int32_t* h_cols, const float* h_vals) {
  if (template_width(plan->width) == 0 && tstep_width(plan->width) == 0) return nullptr;
  FusedPlan* fp = new FusedPlan();
  fp->wide = template_width(plan->width) == 0;
  const size_t nnz = (size_t)plan->n_rows * plan->width;
  fp->h_cols.assign(h_cols, h_cols + nnz);
  fp->h_vals.assign(h_vals, h_vals + nnz);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, plan->device) == hipSuccess && prop.multiProcessorCount > 0)
    fp->num_cu = prop.multiProcessorCount;
  return fp;
}

// The side stream and its two events exist only on plans that fork (a network holds dozens of plans, most of which
// never do): created by dsph_plan_prepare when the prepared tables call for it, or by the first forward that wants to fork --
// unless that forward is being captured into a graph, in which case it runs on the one stream.  Under fork_mu.
static bool side_stream_ready(const dsph_plan* plan, FusedPlan* fp, hipStream_t caller, bool may_create) {
  if (fp->side) return true;
  if (fp->side_failed || !may_create) return false;
  if (caller != nullptr) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(caller, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return false;
  }
  DeviceGuard guard(plan->device);
  hipStream_t st = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (!guard.ok || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { fp->side_failed = true; return false; }
  if (hipEventCreateWithFlags(&e0, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&e1, hipEventDisableTiming) != hipSuccess) {
    if (e0) (void)hipEventDestroy(e0);
    (void)hipStreamDestroy(st);
    fp->side_failed = true;
    return false;
  }
  fp->ev_fork = e0;
  fp->ev_join = e1;
  fp->side = st;
  return true;
}

void fused_plan_destroy(FusedPlan* fp) {
  if (!fp) return;
  for (auto& kv : fp->by_depth) free_tiles(kv.second);
  if (fp->d_gvals8) (void)hipFree(fp->d_gvals8);
  if (fp->d_gdiag) (void)hipFree(fp->d_gdiag);
  if (fp->d_rowflag) (void)hipFree(fp->d_rowflag);
  if (fp->side) {
    (void)hipStreamSynchronize(fp->side);
    (void)hipEventDestroy(fp->ev_fork);
    (void)hipEventDestroy(fp->ev_join);
    (void)hipStreamDestroy(fp->side);
  }
  delete fp;
}

bool fused_host_released(FusedPlan* fp) {
  std::lock_guard<std::mutex> lock(fp->mu);
  return fp->host_released;
}

// The tile tables depend on which rows are outputs (dsph_plan_set_levels): drop the cached ones.
void fused_plan_invalidate(FusedPlan* fp) {
  if (!fp) return;
  std::lock_guard<std::mutex> lock(fp->mu);
  for (auto& kv : fp->by_depth) free_tiles(kv.second);
  fp->by_depth.clear();
}

// Class-T tiles: a tile whose own 256 rows are a 16x16 Morton square and whose D-ring region can be laid out as a
// (16+2D)^2 square of a 9-point stencil although the ROW NUMBERS of the halo are not a Morton continuation (the next base
// pixel of the sphere, possibly rotated; the halo rows of a sharded plan, numbered by hop distance).  The layout is
// found from the graph alone, ring by ring: a new cell is the one row that all of its already-placed neighbours have in
// common and that is not placed yet.  It is then VERIFIED, not trusted: every row that a step evaluates (rings 0..D-1)
// must have each non-zero of L~ on itself or on one of its eight neighbouring cells; only then do the tables exist.
// Anything else (the eight 7-neighbour corners of the sphere, mask edges, ragged tiles, k-NN graphs) stays class G.
// row: [ST_CELLS] row of every plane cell (plane-cell order, pads and cells outside the region: the tile's first row);
// val: [ST_CELLS][ST_TABV] diagonal, then the directions in the order of kDirX / kDirY.
static bool embed_tile(const dsph_plan* plan, const int32_t* cols, const float* vals, int W, int t, int D, int64_t out_rows,
                       int32_t* row, float* val, bool* interior, std::vector<int32_t>& key, std::vector<int32_t>& slot) {
  const int64_t r0 = (int64_t)t * FUSED_P;
  if (r0 + FUSED_P > out_rows) return false;  // ragged last tile
  constexpr int S = ST_S;
  int32_t A[S][S];
  for (int y = 0; y < S; ++y)
    for (int x = 0; x < S; ++x) A[x][y] = -1;
  // open-addressing map row -> cell (x + S * y); 2048 slots for at most 576 entries
  constexpr int HN = 2048;
  key.assign(HN, -1);
  slot.assign(HN, 0);
  auto hput = [&](int32_t r, int cell) {
    unsigned h = ((unsigned)r * 2654435761u) >> 21;
    while (key[h] != -1) h = (h + 1) & (HN - 1);
    key[h] = r;
    slot[h] = cell;
  };
  auto hget = [&](int32_t r) -> int {
    unsigned h = ((unsigned)r * 2654435761u) >> 21;
    while (key[h] != -1) {
      if (key[h] == r) return slot[h];
      h = (h + 1) & (HN - 1);
    }
    return -1;
  };
  for (unsigned y = 0; y < 16; ++y)
    for (unsigned x = 0; x < 16; ++x) {
      const int32_t r = (int32_t)(r0 + st_morton(x, y));
      A[ST_DMAX + x][ST_DMAX + y] = r;
      hput(r, (ST_DMAX + x) + S * (ST_DMAX + y));
    }
  // the one unplaced row adjacent to every placed neighbour of cell (x, y); -1 if there is none or more than one
  auto propose = [&](int x, int y) -> int32_t {
    int32_t cand[16];
    int nc = -1;  // -1: no anchor seen yet
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        const int ax = x + dx, ay = y + dy;
        if ((dx == 0 && dy == 0) || ax < 0 || ay < 0 || ax >= S || ay >= S || A[ax][ay] < 0) continue;
        const int32_t ar = A[ax][ay];
        if (ar >= plan->n_rows) continue;  // an input-only row has no list of neighbours: not an anchor
        const int32_t* c = cols + (size_t)ar * W;
        const float* v = vals + (size_t)ar * W;
        if (nc < 0) {
          nc = 0;
          for (int j = 0; j < W; ++j)
            if (v[j] != 0.f && c[j] != ar && hget(c[j]) < 0 && nc < 16) cand[nc++] = c[j];
        } else {
          int m = 0;
          for (int i = 0; i < nc; ++i) {
            bool in = false;
            for (int j = 0; j < W && !in; ++j) in = v[j] != 0.f && c[j] == cand[i];
            if (in) cand[m++] = cand[i];
          }
          nc = m;
        }
      }
    return nc == 1 ? cand[0] : (nc > 1 ? -1 : -2);  // -1: ambiguous; -2: no row belongs here
  };
  // A cell no row belongs to is a HOLE (round 6: the edge of a survey mask -- the pixel beyond it does not exist): it stays
  // negative in A (never an anchor), its table entry is the tile's first row with nine zeros -- what a step computes there is
  // finite and no row of the map has a non-zero towards it.  Nothing is taken on trust: the verification below still demands
  // that every non-zero of every evaluated row lands on a placed cell at one of the eight offsets, so a row left out by
  // mistake (or a vertex where three base pixels meet, whose far seam is not a 3 x 3 neighbourhood) fails the tile as before.
  auto place = [&](int x, int y) -> bool {
    const int32_t r = propose(x, y);
    if (r == -1) return false;
    if (r == -2) { A[x][y] = -2; return true; }
    A[x][y] = r;
    hput(r, x + S * y);
    return true;
  };
  for (int rho = 1; rho <= D; ++rho) {
    const int lo = ST_DMAX - rho, hi = ST_DMAX + ST_TILE - 1 + rho;
    // the four sides without their corners, from the middle outwards (so that each new cell has placed neighbours on
    // the inner ring and, after the first, on its own side), then the corners
    const int mid = (lo + hi) / 2;
    for (int k = 0; k <= hi - lo; ++k) {
      const int off = (k + 1) / 2 * ((k & 1) ? 1 : -1);  // 0, +1, -1, +2, -2, ...
      const int u = mid + off;
      if (u <= lo || u >= hi) continue;
      if (!place(u, hi) || !place(u, lo) || !place(lo, u) || !place(hi, u)) return false;
    }
    if (!place(lo, lo) || !place(hi, lo) || !place(lo, hi) || !place(hi, hi)) return false;
  }
  // tables + verification
  const int blo = ST_DMAX - D, bhi = ST_DMAX + ST_TILE - 1 + D;
  for (int p = 0; p < ST_CELLS; ++p) {
    row[p] = (int32_t)r0;
    for (int j = 0; j < ST_TABV; ++j) val[(size_t)p * ST_TABV + j] = 0.f;
  }
  *interior = true;
  for (int y = blo; y <= bhi; ++y)
    for (int x = blo; x <= bhi; ++x) {
      const int32_t r = A[x][y];
      const unsigned p = st_cell_off((unsigned)x, (unsigned)y) / 64u;
      if (r < 0) continue;  // a hole: the tile's first row, all zeros (set above)
      row[p] = r;
      if (r >= out_rows) *interior = false;
      if (x == blo || x == bhi || y == blo || y == bhi) continue;  // outermost ring: input only
      if (r >= plan->n_rows) return false;  // a row that a step evaluates has no row of L~
      const int32_t* c = cols + (size_t)r * W;
      const float* v = vals + (size_t)r * W;
      float* o = val + (size_t)p * ST_TABV;
      unsigned seen = 0;
      for (int j = 0; j < W; ++j) {
        if (v[j] == 0.f) continue;
        int d = -1;
        if (c[j] == r) d = 0;
        else {
          const int cell = hget(c[j]);
          if (cell < 0) return false;
          const int dx = cell % S - x, dy = cell / S - y;
          static const int ddx[8] = {-1, -1, 0, 1, 1, 1, 0, -1}, ddy[8] = {0, 1, 1, 1, 0, -1, -1, -1};  // = kDirX / kDirY
          for (int q = 0; q < 8; ++q)
            if (ddx[q] == dx && ddy[q] == dy) d = q + 1;
          if (d < 0) return false;  // a non-zero that is not on one of the eight neighbouring cells
        }
        if (seen & (1u << d)) return false;
        seen |= 1u << d;
        o[d] = v[j];
      }
    }
  return true;
}

// Steps of the busiest workgroup when the strip kernel deals `steps.size()` pairs x N maps the way cheb_strip5_kernel does:
// items q = pair * N + map, a contiguous eighth of them per XCD, dealt to the XCD's workgroups in turn.
static int strip_grid(int num_cu, int64_t n_items) { return (int)std::max<int64_t>(8, std::min<int64_t>(num_cu, (n_items + 7) / 8 * 8)); }
static int64_t strip_makespan(const std::vector<int32_t>& steps, int64_t N, int num_cu) {
  const int64_t Q = (int64_t)steps.size() * N;
  const int G = strip_grid(num_cu, Q);
  int64_t worst = 0;
  std::vector<int64_t> load;
  for (int xcd = 0; xcd < 8; ++xcd) {
    const int nslots = (G + 7 - xcd) / 8;
    const int64_t q0 = Q * xcd / 8, q1 = Q * (xcd + 1) / 8;
    if (nslots <= 0 || q1 <= q0) continue;
    load.assign((size_t)nslots, 0);
    for (int64_t q = q0; q < q1; ++q) load[(size_t)((q - q0) % nslots)] += steps[(size_t)(q / N)];
    for (int64_t v : load) worst = std::max(worst, v);
  }
  return worst;
}

constexpr int SMALL_MAP_TILES = 512;  // structured tiles up to which a plan without strips runs them in ONE launch (class-R tiles with tables)
// Strip kernel: which class-R tiles it takes, and in what pieces.  The interior class-R tiles are covered by rectangles (in
// the virtual Morton plane of the tile indices: tile t sits at (compress(t), compress(t >> 1))) of 3 to 5 tile columns and at
// least 4 tile rows; a rectangle is cut into 32-column strips with 24 output columns each, two strips per workgroup item,
// and into row segments sized so that the items fill the CUs evenly.  The other tiles stay with the tile kernels (`rest`).
static void build_strips(const std::vector<int32_t>& r_interior, int D, int num_cu, const PlanOptions& opt,
                         std::vector<StripPair>& pairs, std::vector<int32_t>& rest, int64_t* n_taken, std::vector<int32_t>& steps,
                         std::vector<StripPair>& whole,  // whole: the same strips uncut along y (input-side strip kernel)
                         std::vector<QStrip>* qstrips = nullptr) {  // quad-strip kernel
  if (qstrips) qstrips->clear();
  steps.clear();
  pairs.clear();
  whole.clear();
  rest.clear();
  *n_taken = 0;
  struct Rect { int tx, ty, wt, ht; };
  // columns of the tile plane and their vertical runs of class-R tiles
  std::vector<uint64_t> keys(r_interior.size());  // (tx, ty)
  for (size_t i = 0; i < r_interior.size(); ++i) {
    const unsigned t = (unsigned)r_interior[i];
    keys[i] = ((uint64_t)st_compress(t) << 32) | st_compress(t >> 1);
  }
  std::sort(keys.begin(), keys.end());
  struct Run { int tx, y0, y1; bool used; };
  std::vector<Run> runs;
  for (size_t i = 0; i < keys.size();) {
    const int tx = (int)(keys[i] >> 32), y0 = (int)(keys[i] & 0xffffffffu);
    size_t j = i + 1;
    while (j < keys.size() && (int)(keys[j] >> 32) == tx && (int)(keys[j] & 0xffffffffu) == y0 + (int)(j - i)) ++j;
    runs.push_back({tx, y0, y0 + (int)(j - i), false});
    i = j;
  }
  // Groups of 3 (at the end of a stretch: 4 or 5) adjacent columns whose runs share at least 4 rows: one rectangle each, of the
  // shared rows.  (Three tile columns = 48 pixels = one pair of strips; tall rather than wide, the strips run along y.  On a
  // full base pixel every column has the same run and nothing is left over; on a mask the rows a group does not share, and
  // stretches of fewer than 3 columns, stay with the tile kernels.)
  auto overlap = [](int a0, int a1, int b0, int b1, int* o0, int* o1) { *o0 = std::max(a0, b0); *o1 = std::min(a1, b1); return *o1 - *o0; };
  const int min_rows = std::max(4, opt.strip_min_rows);  // (tuning, DSPH_OPT_STRIP_MINROWS: least height of a rectangle, in tiles)
  std::vector<Rect> take;
  for (size_t i = 0; i < runs.size(); ++i) {
    if (runs[i].used) continue;
    // how many adjacent columns continue this run with at least 4 shared rows (at most 6 looked at)
    std::vector<size_t> chain{i};
    int y0 = runs[i].y0, y1 = runs[i].y1;
    while (chain.size() < 6) {
      const int want = runs[chain.back()].tx + 1;
      size_t best = runs.size();
      int by0 = 0, by1 = 0;
      for (size_t j = chain.back() + 1; j < runs.size() && runs[j].tx <= want; ++j) {
        int o0, o1;
        if (runs[j].tx == want && !runs[j].used && overlap(y0, y1, runs[j].y0, runs[j].y1, &o0, &o1) >= min_rows &&
            (best == runs.size() || o1 - o0 > by1 - by0)) { best = j; by0 = o0; by1 = o1; }
      }
      if (best == runs.size()) break;
      if (chain.size() < 5) { y0 = by0; y1 = by1; }  // (the sixth only says "the stretch goes on")
      chain.push_back(best);
    }
    const int avail = (int)chain.size();
    if (avail < 3 || y1 - y0 < min_rows) continue;  // stays with the tile kernels
    const int w = avail >= 6 ? 3 : std::min(avail, 5);
    // the shared rows of the w columns actually taken
    y0 = runs[chain[0]].y0; y1 = runs[chain[0]].y1;
    for (int c = 1; c < w; ++c) { int o0, o1; overlap(y0, y1, runs[chain[c]].y0, runs[chain[c]].y1, &o0, &o1); y0 = o0; y1 = o1; }
    if (y1 - y0 < min_rows) continue;
    take.push_back({runs[chain[0]].tx, y0, w, y1 - y0});
    *n_taken += (int64_t)w * (y1 - y0);
    for (int c = 0; c < w; ++c) {
      Run& r = runs[chain[c]];
      // what the rectangle leaves of the run: above / below stay as (used) leftovers for the tile kernels
      for (int y = r.y0; y < r.y1; ++y)
        if (y < y0 || y >= y1) rest.push_back((int32_t)st_morton((unsigned)r.tx, (unsigned)y));
      r.used = true;
    }
  }
  for (const Run& r : runs)
    if (!r.used)
      for (int y = r.y0; y < r.y1; ++y) rest.push_back((int32_t)st_morton((unsigned)r.tx, (unsigned)y));
  std::sort(rest.begin(), rest.end());
#ifdef DSPH_ABLATE  // (diagnostic build only: the shipped library reads no environment variable)
  if (getenv("DSPH_STRIP_DEBUG")) {
    long a3 = 0, a45 = 0;
    int hmin = 1 << 30, hmax = 0;
    for (const Rect& r : take) { (r.wt == 3 ? a3 : a45) += (long)r.wt * r.ht; hmin = std::min(hmin, r.ht); hmax = std::max(hmax, r.ht); }
    fprintf(stderr, "build_strips: %zu rectangles (3 wide: %ld tiles, 4-5 wide: %ld tiles, heights %d..%d), %zu tiles left over\n",
            take.size(), a3, a45, take.empty() ? 0 : hmin, hmax, rest.size());
  }
#endif
  if (take.empty()) return;
  // Segment height: every segment pays 2 D + 1 run-in rows, every workgroup should get the same number of steps.  Candidates
  // from the whole rectangle down to 256 rows (measured on the partial sky of BASELINE configs[4], batch 16, strips forced:
  // 23.0 ms with 64-row segments, 21.6 with 128, 21.3 with 256 and 512, 21.6 unsegmented; the tile kernels: 21.8); the one whose busiest workgroup has the fewest steps
  // for ONE map wins -- a batch only evens things out further, items being (pair, map).
  auto cut = [&](int h, std::vector<StripPair>& out) {
    out.clear();
    for (const Rect& r : take) {
      const int X0 = 16 * r.tx, X1 = 16 * (r.tx + r.wt), Y0 = 16 * r.ty, Y1 = 16 * (r.ty + r.ht);
      const int ns = (X1 - X0 + SP_USE - 1) / SP_USE;
      const int H = Y1 - Y0, nseg = (H + h - 1) / h;
      for (int sg = 0; sg < nseg; ++sg) {
        const int ya = Y0 + (int)((int64_t)H * sg / nseg), yb = Y0 + (int)((int64_t)H * (sg + 1) / nseg);
        for (int s0 = 0; s0 < ns; s0 += 2) {
          StripPair p;
          for (int e = 0; e < 2; ++e) {
            const int s = s0 + e;
            if (s < ns) {
              p.x0[e] = X0 + SP_USE * s;
              p.w[e] = std::min(SP_USE, X1 - p.x0[e]);
            } else {
              p.x0[e] = p.x0[0];
              p.w[e] = 0;
            }
          }
          p.y0 = ya;
          p.y1 = yb;
          p.xlo = X0 - D;
          p.xhi = X1 - 1 + D;
          for (int e = 0; e < 2; ++e)  // lane 0 of the strip: D columns left of the first output column, but never past the halo
            p.xs[e] = std::min(p.x0[e] - D, p.xhi + 1 - SP_PX);
          p.ylo = Y0 - D;
          p.yhi = Y1 - 1 + D;
          out.push_back(p);
        }
      }
    }
    // Items of unequal height (ragged masks): tallest first, then dealt over the eight XCD ranges of the kernel (a range is a
    // contiguous eighth of the list), so that every XCD -- and, the kernel dealing a range to its workgroups in turn, every
    // workgroup -- gets its share of tall and short ones.  Equal heights (a full sphere): the order of the cut stays,
    // neighbours in x next to each other.
    bool ragged = false;
    for (const StripPair& p : out) ragged = ragged || (p.y1 - p.y0 != out[0].y1 - out[0].y0);
    if (ragged) {
      std::stable_sort(out.begin(), out.end(), [](const StripPair& a, const StripPair& b) { return a.y1 - a.y0 > b.y1 - b.y0; });
      std::vector<StripPair> dealt;
      dealt.reserve(out.size());
      for (size_t x = 0; x < 8; ++x)
        for (size_t i2 = x; i2 < out.size(); i2 += 8) dealt.push_back(out[i2]);
      out.swap(dealt);
    }
  };
  auto steps_of = [&](const std::vector<StripPair>& v, std::vector<int32_t>& st) {
    st.resize(v.size());
    for (size_t i2 = 0; i2 < v.size(); ++i2) st[i2] = (v[i2].y1 - v[i2].y0) + 2 * D + 1;
  };
  const int cand[] = {4096, 2048, 1024, 512, 384, 256};
  int64_t best_span = -1;
  int best_h = 256;
  {
    std::vector<StripPair> trial;
    std::vector<int32_t> st;
    for (int h : cand) {
      cut(h, trial);
      steps_of(trial, st);
      const int64_t span = strip_makespan(st, 1, num_cu);
      if (best_span < 0 || span < best_span) { best_span = span; best_h = h; }
    }
  }
  if (opt.strip_seg > 0) best_h = std::max(16, opt.strip_seg);  // (tuning, DSPH_OPT_STRIP_SEG: the segment height, in rows)
  cut(best_h, pairs);
  steps_of(pairs, steps);
  cut(1 << 30, whole);
  // The quad strips: rectangles that stand side by side over the same rows become one (a full base pixel: one rectangle of all
  // its interior columns), cut into strips of 56 output columns (the last one narrower), uncut along y.
  if (qstrips && D == QS_D) {
    std::vector<Rect> wide(take);
    std::sort(wide.begin(), wide.end(), [](const Rect& a, const Rect& b) { return a.ty != b.ty ? a.ty < b.ty : (a.ht != b.ht ? a.ht < b.ht : a.tx < b.tx); });
    std::vector<Rect> merged;
    for (const Rect& r : wide) {
      if (!merged.empty() && merged.back().ty == r.ty && merged.back().ht == r.ht && merged.back().tx + merged.back().wt == r.tx) merged.back().wt += r.wt;
      else merged.push_back(r);
    }
    for (const Rect& r : merged) {
      const int X0 = 16 * r.tx, X1 = 16 * (r.tx + r.wt), Y0 = 16 * r.ty, Y1 = 16 * (r.ty + r.ht);
      for (int x0 = X0; x0 < X1; x0 += QS_USE) {
        QStrip q{};
        q.x0 = x0; q.w = std::min(QS_USE, X1 - x0); q.xs = x0 - D;
        q.y0 = Y0; q.y1 = Y1;
        q.xlo = X0 - D; q.xhi = X1 - 1 + D; q.ylo = Y0 - D; q.yhi = Y1 - 1 + D;
        qstrips->push_back(q);
      }
    }
  }
#ifdef DSPH_ABLATE
  if (getenv("DSPH_STRIP_DEBUG"))
    fprintf(stderr, "build_strips: segments of %d rows, %zu pairs, busiest workgroup %ld / %ld / %ld steps for 1 / 4 / 16 maps; tile cost "
            "per map in the same unit %ld\n", best_h, pairs.size(), (long)strip_makespan(steps, 1, num_cu), (long)strip_makespan(steps, 4, num_cu),
            (long)strip_makespan(steps, 16, num_cu), (long)(*n_taken * 187 / (30 * num_cu)));
#endif
}


// ---- table-addressed quad strips (round 6) ----------------------------------------------------------------------------------
// A candidate tile of the logical grid: its eight neighbour TILES by direction (kDirX / kDirY order: W NW N NE E SE S SW), each
// verified to continue this tile's own 16 x 16 Morton square by a pure translation over the D rows / columns a strip's halo
// reaches into it.  Class R: the neighbours of the virtual Morton plane (what the classification verified).  Class T: read off
// the embedding embed_tile() found and verified (row[] = the row of every plane cell).
struct QCand {
  int32_t tile;
  int32_t nbr[8];
  int32_t tix;       // position in the class-T list (tables), -1 for a class-R tile
  int32_t sheet, u, v;
  bool taken;
  bool barred;       // stays with the tile kernels whatever the rectangles would gain (a cell no rectangle may cover)
};

static bool links_from_table(const int32_t* row, int D, int32_t ntiles, int32_t nbr[8]) {
  static const int ddx[8] = {-1, -1, 0, 1, 1, 1, 0, -1}, ddy[8] = {0, 1, 1, 1, 0, -1, -1, -1};
  for (int d = 0; d < 8; ++d) {
    const int xa = ddx[d] < 0 ? ST_DMAX - D : (ddx[d] == 0 ? ST_DMAX : ST_DMAX + ST_TILE);
    const int xb = ddx[d] < 0 ? ST_DMAX : (ddx[d] == 0 ? ST_DMAX + ST_TILE : ST_DMAX + ST_TILE + D);
    const int ya = ddy[d] < 0 ? ST_DMAX - D : (ddy[d] == 0 ? ST_DMAX : ST_DMAX + ST_TILE);
    const int yb = ddy[d] < 0 ? ST_DMAX : (ddy[d] == 0 ? ST_DMAX + ST_TILE : ST_DMAX + ST_TILE + D);
    int64_t base = -1;
    for (int y = ya; y < yb; ++y)
      for (int x = xa; x < xb; ++x) {
        const int64_t r = row[st_cell_off((unsigned)x, (unsigned)y) / 64u];
        const int64_t b = r - (int64_t)st_morton((unsigned)(x - ST_DMAX) & 15u, (unsigned)(y - ST_DMAX) & 15u);
        if (b < 0 || (b & (FUSED_P - 1)) != 0 || (base >= 0 && b != base)) return false;
        base = b;
      }
    if (base < 0 || base / FUSED_P >= ntiles) return false;
    nbr[d] = (int32_t)(base / FUSED_P);
  }
  return true;
}

// What a rectangle of w x h tiles saves against the tile kernels, in units of 0.1 us of one CU and one map (the constants of
// strips_apply: 18.7 us per tile, 2.8 us per strip step): strips of 56 output columns, 2 D + 1 run-in steps each.
// (the K = 8 strips -- cheb_qstrip8_kernel.h, D = 7 -- have 50 output columns, 16 run-in steps, ~2.6 us a step, against 27.5 us
// per tile on the breadth-first tile kernel with its 7-ring halo)
static int64_t qt_gain(int w, int h, int D) {
  const int use = D == Q8_D ? Q8_USE : QS_PX - 2 * D;
  const int64_t ns = (16 * (int64_t)w + use - 1) / use;
  if (D == Q8_D) return (int64_t)w * h * 275 - ns * (16 * (int64_t)h + Q8_RUNIN) * 26;
  // (a run of rows costs more than its 2 D + 1 run-in steps: two barriers and a waited-for first row before the loop, the step
  // count rounded up to a multiple of three -- measured at the headline map, 254 strips against 216: about 14 steps a run)
  return (int64_t)w * h * 187 - ns * (16 * (int64_t)h + 2 * D + 6) * 28;
}

// Rectangles of candidate tiles on the logical grid and their strips.  cands: every interior class-R tile and every eligible
// interior class-T tile.  Out: strips (coordinates relative to the rectangle's table: the rectangle's first pixel is (16, 16)),
// the tables, `taken` set in cands.
static void build_qtstrips(std::vector<QCand>& cands, int32_t ntiles, int D, std::vector<QStrip>& strips, std::vector<int32_t>& tab,
                           int64_t* n_taken) {
  strips.clear();
  tab.clear();
  *n_taken = 0;
  const int n = (int)cands.size();
  if (n == 0) return;
  std::vector<int32_t> cand_of((size_t)ntiles, -1);
  for (int i = 0; i < n; ++i) cand_of[(size_t)cands[i].tile] = i;
  static const int ddx[8] = {-1, -1, 0, 1, 1, 1, 0, -1}, ddy[8] = {0, 1, 1, 1, 0, -1, -1, -1};
  auto cand_at = [&](int32_t tile) -> int { return tile >= 0 && tile < ntiles ? cand_of[(size_t)tile] : -1; };
  // sheets: logical coordinates by breadth-first search over the four axis links that both ends agree on
  std::unordered_map<uint64_t, int32_t> at;  // (sheet, u, v) -> candidate
  constexpr int64_t OFF = 1 << 20;
  auto key_of = [&](int32_t sheet, int64_t u, int64_t v) { return ((uint64_t)sheet << 44) | ((uint64_t)(u + OFF) << 22) | (uint64_t)(v + OFF); };
  int32_t nsheets = 0;
  std::vector<int32_t> queue;
  for (int i = 0; i < n; ++i) cands[i].sheet = -1;
  for (int i0 = 0; i0 < n; ++i0) {
    if (cands[i0].sheet >= 0) continue;
    if (nsheets >= (1 << 19)) break;
    const int32_t sh = nsheets++;
    cands[i0].sheet = sh; cands[i0].u = 0; cands[i0].v = 0;
    at[key_of(sh, 0, 0)] = i0;
    queue.assign(1, i0);
    for (size_t h = 0; h < queue.size(); ++h) {
      const QCand a = cands[(size_t)queue[h]];
      for (int d = 0; d < 8; d += 2) {
        const int b = cand_at(a.nbr[d]);
        if (b < 0 || cands[(size_t)b].sheet >= 0 || cands[(size_t)b].nbr[(d + 4) & 7] != a.tile) continue;
        const int64_t u = (int64_t)a.u + ddx[d], v = (int64_t)a.v + ddy[d];
        if (u <= -OFF + 2 || u >= OFF - 2 || v <= -OFF + 2 || v >= OFF - 2) continue;
        const uint64_t k = key_of(sh, u, v);
        if (at.find(k) != at.end()) continue;  // (a sphere unrolled onto the plane meets itself again: the place is taken)
        cands[(size_t)b].sheet = sh; cands[(size_t)b].u = (int32_t)u; cands[(size_t)b].v = (int32_t)v;
        at[k] = b;
        queue.push_back(b);
      }
    }
  }
  // per sheet: occupancy grid over the bounding box, greedy extraction of the rectangle with the largest gain
  std::vector<std::vector<int32_t>> members((size_t)nsheets);
  for (int i = 0; i < n; ++i)
    if (cands[i].sheet >= 0) members[(size_t)cands[i].sheet].push_back(i);
  for (int32_t sh = 0; sh < nsheets; ++sh) {
    const std::vector<int32_t>& mem = members[(size_t)sh];
    if (mem.size() < 6) continue;
    int u0 = 1 << 30, u1 = -(1 << 30), v0 = 1 << 30, v1 = -(1 << 30);
    for (int32_t i : mem) { u0 = std::min(u0, cands[(size_t)i].u); u1 = std::max(u1, cands[(size_t)i].u); v0 = std::min(v0, cands[(size_t)i].v); v1 = std::max(v1, cands[(size_t)i].v); }
    const int64_t GW = (int64_t)u1 - u0 + 1, GH = (int64_t)v1 - v0 + 1;
    if (GW * GH > (1ll << 26)) continue;  // (a sheet that sprawls: left to the tile kernels)
    std::vector<int32_t> grid((size_t)(GW * GH), -1);  // candidate index, -1 empty, -2 blocked
    for (int32_t i : mem) grid[(size_t)((cands[(size_t)i].v - v0) * GW + (cands[(size_t)i].u - u0))] = cands[(size_t)i].barred ? -2 : i;
    std::vector<int32_t> hgt((size_t)GW);
    std::vector<std::pair<int, int>> stack;  // (start column, height)
    // (every extraction sweeps the sheet once: a budget of sweeps bounds the set-up time on ragged masks -- what is not taken
    // by then stays with the tile kernels)
    for (int64_t sweeps = 0; sweeps * GW * GH < (3ll << 28) && sweeps < 8192; ++sweeps) {
      // the best (gain) rectangle of free cells: histogram of free runs along v, one sweep per row
      int64_t best = 0;
      int bu = 0, bv = 0, bw = 0, bh = 0;
      std::fill(hgt.begin(), hgt.end(), 0);
      for (int64_t y = 0; y < GH; ++y) {
        for (int64_t x = 0; x < GW; ++x) hgt[(size_t)x] = grid[(size_t)(y * GW + x)] >= 0 ? hgt[(size_t)x] + 1 : 0;
        stack.clear();
        for (int64_t x = 0; x <= GW; ++x) {
          const int hx = x < GW ? hgt[(size_t)x] : 0;
          int start = (int)x;
          while (!stack.empty() && stack.back().second > hx) {
            const int s0 = stack.back().first, hh = stack.back().second;
            stack.pop_back();
            const int wmax = (int)x - s0;
            // the widest rectangle of this height, or one a few columns narrower where the last strip would be nearly empty
            for (int w = std::min(wmax, 1000); w >= std::max(1, std::min(wmax, 1000) - 3); --w) {  // (1,000: the kernels keep a tile column in ten bits)
              const int64_t g = qt_gain(w, hh, D);
              if (g > best) { best = g; bu = s0; bv = (int)y - hh + 1; bw = w; bh = hh; }
            }
            start = s0;
          }
          if (hx > 0 && (stack.empty() || stack.back().second < hx)) stack.push_back({start, hx});
        }
      }
      if (best <= 0) break;
      // the table of tile bases: the rectangle and one ring of tiles around it
      const int TW = bw + 2, TH = bh + 2;
      std::vector<int32_t> t((size_t)TW * TH, -1);
      bool good = true;
      auto inside = [&](int i, int j) { return i >= 1 && i <= bw && j >= 1 && j <= bh; };
      for (int j = 1; j <= bh && good; ++j)
        for (int i = 1; i <= bw && good; ++i) {
          const int32_t c = grid[(size_t)((bv + j - 1) * GW + (bu + i - 1))];
          const QCand& q = cands[(size_t)c];
          t[(size_t)j * TW + i] = q.tile;
          for (int d = 0; d < 8 && good; ++d) {
            const int i2 = i + ddx[d], j2 = j + ddy[d];
            int32_t& slot = t[(size_t)j2 * TW + i2];
            if (inside(i2, j2)) {  // the neighbour inside the rectangle must be the tile the grid holds there
              const int32_t c2 = grid[(size_t)((bv + j2 - 1) * GW + (bu + i2 - 1))];
              if (cands[(size_t)c2].tile != q.nbr[d]) good = false;
            } else if (slot < 0) slot = q.nbr[d];
            else if (slot != q.nbr[d]) good = false;  // two tiles of the rectangle name different tiles for one place of the ring
          }
        }
      if (!good) {  // (a seam the sheet's coordinates paper over: these tiles stay with the tile kernels)
        for (int j = 0; j < bh; ++j)
          for (int i = 0; i < bw; ++i) grid[(size_t)((bv + j) * GW + (bu + i))] = -2;
        continue;
      }
      const int32_t toff = (int32_t)tab.size();
      for (int32_t tile : t) tab.push_back(tile < 0 ? 0 : tile * FUSED_P);  // (the four corners of the ring of a 1-wide rectangle ... are never -1: every ring place has a neighbour inside)
      for (int j = 0; j < bh; ++j)
        for (int i = 0; i < bw; ++i) {
          int32_t& c = grid[(size_t)((bv + j) * GW + (bu + i))];
          cands[(size_t)c].taken = true;
          c = -2;
        }
      *n_taken += (int64_t)bw * bh;
      const int X0 = 16, X1 = 16 + 16 * bw, Y0 = 16, Y1 = 16 + 16 * bh;
      // output columns of a strip, and how far left of the first one its lane 0 stands: the kernels take a lane's four pixels
      // for one aligned group of four inside one tile, so both are multiples of four (D = 7: 48 columns behind 8 of lead-in)
      const int use = D == Q8_D ? Q8_USE : QS_PX - 2 * D, lead = D == Q8_D ? 8 : D;
      for (int x0 = X0; x0 < X1; x0 += use) {
        QStrip q{};
        q.x0 = x0; q.w = std::min(use, X1 - x0); q.xs = x0 - lead;
        q.y0 = Y0; q.y1 = Y1;
        q.xlo = X0 - D; q.xhi = X1 - 1 + D; q.ylo = Y0 - D; q.yhi = Y1 - 1 + D;
        q.tab = toff; q.tws = TW;
        strips.push_back(q);
      }
    }
  }
  // The kernel cuts the tape of all strips' rows into equal pieces, and a piece pays 2 D + 1 run-in steps for every strip that
  // begins in it: the greedy extraction leaves the low strips (a few tiles high) at the end of the list, where a piece would
  // hold dozens of them (measured at the headline map: the last pieces took 14 % longer than the first, and the forward with
  // them).  So the low strips are dealt out evenly between the tall ones, by rows.
  {
    std::vector<QStrip> tall, low;
    int64_t rows_tall = 0, rows_low = 0;
    for (const QStrip& q : strips) {
      if (q.y1 - q.y0 >= 256) { tall.push_back(q); rows_tall += q.y1 - q.y0; }
      else { low.push_back(q); rows_low += q.y1 - q.y0; }
    }
    if (!tall.empty() && !low.empty()) {
      strips.clear();
      size_t li = 0;
      int64_t done_tall = 0, done_low = 0;
      for (const QStrip& q : tall) {
        strips.push_back(q);
        done_tall += q.y1 - q.y0;
        while (li < low.size() && done_low * rows_tall < rows_low * done_tall) {
          strips.push_back(low[li]);
          done_low += low[li].y1 - low[li].y0;
          ++li;
        }
      }
      for (; li < low.size(); ++li) strips.push_back(low[li]);
    }
  }
}

// Breadth-first rings of every tile; uploads the tables.  Returns a reference to the cached entry.
// full: BFS tables of every tile (planes / weight-gradient modes of the BFS kernel); otherwise the tiles are first
// classified and only the class-G ones (not a plain 2-D stencil square) get BFS tables, the rest go to d_rlist.
static const FusedTiles& get_tiles(const dsph_plan* plan, int D, bool full = false) {
  FusedPlan* fp = plan->fused;
  std::lock_guard<std::mutex> lock(fp->mu);
  DeviceGuard guard(plan->device);  // tables live on the plan's device, whatever the caller's current one is
  const int key = 2 * D + (full ? 1 : 0);
  auto it = fp->by_depth.find(key);
  if (it != fp->by_depth.end()) return it->second;
  if (fp->host_released) {  // the ELL arrays are gone: report "not tileable" without caching anything
    static const FusedTiles none;
    return none;
  }
  if (fp->wide && !(D == 1 && full)) {  // a wide graph has the tiled step's tables and nothing else
    static const FusedTiles none;
    return none;
  }
  FusedTiles& ft = fp->by_depth[key];
  ft.D = D;
  ft.width = fp->wide ? tstep_width(plan->width) : template_width(plan->width);
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
  std::vector<int32_t> ring, next, interior, boundary, r_interior, r_boundary;
  std::vector<unsigned char> cls((size_t)ntiles, 0);
  if (!full && D <= ST_DMAX && plan->opt.use_struct) {
    if (!fp->rows_tried) {
      fp->rows_tried = true;
      if (struct_build_rows(plan, &fp->d_gvals8, &fp->d_gdiag, &fp->d_rowflag) != DSPH_OK) {
        if (fp->d_gvals8) (void)hipFree(fp->d_gvals8);
        if (fp->d_gdiag) (void)hipFree(fp->d_gdiag);
        if (fp->d_rowflag) (void)hipFree(fp->d_rowflag);
        fp->d_gvals8 = fp->d_gdiag = nullptr;
        fp->d_rowflag = nullptr;
      }
    }
    if (fp->d_rowflag && struct_classify_tiles(plan, fp->d_rowflag, ntiles, D, out_rows, cls.data()) != DSPH_OK)
      std::fill(cls.begin(), cls.end(), 0);
  }
  // K = 8 quad strips: which tiles' 7-ring regions are regular squares of the Morton plane (the same row flags, a deeper ring).
  // Every tile still gets its breadth-first tables (other shapes and the weight gradient run on them).  On a sharded plan
  // the candidates are the INTERIOR tiles (region inside the rank's own rows): they run with the interior part of a forward.
  std::vector<unsigned char> cls8;
  if (D == Q8_D && plan->opt.use_struct && plan->opt.strips != 2 && plan->opt.strip_form == 0 && !fp->wide) {
    if (!fp->rows_tried) {
      fp->rows_tried = true;
      if (struct_build_rows(plan, &fp->d_gvals8, &fp->d_gdiag, &fp->d_rowflag) != DSPH_OK) {
        if (fp->d_gvals8) (void)hipFree(fp->d_gvals8);
        if (fp->d_gdiag) (void)hipFree(fp->d_gdiag);
        if (fp->d_rowflag) (void)hipFree(fp->d_rowflag);
        fp->d_gvals8 = fp->d_gdiag = nullptr;
        fp->d_rowflag = nullptr;
      }
    }
    if (fp->d_rowflag) {
      cls8.assign((size_t)ntiles, 0);
      if (struct_classify_tiles(plan, fp->d_rowflag, ntiles, D, out_rows, cls8.data(), Q8_D) != DSPH_OK) cls8.clear();
    }
  }
  // class-T candidates: whatever the classification left over, when the structured kernel is in play at all
  const bool try_tables = !full && D <= ST_DMAX && plan->opt.use_struct && plan->opt.use_tables && fp->d_rowflag != nullptr;
  std::vector<int32_t> t_interior, t_boundary, trow_i, trow_b, e_row(ST_CELLS), h_key, h_slot;
  std::vector<float> tval_i, tval_b, e_val((size_t)ST_CELLS * ST_TABV);
  int rmax = 0, emax = 0;
  int64_t ell_rows = 0;
  for (int t = 0; t < ntiles; ++t) {
    const int64_t r0 = (int64_t)t * FUSED_P, r1 = std::min<int64_t>(out_rows, r0 + FUSED_P);
    const size_t base = region.size();
    if (base > 0x7fffffffULL - 70000) return ft;  // offsets are int32
    tile_off[t] = (int32_t)base;
    if (cls[t] & 1) {  // class R: no BFS tables
      ((cls[t] & 2) ? r_interior : r_boundary).push_back(t);
      ell_off[t] = ell_rows;
      continue;
    }
    if (try_tables) {  // class T: the structured kernel with per-tile tables
      bool inner = false;
      if (embed_tile(plan, cols, vals, W, t, D, out_rows, e_row.data(), e_val.data(), &inner, h_key, h_slot)) {
        std::vector<int32_t>& lst = inner ? t_interior : t_boundary;
        std::vector<int32_t>& rw = inner ? trow_i : trow_b;
        std::vector<float>& vl = inner ? tval_i : tval_b;
        lst.push_back(t);
        rw.insert(rw.end(), e_row.begin(), e_row.end());
        vl.insert(vl.end(), e_val.begin(), e_val.end());
        ell_off[t] = ell_rows;
        continue;
      }
    }
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
      // Emit the ring so that (local index & 3) == (row id & 3) wherever possible.  On a HEALPix map
      // the low two NEST bits are the pixel's (x, y) parity; the recurrence reads the j-th neighbours
      // of four rows with four different parities in one LDS access group, and those neighbours then
      // sit in four different bank quarters -- halo rows included, not only the tile's own rows.
      {
        std::vector<int32_t> bucket[4];
        for (int32_t r : next) bucket[r & 3].push_back(r);
        size_t head[4] = {0, 0, 0, 0};
        size_t left = next.size();
        next.clear();
        while (left > 0) {
          const int want = (int)((region.size() - base + next.size()) & 3);
          int take = want;
          if (head[take] >= bucket[take].size()) {  // that parity is used up: take from the fullest
            size_t best = 0;
            for (int q = 0; q < 4; ++q) {
              const size_t rem = bucket[q].size() - head[q];
              if (rem > best) { best = rem; take = q; }
            }
          }
          next.push_back(bucket[take][head[take]++]);
          --left;
        }
      }
      for (int32_t r : next) {
        local[r] = (int32_t)(region.size() - base);
        region.push_back(r);
      }
      re[d] = (int32_t)(region.size() - base);
      ring.swap(next);
    }
    for (int d = D + 1; d <= FUSED_DMAX; ++d) re[d] = re[D];
    const int R = re[D], E = re[D - 1];
    {
      bool inner = true;
      for (size_t i = base; i < region.size() && inner; ++i) inner = region[i] < out_rows;
      (inner ? interior : boundary).push_back(t);
    }
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
      if (fp->wide) {
        // The tiled step (cheb_tstep.hip) reads slot j of four consecutive rows in one 16-lane LDS access: conflict-free when
        // the four neighbours' local indices differ mod 4.  So the slots of a row are dealt by residue: slot j of local row i
        // takes a neighbour with (local index & 3) == ((i + j) & 3) while there is one, else one from the fullest residue
        // class (the sum's ORDER changes, not its terms).
        std::vector<std::pair<uint16_t, float>> bucket[4];
        for (int j = 0; j < W; ++j)
          if (v[j] != 0.f) bucket[local[c[j]] & 3].push_back({(uint16_t)local[c[j]], v[j]});
        size_t head[4] = {0, 0, 0, 0};
        for (int j = 0; j < WT; ++j) {
          int b = (i + j) & 3;
          if (head[b] >= bucket[b].size()) {
            size_t best = 0;
            int bb = -1;
            for (int q = 0; q < 4; ++q)
              if (bucket[q].size() - head[q] > best) { best = bucket[q].size() - head[q]; bb = q; }
            b = bb;
          }
          uint16_t lc = (uint16_t)i;
          float lv = 0.f;
          if (b >= 0) {
            lc = bucket[b][head[b]].first;
            lv = bucket[b][head[b]].second;
            ++head[b];
          } else {
            const int want = (i & ~3) | ((i + j) & 3);  // padding: a row of this quad with the slot's residue (value 0)
            lc = (uint16_t)(want < R ? want : i);
          }
          lcols[lbase + (size_t)j * E + i] = lc;
          lvals[lbase + (size_t)j * E + i] = lv;
        }
        continue;
      }
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

  auto up = [](void** dst, const void* src, size_t bytes) -> bool {
    if (bytes == 0) bytes = 16;
    if (hipMalloc(dst, bytes) != hipSuccess) return false;
    return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess;
  };
  if (lcols.empty()) { lcols.push_back(0); lvals.push_back(0.f); }
  // Small maps (BASELINE configs[0]: nside 64, 192 tiles -- 48 class R, 120 class T, 24 class G) are bound by launches, not by
  // work: every structured tile fits the device at once, and two launches (class R, then class T) take twice as long as one.
  // Where the strips take nothing anyway, the class-R tiles get tables too and join the class-T launch.
  if (try_tables && D <= SP_DMAX && !(r_interior.empty() && r_boundary.empty()) &&
      r_interior.size() + r_boundary.size() + t_interior.size() + t_boundary.size() <= (size_t)SMALL_MAP_TILES) {
    int64_t taken = 0;
    if (plan->opt.strips != 2) {
      std::vector<StripPair> p0, ip0;
      std::vector<int32_t> rest0, steps0;
      build_strips(r_interior, D, fp->num_cu, plan->opt, p0, rest0, &taken, steps0, ip0);
    }
    if (taken == 0) {
      std::vector<int32_t> keep_i, keep_b;
      for (int pass = 0; pass < 2; ++pass) {
        for (int32_t t : pass == 0 ? r_interior : r_boundary) {
          bool inner = false;
          if (embed_tile(plan, cols, vals, W, t, D, out_rows, e_row.data(), e_val.data(), &inner, h_key, h_slot)) {
            (pass == 0 ? t_interior : t_boundary).push_back(t);
            std::vector<int32_t>& rw = pass == 0 ? trow_i : trow_b;
            std::vector<float>& vl = pass == 0 ? tval_i : tval_b;
            rw.insert(rw.end(), e_row.begin(), e_row.end());
            vl.insert(vl.end(), e_val.begin(), e_val.end());
          } else {
            (pass == 0 ? keep_i : keep_b).push_back(t);
          }
        }
      }
      r_interior.swap(keep_i);
      r_boundary.swap(keep_b);
    }
  }
  std::vector<int32_t> all_tiles;  // every tile, whatever its class: interior ones first
  all_tiles.insert(all_tiles.end(), r_interior.begin(), r_interior.end());
  all_tiles.insert(all_tiles.end(), t_interior.begin(), t_interior.end());
  all_tiles.insert(all_tiles.end(), interior.begin(), interior.end());
  ft.n_all_interior = (int)all_tiles.size();
  all_tiles.insert(all_tiles.end(), r_boundary.begin(), r_boundary.end());
  all_tiles.insert(all_tiles.end(), t_boundary.begin(), t_boundary.end());
  all_tiles.insert(all_tiles.end(), boundary.begin(), boundary.end());
  ft.n_all = (int)all_tiles.size();
  if (all_tiles.empty()) all_tiles.push_back(0);
  ft.n_interior = (int)interior.size();
  interior.insert(interior.end(), boundary.begin(), boundary.end());
  ft.n_part = (int)interior.size();
  if (interior.empty()) interior.push_back(0);
  std::vector<StripPair> pairs, ipairs;
  std::vector<int32_t> rrest;
  ft.n_strip_tiles = 0;
  ft.strip_steps.clear();
  ft.strip_span.clear();
  ft.strip_forced = plan->opt.strips == 1;
  std::vector<QStrip> qstrips;
  if (!full && D <= SP_DMAX && plan->opt.strips != 2)
    build_strips(r_interior, D, fp->num_cu, plan->opt, pairs, rrest, &ft.n_strip_tiles, ft.strip_steps, ipairs, nullptr);
  else
    rrest = r_interior;
  // The quad strips (K = 5): rectangles on the logical tile grid, addressed through tables (build_qtstrips) -- every interior
  // class-R tile and every interior class-T tile whose eight neighbour tiles are pure translations.
  std::vector<int32_t> qtab, qrrest, qt_list, qt_rows;
  std::vector<float> qt_vals;
  std::vector<int32_t> patch_rows;
  std::vector<float> patch_vals;
  ft.n_qstrip_tiles = 0;
  if (!full && D == QS_D && plan->opt.strips != 2 && plan->opt.strip_form == 0 && fp->d_gvals8 != nullptr) {
    std::vector<QCand> cands;
    cands.reserve(r_interior.size() + t_interior.size());
    static const int ddx[8] = {-1, -1, 0, 1, 1, 1, 0, -1}, ddy[8] = {0, 1, 1, 1, 0, -1, -1, -1};
    for (int32_t t : r_interior) {
      QCand c{};
      c.tile = t; c.tix = -1; c.taken = false;
      const int tx = (int)st_compress((unsigned)t), ty = (int)st_compress((unsigned)t >> 1);
      for (int d = 0; d < 8; ++d) {
        const int nx = tx + ddx[d], ny = ty + ddy[d];
        const int64_t nt = nx < 0 || ny < 0 ? -1 : (int64_t)st_morton((unsigned)nx, (unsigned)ny);
        c.nbr[d] = nt >= 0 && nt < ntiles ? (int32_t)nt : -1;
      }
      cands.push_back(c);
    }
    const size_t n_rc = cands.size();
    // The kernel does not clamp a row to the strip's halo (cheb_qstrip_kernel.h, "rows need no clamp"): past the halo it reads
    // rows of the table's ring tiles that feed nothing -- rows that must exist.  The one tile whose rows may not is the map's
    // last, incomplete one: no tile beside it is a candidate, so it is in no rectangle's ring.
    const int32_t ragged = out_rows % FUSED_P != 0 ? ntiles - 1 : -1;
    auto beside_ragged = [&](const QCand& c) {
      if (ragged < 0) return false;
      if (c.tile == ragged) return true;
      for (int d = 0; d < 8; ++d)
        if (c.nbr[d] == ragged) return true;
      return false;
    };
#ifdef DSPH_ABLATE  // (diagnostic build only -- make ABLATE=1: the shipped library reads no environment variable)
    const bool only_r = getenv("DSPH_QT_ONLY_R") != nullptr;  // (the strips of round 5's tile set on this round's kernel: tools/ab_r5_r6.sh)
#else
    const bool only_r = false;
#endif
    for (size_t i = 0; i < t_interior.size() && !only_r; ++i) {
      QCand c{};
      c.tile = t_interior[i]; c.tix = (int32_t)i; c.taken = false;
      if (links_from_table(&trow_i[i * ST_CELLS], D, ntiles, c.nbr)) cands.push_back(c);
    }
    for (QCand& c : cands) c.barred = beside_ragged(c);
    build_qtstrips(cands, ntiles, D, qstrips, qtab, &ft.n_qstrip_tiles);
    std::vector<unsigned char> t_taken(t_interior.size(), 0);
    for (size_t i = 0; i < cands.size(); ++i) {
      if (i < n_rc) { if (!cands[i].taken) qrrest.push_back(cands[i].tile); }
      else if (cands[i].taken) t_taken[(size_t)cands[i].tix] = 1;
    }
    // (r_interior is sorted; cands keeps its order)
    ft.n_qrrest_interior = (int)qrrest.size();
    qrrest.insert(qrrest.end(), r_boundary.begin(), r_boundary.end());
    for (size_t i = 0; i < t_interior.size(); ++i) {
      if (!t_taken[i]) {
        qt_list.push_back(t_interior[i]);
        qt_rows.insert(qt_rows.end(), trow_i.begin() + i * ST_CELLS, trow_i.begin() + (i + 1) * ST_CELLS);
        qt_vals.insert(qt_vals.end(), tval_i.begin() + i * ST_CELLS * ST_TABV, tval_i.begin() + (i + 1) * ST_CELLS * ST_TABV);
        continue;
      }
      // a class-T tile on the strips: the strips read L~ by direction from gvals8 / gdiag, which the row pass filled from
      // the VIRTUAL Morton plane (a neighbour beyond a base-pixel border has no direction there and was dropped).  The
      // tile's verified embedding has every evaluated row's values by direction: written over the rows' entries (no other
      // kernel reads them for rows that are irregular in the virtual plane; for regular rows the two agree).
      const int blo = ST_DMAX - D + 1, bhi = ST_DMAX + ST_TILE - 1 + D - 1;  // rings 0 .. D-1
      for (int y = blo; y <= bhi; ++y)
        for (int x = blo; x <= bhi; ++x) {
          const size_t cell = i * ST_CELLS + st_cell_off((unsigned)x, (unsigned)y) / 64u;
          patch_rows.push_back(trow_i[cell]);
          patch_vals.insert(patch_vals.end(), tval_i.begin() + cell * ST_TABV, tval_i.begin() + cell * ST_TABV + 9);
        }
    }
    ft.n_qt_interior = (int)qt_list.size();
    qt_list.insert(qt_list.end(), t_boundary.begin(), t_boundary.end());
    qt_rows.insert(qt_rows.end(), trow_b.begin(), trow_b.end());
    qt_vals.insert(qt_vals.end(), tval_b.begin(), tval_b.end());
  }
  // K = 8: rectangles of the depth-7 regular tiles (virtual Morton neighbours: what the classification verified)
  std::vector<QStrip> q8strips;
  std::vector<int32_t> q8tab, q8rest;
  ft.n_q8_tiles = 0;
  if (!cls8.empty()) {
    std::vector<QCand> cands;
    static const int ddx[8] = {-1, -1, 0, 1, 1, 1, 0, -1}, ddy[8] = {0, 1, 1, 1, 0, -1, -1, -1};
    for (int32_t t = 0; t < ntiles; ++t) {
      if ((cls8[(size_t)t] & 3) != 3) continue;
      QCand c{};
      c.tile = t; c.tix = -1; c.taken = false;
      const int tx = (int)st_compress((unsigned)t), ty = (int)st_compress((unsigned)t >> 1);
      for (int d = 0; d < 8; ++d) {
        const int nx = tx + ddx[d], ny = ty + ddy[d];
        const int64_t nt = nx < 0 || ny < 0 ? -1 : (int64_t)st_morton((unsigned)nx, (unsigned)ny);
        c.nbr[d] = nt >= 0 && nt < ntiles ? (int32_t)nt : -1;
      }
      cands.push_back(c);
    }
    build_qtstrips(cands, ntiles, D, q8strips, q8tab, &ft.n_q8_tiles);
    std::vector<unsigned char> took((size_t)ntiles, 0);
    for (const QCand& c : cands)
      if (c.taken) took[(size_t)c.tile] = 1;
    for (int i = 0; i < ft.n_part; ++i) {
      if (i == ft.n_interior) ft.n_q8rest_interior = (int)q8rest.size();
      if (!took[(size_t)interior[(size_t)i]]) q8rest.push_back(interior[(size_t)i]);
    }
    if (ft.n_interior >= ft.n_part) ft.n_q8rest_interior = (int)q8rest.size();
  }
  ft.n_q8strips = (int)q8strips.size();
  ft.h_q8strips = q8strips;
  ft.h_q8tab = q8tab;
  ft.n_q8rest = (int)q8rest.size();
  std::vector<int32_t> q8prefix(1, 0);
  for (const QStrip& q : q8strips) q8prefix.push_back(q8prefix.back() + (q.y1 - q.y0));
  ft.q8tape_rows = q8prefix.back();
  if (q8strips.empty()) q8strips.push_back(QStrip());
  if (q8rest.empty()) q8rest.push_back(0);
  q8tab.resize(q8tab.size() + 8, 0);
  ft.n_qrrest = (int)qrrest.size();
  ft.n_qt = (int)qt_list.size();
  if (qrrest.empty()) qrrest.push_back(0);
  if (qt_list.empty()) { qt_list.push_back(0); qt_rows.push_back(0); qt_vals.push_back(0.f); }
  ft.h_qtab = qtab;
  qtab.resize(qtab.size() + 8, 0);  // (a lane's five tile columns are read by scalar loads from the strip's first column on: readable past the end)
  ft.n_pairs = (int)pairs.size();
  ft.h_pairs = pairs;
  ft.n_qstrips = (int)qstrips.size();
  ft.h_qstrips = qstrips;
  std::vector<int32_t> qprefix(1, 0);
  for (const QStrip& q : qstrips) qprefix.push_back(qprefix.back() + (q.y1 - q.y0));
  ft.qtape_rows = qprefix.back();
  if (qstrips.empty()) qstrips.push_back(QStrip());
  ft.n_ipairs = (int)ipairs.size();
  ft.ipair_h.clear();
  ft.ipair_second.clear();
  ft.iseg.clear();
  for (const StripPair& ip : ipairs) {
    ft.ipair_h.push_back(ip.y1 - ip.y0);
    ft.ipair_second.push_back(ip.w[1] > 0 ? 1 : 0);
  }
  if (ipairs.empty()) ipairs.push_back(StripPair());
  ft.n_rrest_interior = (int)rrest.size();
  rrest.insert(rrest.end(), r_boundary.begin(), r_boundary.end());
  ft.n_rrest = (int)rrest.size();
  // every tile the QUAD strips leave over, whatever its class
  std::vector<int32_t> nonq(qrrest.begin(), qrrest.begin() + ft.n_qrrest);
  nonq.insert(nonq.end(), qt_list.begin(), qt_list.begin() + ft.n_qt);
  if (ft.n_qstrip_tiles == 0) {  // (no quad strips: the list is not used; keep it the complement of nothing)
    nonq.assign(r_interior.begin(), r_interior.end());
    nonq.insert(nonq.end(), r_boundary.begin(), r_boundary.end());
    nonq.insert(nonq.end(), t_interior.begin(), t_interior.end());
    nonq.insert(nonq.end(), t_boundary.begin(), t_boundary.end());
  }
  nonq.insert(nonq.end(), interior.begin(), interior.begin() + ft.n_part);
  ft.n_nonq = (int)nonq.size();
  if (nonq.empty()) nonq.push_back(0);
  if (rrest.empty()) rrest.push_back(0);
  if (pairs.empty()) pairs.push_back(StripPair());
  ft.n_r_interior = (int)r_interior.size();
  r_interior.insert(r_interior.end(), r_boundary.begin(), r_boundary.end());
  ft.n_r = (int)r_interior.size();
  if (r_interior.empty()) r_interior.push_back(0);
  ft.n_t_interior = (int)t_interior.size();
  t_interior.insert(t_interior.end(), t_boundary.begin(), t_boundary.end());
  trow_i.insert(trow_i.end(), trow_b.begin(), trow_b.end());
  tval_i.insert(tval_i.end(), tval_b.begin(), tval_b.end());
  ft.n_t = (int)t_interior.size();
  if (t_interior.empty()) { t_interior.push_back(0); trow_i.push_back(0); tval_i.push_back(0.f); }
  bool good = up((void**)&ft.d_tlist, t_interior.data(), t_interior.size() * 4) &&
              up((void**)&ft.d_tabrow, trow_i.data(), trow_i.size() * 4) &&
              up((void**)&ft.d_tabvals, tval_i.data(), tval_i.size() * 4) &&
              up((void**)&ft.d_tile_off, tile_off.data(), tile_off.size() * 4) &&
              up((void**)&ft.d_ring_end, ring_end.data(), ring_end.size() * 4) &&
              up((void**)&ft.d_ell_off, ell_off.data(), ell_off.size() * 8) &&
              up((void**)&ft.d_region, region.data(), region.size() * 4) &&
              up((void**)&ft.d_lcols, lcols.data(), lcols.size() * 2) &&
              up((void**)&ft.d_lvals, lvals.data(), lvals.size() * 4) &&
              up((void**)&ft.d_part, interior.data(), interior.size() * 4) &&
              up((void**)&ft.d_rlist, r_interior.data(), r_interior.size() * 4) &&
              up((void**)&ft.d_rrest, rrest.data(), rrest.size() * 4) &&
              up((void**)&ft.d_nonq, nonq.data(), nonq.size() * 4) &&
              up((void**)&ft.d_all, all_tiles.data(), all_tiles.size() * 4) &&
              up((void**)&ft.d_pairs, pairs.data(), pairs.size() * sizeof(StripPair)) &&
              up((void**)&ft.d_qstrips, qstrips.data(), qstrips.size() * sizeof(QStrip)) &&
              up((void**)&ft.d_qprefix, qprefix.data(), qprefix.size() * 4) &&
              up((void**)&ft.d_qtab, qtab.data(), qtab.size() * 4) &&
              up((void**)&ft.d_q8strips, q8strips.data(), q8strips.size() * sizeof(QStrip)) &&
              up((void**)&ft.d_q8prefix, q8prefix.data(), q8prefix.size() * 4) &&
              up((void**)&ft.d_q8tab, q8tab.data(), q8tab.size() * 4) &&
              up((void**)&ft.d_q8rest, q8rest.data(), q8rest.size() * 4) &&
              up((void**)&ft.d_qrrest, qrrest.data(), qrrest.size() * 4) &&
              up((void**)&ft.d_qtlist, qt_list.data(), qt_list.size() * 4) &&
              up((void**)&ft.d_qtabrow, qt_rows.data(), qt_rows.size() * 4) &&
              up((void**)&ft.d_qtabvals, qt_vals.data(), qt_vals.size() * 4) &&
              up((void**)&ft.d_ipairs, ipairs.data(), ipairs.size() * sizeof(StripPair));
  if (good && !patch_rows.empty()) good = struct_patch_rows(plan, fp->d_gvals8, fp->d_gdiag, patch_rows.data(), patch_vals.data(), (int64_t)patch_rows.size()) == DSPH_OK;
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
static int plane_rows_for(int rmax, int emax, int wt) {
  if (rmax <= 576 && emax <= 512) return 576;
  if (rmax <= 768 && emax <= 768) return 768;
  if (rmax <= 928 && emax <= 928) return 928;
  if (rmax <= 1024 && emax <= 1024) return 1024;
  return (rmax <= 1168 && emax <= 1024 && wt == 9) ? 1168 : 0;  // (K = 10 on the 8-neighbour grid: 34 x 34 rows, ELL rows on the inner 32 x 32; width 9 only)
}

// Layers the BFS-tile kernel runs with four maps per item (FusedArgs::pack): at most four input channels (padded to four),
// at most 16 output columns.  Its weight image then has two column blocks whatever the layer's width.
static int bfs_packs(int32_t Fin, int32_t Fout) { return Fin == 4 && Fout <= 16 ? 4 : (Fin == 8 && Fout <= 32 ? 2 : 0); }
static size_t wfrag_bytes(int32_t Fin, int32_t Fout, int32_t K) {
  const int C = (Fin + FUSED_CH - 1) / FUSED_CH, NB = bfs_packs(Fin, Fout) ? 2 : (Fout + 31) / 32;
  return (size_t)K * C * NB * 2048;
}

// weight images of the three fused kernels, back to back in the workspace: BFS-tile | structured-tile | strip
static size_t all_frag_bytes(int32_t Fin, int32_t Fout, int32_t K) {
  return wfrag_bytes(Fin, Fout, K) + struct_wfrag_bytes(Fin, Fout, K) + strip_wimg_bytes(Fin, Fout, K) +
         2 * istrip_wimg_bytes(K, DSPH_PREC_BF16X6) +  // (the largest of the three arithmetics, two 32-column blocks)
         (qstrip_shape_ok(Fin, Fout, K) ? qstrip_wimg_bytes() : 0) + (qstrip8_shape_ok(Fin, Fout, K) ? qstrip8_wimg_bytes() : 0);
}

// The structured-tile kernel addresses x by 32-bit byte offsets inside a map: larger maps take BFS tables throughout.
static bool want_full(const dsph_plan* plan, int32_t Fin, bool full) {
  return full || (uint64_t)plan->n_cols * (uint64_t)Fin * 4ull >= (1ull << 32);
}

static bool supported_impl(const dsph_plan* plan, int32_t Fin, int32_t Fout, int32_t K, bool full) {
  full = want_full(plan, Fin, full);
  if (!plan->fused || plan->fused->wide) return false;
  if (K < 2 || K - 1 > FUSED_DMAX) return false;
  if (Fin % 4 != 0 || Fin < 4 || Fout < 1) return false;  // Fout > 64: one launch per 64-column block
  const FusedTiles& ft = get_tiles(plan, K - 1, full);
  if (!ft.ok) return false;
  if (ft.n_r + ft.n_t > 0 && !struct_shape_ok(Fin, std::min(Fout, 64), K)) return false;
  if (ft.n_part == 0) return true;  // every tile is class R
  const int pr = plane_rows_for(ft.rmax, ft.emax, ft.width);
  if (pr == 0) return false;
  // (the weight fragments need not fit beside the planes: the forward then reads them from global memory, cheb_fused_kernel's
  // WG variant; the planes and weight-gradient modes hold no weights)
  return (size_t)2 * pr * FUSED_CH * 4 + FUSED_BIAS_BYTES <= (size_t)LDS_BYTES;
}

// The forward takes any Fin >= 1: channel counts that are not a multiple of four are zero-padded into the workspace first
// (fused_pad_kernel; the kernels load x in 16-byte pieces).  The first layer of every reference model has Fin = 1.
static inline int32_t pad4(int32_t Fin) { return (Fin + 3) & ~3; }

int fused_dmax() { return FUSED_DMAX; }
int fused_num_cu(const dsph_plan* plan) { return plan->fused ? plan->fused->num_cu : 256; }

// Depth-1 breadth-first tables of a whole graph too wide for the fused kernels: what the tiled step gathers from (cheb_tstep.hip)
bool fused_tstep_tables(const dsph_plan* plan, TStepTables* out) {
  if (!plan->fused || !plan->fused->wide || plan->n_cols != plan->n_rows || !plan->levels.empty()) return false;
  const FusedTiles& ft = get_tiles(plan, 1, true);
  if (!ft.ok || ft.rmax > 768 || ft.n_part != ft.ntiles) return false;  // (TS_RMAX of cheb_tstep.hip)
  out->tile_off = ft.d_tile_off;
  out->ring_end = ft.d_ring_end;
  out->ell_off = ft.d_ell_off;
  out->region = ft.d_region;
  out->lcols = ft.d_lcols;
  out->lvals = ft.d_lvals;
  out->ntiles = ft.ntiles;
  out->width = ft.width;
  return true;
}

bool fused_supported(const dsph_plan* plan, int32_t Fin, int32_t Fout, int32_t K) {
  return Fin >= 1 && supported_impl(plan, pad4(Fin), Fout, K, false);
}

// ... and with the BFS-tile kernel's weight fragments resident in the LDS (its fast variant: what "supported" meant before the
// WG variant existed; the K > 5 routing of dsphere_api.hip prefers the chain of passes to the WG variant)
bool fused_weights_resident(const dsph_plan* plan, int32_t Fin, int32_t Fout, int32_t K) {
  if (!fused_supported(plan, Fin, Fout, K)) return false;
  const int32_t Fp = pad4(Fin);
  const FusedTiles& ft = get_tiles(plan, K - 1, want_full(plan, Fp, false));
  if (ft.n_part == 0) return true;
  const int pr = plane_rows_for(ft.rmax, ft.emax, ft.width);
  return (size_t)2 * pr * FUSED_CH * 4 + wfrag_bytes(Fp, std::min(Fout, 64), K) + FUSED_BIAS_BYTES <= (size_t)LDS_BYTES;
}

static bool fused_symmetric(const dsph_plan* plan);
int fused_prepare(const dsph_plan* plan, int32_t K, int32_t Fin, int32_t flags) {
  if (plan->fused && plan->fused->wide) {  // the tiled step's tables, so that the first forward does not build them
    TStepTables tb;
    if (K >= 2) (void)fused_tstep_tables(plan, &tb);
    if (flags & DSPH_PREPARE_RELEASE_HOST) {
      FusedPlan* fpw = plan->fused;
      std::lock_guard<std::mutex> lock(fpw->mu);
      std::vector<int32_t>().swap(fpw->h_cols);
      std::vector<float>().swap(fpw->h_vals);
      fpw->host_released = true;
    }
    return DSPH_OK;
  }
  if (!plan->fused || K < 2 || K - 1 > FUSED_DMAX) return DSPH_OK;  // nothing to build: the unfused path serves it
  const FusedTiles& ftp = get_tiles(plan, K - 1, want_full(plan, pad4(std::max(Fin, 1)), false));
  if (ftp.ok && plan->opt.fork && ftp.n_r + ftp.n_t > 0 && ftp.n_part > 0) {  // a forward of this K may fork: the side stream exists before it
    std::lock_guard<std::mutex> lock(plan->fused->fork_mu);
    (void)side_stream_ready(plan, plan->fused, nullptr, true);
  }
  if (flags & DSPH_PREPARE_BACKWARD) {
    (void)get_tiles(plan, K - 1, true);
    (void)fused_symmetric(plan);  // (what the quad-strip weight gradient asks; the host arrays may be released below)
  }
  if (flags & DSPH_PREPARE_RELEASE_HOST) {
    FusedPlan* fp = plan->fused;
    std::lock_guard<std::mutex> lock(fp->mu);
    std::vector<int32_t>().swap(fp->h_cols);
    std::vector<float>().swap(fp->h_vals);
    fp->host_released = true;
  }
  return DSPH_OK;
}

// tiles of the K-term forward by kernel: class R (structured-tile kernel) and class G (BFS-tile kernel)
bool fused_tile_counts(const dsph_plan* plan, int32_t K, int64_t* n_struct, int64_t* n_bfs) {
  if (!plan->fused || K < 2 || K - 1 > FUSED_DMAX) return false;
  const FusedTiles& ft = get_tiles(plan, K - 1, false);
  if (!ft.ok) return false;
  *n_struct = ft.n_r + ft.n_t;
  *n_bfs = ft.n_part;
  return true;
}

// One rule for "does the strip kernel take this forward" (launch_fused_common and dsph_plan_strip_tiles): the shape is the
// kernel's, and the strips are worth it for this batch.  A strip step (48 output pixels of one row, one map) takes a workgroup
// ~3.0 us, a 256-pixel tile of one map takes the tile kernels ~18.7 us of a CU (both measured at the headline shape): in units
// of 0.1 us,   strips: (steps of the busiest workgroup, strip_makespan) x 30      tile kernels: tiles x N x 187 / CUs,
// with 3 % in favour of the tile kernels.  Small maps (fewer items than CUs) and ragged masks at small batches lose that
// comparison and keep their tiles on the tile kernels.  The rule depends on the batch and on the device's CU count, so the
// same map can be summed in two different orders at two batch sizes (both within the tolerance of the precision); a caller
// that needs batch- or shard-invariant bits fixes the choice per plan: dsph_plan_set_option(DSPH_OPT_STRIPS, 1 always | 2 never).
//   Fout: the columns of THIS launch (one 64-column block of the layer); ld: the layer's row stride of y.
static bool use_qstrips(const dsph_plan* plan, const FusedTiles& ft, int32_t Fin, int32_t Fout, int32_t K) {
  return plan->opt.strip_form == 0 && ft.n_qstrips > 0 && qstrip_shape_ok(Fin, Fout, K);
}
static bool strips_apply(const dsph_plan* plan, const FusedTiles& ft, int32_t Fin, int32_t Fout, int32_t K, int32_t precision, int64_t N,
                         int32_t ld) {
  const bool quad = use_qstrips(plan, ft, Fin, Fout, K);
  if (!((quad || ft.n_pairs > 0) && (precision == DSPH_PREC_BF16X3 || (precision == DSPH_PREC_F16X3 && quad)) &&
        strip_shape_ok(Fin, Fout, K) && ld % 4 == 0 && plan->n_cols * (int64_t)std::max(Fin, ld) * 4 < (1ll << 32)))
    return false;
  if (ft.strip_forced) return true;
  FusedPlan* fp = plan->fused;
  if (quad) {
    // (a quad-strip step takes 2.8 us, a tile-map 18.7; the tape of rows is cut evenly, so the span is a formula)
    if (N < 1) return false;
    const int64_t span = qstrip_split(fp->num_cu, ft.qtape_rows, N, ft.qtape_rows / std::max(1, ft.n_qstrips), nullptr, nullptr, nullptr);
    return span * 28 * 103 < ft.n_qstrip_tiles * N * 187 / fp->num_cu * 100;
  }
  const std::vector<int32_t>& steps = ft.strip_steps;
  if (N < 1 || (int64_t)steps.size() * N > (1ll << 24)) return false;
  int64_t span;
  {
    std::lock_guard<std::mutex> lock(fp->mu);
    auto it = ft.strip_span.find(N);
    if (it == ft.strip_span.end()) it = ft.strip_span.emplace(N, strip_makespan(steps, N, fp->num_cu)).first;
    span = it->second;
  }
  return span * 30 * 103 < ft.n_strip_tiles * N * 187 / fp->num_cu * 100;
}

// The K = 8 quad strips take their rectangles for a whole-map forward of their shape (32 -> 32, three-term bf16, Chebyshev basis,
// bias / ReLU epilogue) when the strips pay for the batch by the same kind of rule (a step 2.6 us, a tile on the breadth-first
// kernel 27.5 us of a CU); DSPH_OPT_STRIPS 1 / 2: always / never.
static bool q8_applies(const dsph_plan* plan, const FusedTiles& ft, int32_t Fin, int32_t Fout, int32_t K, int32_t precision, int64_t N,
                       int32_t ld) {
  // (no limit on the size of a map: this kernel forms its addresses in 64 bits -- configs[3] is 6.4 GB of x)
  // (ld == Fout: the layer IS 32 columns wide -- the last 32 columns of a wider layer would find no room for this kernel's weight
  // image in their block of the workspace, which is sized for 64-column blocks)
  if (ft.n_q8strips == 0 || !qstrip8_shape_ok(Fin, Fout, K) || (precision != DSPH_PREC_BF16X3 && precision != DSPH_PREC_F16X3) || ld != Fout || N < 1) return false;
  if (plan->opt.strips == 1) return true;
  const int64_t span = qstrip8_split(plan->fused->num_cu, ft.q8tape_rows, N, ft.q8tape_rows / std::max(1, ft.n_q8strips), nullptr, nullptr, nullptr);
  return span * 26 * 103 < ft.n_q8_tiles * N * 275 / plan->fused->num_cu * 100;
}

// The input-side strip kernel takes the rectangles of every layer with at most 16 input channels (any arithmetic, K = 2 .. 5,
// any output width), unless DSPH_OPT_STRIPS says never.  No cost rule: its workers are single waves and the kernel cuts the
// strips into as many row segments as the batch needs, so small maps fill the device too (istrip_segments).
static bool istrips_apply(const dsph_plan* plan, const FusedTiles& ft, int32_t Fin, int32_t K, int32_t Fout, int32_t ld) {
  return ft.n_ipairs > 0 && plan->opt.strips != 2 && istrip_shape_ok(Fin, K) && Fout % 4 == 0 && ld % 4 == 0;
}
static int istrip_nseg(const dsph_plan* plan, const FusedTiles& ft, int64_t N, int D, bool narrow) {
  FusedPlan* fp = plan->fused;
  std::lock_guard<std::mutex> lock(fp->mu);
  const int64_t key = 2 * N + (narrow ? 1 : 0);
  auto it = ft.iseg.find(key);
  if (it == ft.iseg.end()) it = ft.iseg.emplace(key, istrip_segments(ft.ipair_h, ft.ipair_second, N, fp->num_cu, D, narrow)).first;
  return it->second;
}

// conv + HealpyPool(p = 1) in one forward (launch_cheb_fused with a FusedPool): the input-side strip kernels, the structured
// kernel and the BFS-tile kernel store the pooled map themselves.  Every layer whose tiles those kernels take -- not the
// 64 -> 64 shape on maps large enough for the Clenshaw strips --, whole unsharded maps of whole tiles,
// a width that is a multiple of four, bias and ReLU only (the other activations run as a separate pass over the full-resolution map).
bool fused_pool_ok(const dsph_plan* plan, int64_t N, int32_t Fin, int32_t Fout, int32_t K, int32_t act) {
  if (!plan->fused || N < 1 || !(act == DSPH_ACT_NONE || act == DSPH_ACT_RELU)) return false;
  if (!plan->levels.empty() || plan->n_cols != plan->n_rows || plan->n_rows % FUSED_P != 0 || K < 2 || K - 1 > FUSED_DMAX) return false;
  if (!fused_supported(plan, Fin, Fout, K)) return false;
  const FusedTiles& ft = get_tiles(plan, K - 1, want_full(plan, pad4(Fin), false));
  if (!ft.ok || Fout % 4 != 0) return false;  // (pooled stores are 16 bytes wide)
  // the Clenshaw strip kernel has no pooled epilogue: not where it would take tiles (either precision could be asked for)
  return !strips_apply(plan, ft, pad4(Fin), std::min(Fout, 64), K, DSPH_PREC_BF16X3, N, Fout);
}


// tiles a forward of this shape hands to the strip kernel: the same predicate the launch uses, for the layer's first 64-column
// block (a layer with Fout = 96 runs its first block through the strips and reports them; one with Fout < 64 has none)
int64_t fused_strip_tiles(const dsph_plan* plan, int64_t N, int32_t Fin, int32_t Fout, int32_t K, int32_t precision) {
  if (!plan->fused || K < 2 || K - 1 > FUSED_DMAX) return 0;
  if (istrip_shape_ok(pad4(Fin), K)) {
    const FusedTiles& fti = get_tiles(plan, K - 1, want_full(plan, pad4(Fin), false));
    return fti.ok && istrips_apply(plan, fti, pad4(Fin), K, std::min(Fout, 64), Fout) ? fti.n_strip_tiles : 0;
  }
  if (qstrip8_shape_ok(Fin, Fout, K)) {
    const FusedTiles& ft8 = get_tiles(plan, K - 1, want_full(plan, Fin, false));
    return ft8.ok && q8_applies(plan, ft8, Fin, Fout, K, precision, N, Fout) ? ft8.n_q8_tiles : 0;
  }
  if (Fin != pad4(Fin) || Fout < 64) return 0;
  const FusedTiles& ft = get_tiles(plan, K - 1, want_full(plan, Fin, false));
  if (!ft.ok) return 0;
  return strips_apply(plan, ft, Fin, 64, K, precision, N, Fout) ? (use_qstrips(plan, ft, Fin, 64, K) ? ft.n_qstrip_tiles : ft.n_strip_tiles) : 0;
}

// the strip pairs of the K-term tables, 12 int32 each: x0[2], w[2], xs[2], y0, y1, xlo, xhi, ylo, yhi (StripPair); returns how
// many there are (also when cap is smaller), -1 when the plan has no fused tables for this K
int64_t fused_strip_pairs(const dsph_plan* plan, int32_t K, int32_t* out, int64_t cap) {
  if (!plan->fused || K < 2 || K - 1 > FUSED_DMAX) return -1;
  const FusedTiles& ft = get_tiles(plan, K - 1, false);
  if (!ft.ok) return -1;
  static_assert(sizeof(StripPair) == 12 * sizeof(int32_t), "StripPair is twelve int32");
  if (plan->opt.strip_form == 0 && ((K == 5 && !ft.h_qstrips.empty()) || (K == Q8_K && !ft.h_q8strips.empty()))) {  // the quad strips, in the same record: one strip, the second empty
    const std::vector<QStrip>& hq = K == Q8_K ? ft.h_q8strips : ft.h_qstrips;
    const int64_t n = (int64_t)hq.size();
    for (int64_t i = 0; i < n && i < cap; ++i) {
      const QStrip& q = hq[(size_t)i];
      const int32_t rec[12] = {q.x0, q.x0, q.w, 0, q.xs, q.xs, q.y0, q.y1, q.xlo, q.xhi, q.ylo, q.yhi};
      memcpy(out + 12 * i, rec, sizeof(rec));
    }
    return n;
  }
  const int64_t n = (int64_t)ft.h_pairs.size();
  for (int64_t i = 0; i < n && i < cap; ++i) memcpy(out + 12 * i, &ft.h_pairs[(size_t)i], sizeof(StripPair));
  return n;
}

// rows of n pixels of strip record `strip` (dsph_plan_strip_rows): through the rectangle's table for the quad strips, the
// virtual Z-order plane for the strip pairs; -1 when there is no such record
int64_t fused_strip_rows(const dsph_plan* plan, int32_t K, int64_t strip, int64_t n, const int32_t* xy, int64_t* rows) {
  if (!plan->fused || K < 2 || K - 1 > FUSED_DMAX) return -1;
  const FusedTiles& ft = get_tiles(plan, K - 1, false);
  if (!ft.ok || strip < 0) return -1;
  if (plan->opt.strip_form == 0 && ((K == 5 && !ft.h_qstrips.empty()) || (K == Q8_K && !ft.h_q8strips.empty()))) {
    const std::vector<QStrip>& hq = K == Q8_K ? ft.h_q8strips : ft.h_qstrips;
    const std::vector<int32_t>& htab = K == Q8_K ? ft.h_q8tab : ft.h_qtab;
    if (strip >= (int64_t)hq.size()) return -1;
    const QStrip& q = hq[(size_t)strip];
    // (K = 5: the kernel steps a row's bits without clamping it to the halo -- a run of steps reads rows ylo - 1 .. yhi + 6 of
    // the table's ring tiles, beyond the halo for nothing: they are answered as the kernel reads them)
    const int ya = K == 5 ? q.ylo - 1 : q.ylo, yb = K == 5 ? q.yhi + 6 : q.yhi;
    for (int64_t i = 0; i < n; ++i) {
      const int x = std::min(std::max(xy[2 * i], q.xlo), q.xhi), y = std::min(std::max(xy[2 * i + 1], ya), yb);
      rows[i] = (int64_t)htab[(size_t)(q.tab + (y >> 4) * q.tws + (x >> 4))] + (int64_t)st_morton((unsigned)x & 15u, (unsigned)y & 15u);
    }
    return n;
  }
  if (strip >= (int64_t)ft.h_pairs.size()) return -1;
  for (int64_t i = 0; i < n; ++i) rows[i] = (int64_t)st_morton((unsigned)xy[2 * i], (unsigned)xy[2 * i + 1]);
  return n;
}

// how a quad-strip forward of N maps on the K = 5 tables cuts its work (qstrip_split)
bool fused_strip_split(const dsph_plan* plan, int64_t N, int32_t* grid, int32_t* pieces, int32_t* wg_per_piece, int64_t* tape_rows) {
  if (!plan->fused || N < 1) return false;
  const FusedTiles& ft = get_tiles(plan, QS_D, false);
  if (!ft.ok || ft.n_qstrips == 0) return false;
  int g = 0, p = 0, w = 0;
  (void)qstrip_split(plan->fused->num_cu, ft.qtape_rows, N, ft.qtape_rows / std::max(1, ft.n_qstrips), &g, &p, &w);
  *grid = g; *pieces = p; *wg_per_piece = w; *tape_rows = ft.qtape_rows;
  return true;
}

// two fragment layouts: the BFS-tile kernel's and, behind it, the structured-tile kernel's
// (one fragment area per 64-column block of the layer, so that the packed images of ALL blocks survive the call:
// DSPH_FWD_KEEP_WEIGHTS)
static size_t frag_area_bytes(int32_t Fp, int32_t Fout, int32_t K) {
  return (size_t)((Fout + 63) / 64) * all_frag_bytes(Fp, std::min(Fout, 64), K);
}
size_t fused_workspace_bytes(const dsph_plan* plan, int64_t N, int32_t Fin, int32_t Fout, int32_t K, int32_t) {
  const int32_t Fp = pad4(Fin);
  const size_t frag = frag_area_bytes(Fp, Fout, K);
  return frag + (Fp != Fin ? (size_t)N * (size_t)plan->n_cols * (size_t)Fp * 4 : 0);  // + the zero-padded copy of x
}

__global__ __launch_bounds__(256) void fused_pad_kernel(const float* __restrict__ x, float4* __restrict__ out, int64_t rows,
                                                        int Fin, int Q) {  // out[r][q] <- x[r][4q .. 4q+3], zeros past Fin
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows * Q; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / Q;
    const int c = 4 * (int)(i - r * Q);
    const float* __restrict__ src = x + r * Fin + c;
    float4 v;
    v.x = src[0];
    v.y = c + 1 < Fin ? src[1] : 0.f;
    v.z = c + 2 < Fin ? src[2] : 0.f;
    v.w = c + 3 < Fin ? src[3] : 0.f;
    out[i] = v;
  }
}

// ------------------------------------------------------------------------------------------
// weight preparation + launch
// ------------------------------------------------------------------------------------------

// Weight fragments in MFMA operand order, one 2 KiB block per (order k, slice c, column block nb):
//   bf16x3: lane l, element j  <- w[(c*16 + 8*(l>>5) + j)*K + k][32*nb + (l&31)], hi at +0, lo at +1024
//   fp32  : step t, lane l     <- w[(c*16 + 8*(l>>5) + t)*K + k][32*nb + (l&31)] at t*256 + l*4
// pack (four maps per item, FusedArgs::pack): one slice, two column blocks, block diagonal -- inner index 4 q + c against columns
// 16 q + o holds w[c*K + k][o].
__global__ __launch_bounds__(256) void fused_wprep_kernel(const float* __restrict__ w,
                                                          unsigned char* __restrict__ out, int Fin,
                                                          int Fout, int K, int C, int NB, int prec, int ld, int pack) {
  const int blk = blockIdx.x;  // (k*C + c)*NB + nb
  const int nb = blk % NB, c = (blk / NB) % C, k = blk / (NB * C);
  for (int e = threadIdx.x; e < 512; e += 256) {
    const int l = e >> 3, j = e & 7;
    const int ch = c * FUSED_CH + 8 * (l >> 5) + j, col = 32 * nb + (l & 31);
    float v;
    if (pack) {  // pack = P maps: 16 / P inner indices and 64 / P columns each
      const int ci = 16 / pack, cm = 64 / pack;
      v = (ch / ci == col / cm && ch % ci < Fin && col % cm < Fout) ? w[((int64_t)(ch % ci) * K + k) * ld + col % cm] : 0.f;
    }
    else v = (ch < Fin && col < Fout) ? w[((int64_t)ch * K + k) * ld + col] : 0.f;
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

// Diagnostic build only (make ABLATE=1): DSPH_DBG_ONLY=s / b launches only the structured-tile / only the BFS-tile kernel
// (wrong results by construction: the other tiles of y stay unwritten).  The shipped library always launches both.
static inline bool dbg_only(char which) {
#ifdef DSPH_ABLATE
  const char* e = getenv("DSPH_DBG_ONLY");
  return e && e[0] == which;
#else
  (void)which;
  return false;
#endif
}

static int launch_fused_common(const dsph_plan* plan, const float* x, const float* w, const float* bias,
                               float* y, float* planes_out, int64_t N, int32_t Fin, int32_t Fout, int32_t K,
                               int32_t act, int32_t precision, float alpha_rest, float beta_rest,
                               void* workspace, size_t workspace_bytes, hipStream_t stream,
                               const float* dy = nullptr, float* dw = nullptr, int32_t ld = 0, int32_t part = 0,
                               int32_t Fin_w = 0, int32_t only = 0,  // only: 0 every launch, 1 the structured ones, 2 the BFS-tile one
                               bool keep_weights = false,            // the weight images in the workspace are those of an earlier call
                               const FusedPool* pool = nullptr,      // conv + pool: the strips store the pooled map (launch_cheb_fused)
                               const int32_t* wg_tiles = nullptr, int wg_ntiles = -1);  // weight-gradient mode on these tiles only

int launch_cheb_fused(const dsph_plan* plan, const float* x, const float* w, const float* bias,
                      float* y, int64_t N, int32_t Fin, int32_t Fout, int32_t K, int32_t act,
                      int32_t precision, float alpha_rest, float beta_rest, void* workspace,
                      size_t workspace_bytes, hipStream_t stream, int32_t part, bool keep_weights, const FusedPool* pool) {
  if (pool != nullptr && !(part == 0 && fused_pool_ok(plan, N, Fin, Fout, K, act) && pool->y != nullptr && (pool->type == 1 || pool->type == 2) &&
                           ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(pool->y)) & 15) == 0)) {
    set_error("cheb_fused: this plan / shape has no fused pooling (whole unsharded maps, no or ReLU activation, not the Clenshaw strips' shape)");
    return DSPH_E_UNSUPPORTED;
  }
  // more than 64 output columns: one launch per 64-column block (the recurrence is repeated; still one pass
  // over x per block instead of the unfused path's K planes through HBM)
  // The structured-tile kernel fuses bias and ReLU; with any other activation both fused kernels write the pre-activation
  // and one elementwise pass finishes y (only when class-R tiles exist: the BFS-tile kernel knows every activation).
  bool defer_act = false;
  if (act != DSPH_ACT_NONE && act != DSPH_ACT_RELU && K - 1 <= FUSED_DMAX && K >= 2) {
    const FusedTiles& ft = get_tiles(plan, K - 1, want_full(plan, pad4(Fin), false));
    defer_act = ft.ok && ft.n_r + ft.n_t > 0;
  }
  // (two-part launches: each part writes the pre-activation of its tiles and then finishes exactly those tiles' rows)
  // Fin not a multiple of four: a zero-padded copy of x behind the weight fragments in the workspace (with a two-part
  // launch both parts copy: the halo rows arrive between them)
  const int32_t Fin_w = Fin;
  if (Fin != pad4(Fin)) {
    const int32_t Fp = pad4(Fin);
    const size_t frag = frag_area_bytes(Fp, Fout, K);
    const size_t need = frag + (size_t)N * (size_t)plan->n_cols * (size_t)Fp * 4;
    if (!workspace || workspace_bytes < need) {
      set_error("cheb_fused: workspace %zu < %zu", workspace_bytes, need);
      return DSPH_E_WORKSPACE;
    }
    float* xp = reinterpret_cast<float*>(static_cast<unsigned char*>(workspace) + frag);
    const int64_t rows = N * plan->n_cols;
    const int64_t work = rows * (Fp / 4);
    hipLaunchKernelGGL(fused_pad_kernel, dim3((unsigned)std::min<int64_t>((work + 255) / 256, 65536)), dim3(256), 0, stream, x,
                       reinterpret_cast<float4*>(xp), rows, (int)Fin, (int)(Fp / 4));
    DSPH_HIP(hipGetLastError());
    x = xp;
    Fin = Fp;
  }
  // structured launches and the BFS-tile launch write disjoint tiles: when a forward has both, the latter goes to the plan's
  // side stream (FusedPlan::side), forked behind whatever the caller's stream holds so far and joined before this call returns
  FusedPlan* fp = plan->fused;
  bool fork = false;
  if (fp && plan->opt.fork && K >= 2 && K - 1 <= FUSED_DMAX) {
    const FusedTiles& ft = get_tiles(plan, K - 1, want_full(plan, Fin, false));
    const int ng = !ft.ok ? 0 : (part == 0 ? ft.n_part : (part == 1 ? ft.n_interior : ft.n_part - ft.n_interior));
    // (worth its two event operations only when the structured launches run for a while: two tile-maps per CU and more --
    // BASELINE configs[0], 168 tile-maps, is 13 us faster without it, configs[1] 29 us faster with it.  Under a stream
    // capture the fork and the join are edges of the graph and cost nothing at replay: every forward with both kinds of tiles
    // forks there, provided the side stream exists already -- dsph_plan_prepare made it)
    bool big = N * (int64_t)(ft.n_r + ft.n_t) >= 2 * (int64_t)fp->num_cu;
    if (!big && fp->side && stream != nullptr) {
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
      big = hipStreamIsCapturing(stream, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive;
    }
    fork = ft.ok && ft.n_r + ft.n_t > 0 && ng > 0 && big;
  }
  const size_t blk_frag = all_frag_bytes(Fin, std::min(Fout, 64), K);
  if (!workspace || workspace_bytes < frag_area_bytes(Fin, Fout, K)) {
    set_error("cheb_fused: workspace %zu < %zu", workspace_bytes, frag_area_bytes(Fin, Fout, K));
    return DSPH_E_WORKSPACE;
  }
  for (int32_t cb = 0; cb < Fout; cb += 64) {
    unsigned char* blk_ws = static_cast<unsigned char*>(workspace) + (size_t)(cb / 64) * blk_frag;  // this block's weight images
    fused_images_begin(plan, blk_ws, fused_images_key(Fin, Fin_w, std::min<int32_t>(64, Fout - cb), K, Fout, precision, beta_rest != 0.f,
                                                      N >= 2, plan->opt.pack), keep_weights);
    const FusedPool pool_blk{pool ? pool->y + cb : nullptr, pool ? pool->type : 0};  // (this block's columns of the pooled map)
    const FusedPool* pool_b = pool ? &pool_blk : nullptr;
    auto run = [&](hipStream_t st, int32_t only) {
      return launch_fused_common(plan, x, w + cb, bias ? bias + cb : nullptr, y + cb, nullptr, N, Fin,
                                 std::min<int32_t>(64, Fout - cb), K, defer_act ? DSPH_ACT_NONE : act, precision, alpha_rest,
                                 beta_rest, blk_ws, blk_frag, st, nullptr, nullptr, Fout, part, Fin_w, only, keep_weights, pool_b);
    };
    if (fork) {
      std::unique_lock<std::mutex> lock(fp->fork_mu);
      if (side_stream_ready(plan, fp, stream, true)) {
        DSPH_HIP(hipEventRecord(fp->ev_fork, stream));             // (nothing is on the side stream yet: a failure here or in the
        DSPH_HIP(hipStreamWaitEvent(fp->side, fp->ev_fork, 0));    //  next line leaves nothing to join)
        const int rc_b = run(fp->side, 2);
        const hipError_t e_rec = hipEventRecord(fp->ev_join, fp->side);
        const int rc_s = run(stream, 1);
        // the join happens whatever went wrong in between: the side stream never outlives the call.  (Without the join
        // event the only way to join is to wait for the side stream on the host -- not while the caller is capturing, where
        // a synchronisation would invalidate the capture: the error goes back instead, the capture is lost either way.)
        hipError_t e_join = hipSuccess;
        if (e_rec == hipSuccess) e_join = hipStreamWaitEvent(stream, fp->ev_join, 0);
        else {
          hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
          if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone) e_join = hipStreamSynchronize(fp->side);
        }
        if (rc_b != DSPH_OK || rc_s != DSPH_OK) fused_images_forget(plan, blk_ws);
        if (rc_b != DSPH_OK) return rc_b;
        if (rc_s != DSPH_OK) return rc_s;
        if (e_rec != hipSuccess) return hip_fail(e_rec, "hipEventRecord(join)");
        if (e_join != hipSuccess) return hip_fail(e_join, "join of the side stream");
        continue;
      }
    }
    const int rc = run(stream, 0);
    if (rc != DSPH_OK) { fused_images_forget(plan, blk_ws); return rc; }
  }
  if (pool != nullptr) {
    return DSPH_OK;  // (every kernel has stored its tiles pooled; y, the scratch of the C ABI, stays untouched)
  }
  if (defer_act) {
    const int64_t orows = plan->levels.empty() ? plan->n_rows : plan->levels[0];
    if (part == 0) return launch_struct_act(y, N * orows, Fout, Fout, act, stream);
    // a part finishes the rows of its own tiles only: INTERIOR and BOUNDARY can be issued in any order, repeated, or alone
    const FusedTiles& ft = get_tiles(plan, K - 1, want_full(plan, Fin, false));
    const int32_t* tl = part == 1 ? ft.d_all : ft.d_all + ft.n_all_interior;
    const int nt = part == 1 ? ft.n_all_interior : ft.n_all - ft.n_all_interior;
    return launch_struct_act_tiles(y, tl, nt, N, orows, Fout, Fout, act, stream);
  }
  return DSPH_OK;
}

// Planes mode of the same kernel: T_1 .. T_{K-1} of x, each (N, n_cols, Fin), valid on the plan's output rows.
bool fused_planes_supported(const dsph_plan* plan, int32_t Fin, int32_t K) { return supported_impl(plan, Fin, 1, K, true); }

int launch_cheb_fused_planes(const dsph_plan* plan, const float* x, float* planes_out, int64_t N, int32_t Fin,
                             int32_t K, float alpha_rest, float beta_rest, hipStream_t stream) {
  return launch_fused_common(plan, x, nullptr, nullptr, nullptr, planes_out, N, Fin, 1, K, DSPH_ACT_NONE,
                             DSPH_PREC_FP32, alpha_rest, beta_rest, nullptr, 0, stream);
}

// ---- weight-gradient mode ------------------------------------------------------------------------------
static int fused_grid(const dsph_plan* plan, const FusedTiles& ft, int ntiles = -1) {
  if (ntiles < 0) ntiles = ft.ntiles;
  return std::max(8, std::min(plan->fused->num_cu, (ntiles + 7) / 8 * 8));
}

// slices per launch: as many accumulator tiles (one per slice and order, 8 KiB each) as fit the LDS next to the planes
static int wgrad_slices_per_launch(const dsph_plan* plan, int32_t K) {
  const FusedTiles& ft = get_tiles(plan, K - 1, true);
  const int pr = plane_rows_for(ft.rmax, ft.emax, ft.width);
  if (pr == 0) return 0;
  const long freeb = (long)LDS_BYTES - 2L * pr * FUSED_CH * 4;
  return (int)(freeb / ((long)K * WG_TILE_BYTES));
}

bool fused_wgrad_supported(const dsph_plan* plan, int32_t Fin, int32_t Fout, int32_t K) {
  return supported_impl(plan, Fin, Fout, K, true) && wgrad_slices_per_launch(plan, K) >= 1;
}

size_t fused_wgrad_workspace_bytes(const dsph_plan* plan, int32_t Fin, int32_t Fout, int32_t K) {
  if (!fused_wgrad_supported(plan, Fin, Fout, K)) return 0;
  const FusedTiles& ft = get_tiles(plan, K - 1, true);
  const int C = (Fin + FUSED_CH - 1) / FUSED_CH;
  // (a slab per workgroup and pixel half; a small map's batch is split over up to num_cu workgroups in all: fused_wgrad_gy)
  return (size_t)2 * std::max(fused_grid(plan, ft), plan->fused->num_cu) * C * K * 16 * 64 * sizeof(float);
}

// dw[(f*K + k)*Fout + o] = sum over slabs, in a fixed order (deterministic): sixteen lanes per element, lane p sums the slabs
// p, p + 16, ... in ascending order, the sixteen partial sums are added pairwise in a fixed tree.  (One thread per element with
// a serial loop over 512 slabs took 0.14 ms at 16 -> 32: more than a seventh of that layer's weight gradient.)
// mirror_grid > 0 (the bf16 arithmetic): the slabs of the odd workgroups blockIdx.x of a grid of that many hold -dW
// (cheb_fused_kernel.h, weight-gradient mode) and are subtracted.
__global__ __launch_bounds__(256) void fused_wgrad_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw,
                                                                 int nslabs, int Fin, int Fout, int K, int C, int ld, int mirror_grid) {
  __shared__ float part[16][17];
  const int el = threadIdx.x & 15, p = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + el;
  const bool live = e < Fin * K * Fout;
  const int o = live ? e % Fout : 0, fk = live ? e / Fout : 0, k = fk % K, f = fk / K;
  const size_t slab = (size_t)C * K * 16 * 64;
  const size_t off = ((size_t)((f >> 4) * K + k) * 16 + (f & 15)) * 64 + o;
  float s = 0.f;
  if (live)
    for (int i = p; i < nslabs; i += 16) {
      const float v = slabs[(size_t)i * slab + off];
      s += (mirror_grid > 0 && (((i >> 1) % mirror_grid) & 1)) ? -v : v;
    }
  part[p][el] = s;
  __syncthreads();
  if (p == 0 && live) {
    float v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = part[q][el];
#pragma unroll
    for (int h = 8; h >= 1; h >>= 1)
#pragma unroll
      for (int q = 0; q < h; ++q) v[q] += v[q + h];
    dw[(size_t)fk * ld + o] = v[0];
  }
}

int launch_fused_pad(const float* x, float* xp, int64_t rows, int32_t Fin, int32_t Fp, hipStream_t stream) {
  const int64_t work = rows * (Fp / 4);
  hipLaunchKernelGGL(fused_pad_kernel, dim3((unsigned)std::min<int64_t>((work + 255) / 256, 65536)), dim3(256), 0, stream, x,
                     reinterpret_cast<float4*>(xp), rows, (int)Fin, (int)(Fp / 4));
  DSPH_HIP(hipGetLastError());
  return DSPH_OK;
}

int launch_cheb_fused_wgrad(const dsph_plan* plan, const float* x, const float* dy, float* dw, int64_t N,
                            int32_t Fin, int32_t Fout, int32_t K, int32_t precision, float alpha_rest, float beta_rest,
                            void* workspace, size_t workspace_bytes, hipStream_t stream, int32_t Fin_w) {
  if (!fused_wgrad_supported(plan, Fin, Fout, K)) {
    set_error("cheb_fused_wgrad: plan/shape not supported");
    return DSPH_E_UNSUPPORTED;
  }
  const size_t need = fused_wgrad_workspace_bytes(plan, Fin, Fout, K);
  if (!workspace || workspace_bytes < need) {
    set_error("cheb_fused_wgrad: workspace %zu < %zu", workspace_bytes, need);
    return DSPH_E_WORKSPACE;
  }
  for (int32_t cb = 0; cb < Fout; cb += 64) {
    const int rc = launch_fused_common(plan, x, nullptr, nullptr, static_cast<float*>(workspace), nullptr, N, Fin,
                                       std::min<int32_t>(64, Fout - cb), K, DSPH_ACT_NONE, precision, alpha_rest,
                                       beta_rest, nullptr, 0, stream, dy + cb, dw + cb, Fout, 0, Fin_w);
    if (rc != DSPH_OK) return rc;
  }
  return DSPH_OK;
}

// ---- quad-strip weight gradient (cheb_qwgrad.hip) -----------------------------------------------------------------------
// L~ equal to its transpose, entry for entry (to fp32 rounding): the product rule the quad-strip weight gradient stands on
// moves T_j from x to dy.
static bool fused_symmetric(const dsph_plan* plan) {
  FusedPlan* fp = plan->fused;
  std::lock_guard<std::mutex> lock(fp->mu);
  if (fp->symmetric >= 0) return fp->symmetric == 1;
  if (fp->host_released || plan->n_rows != plan->n_cols) return false;  // (not cached: unknown, or not a square operator)
  const int W = plan->width;
  const int64_t n = plan->n_rows;
  const int32_t* cols = fp->h_cols.data();
  const float* vals = fp->h_vals.data();
  bool sym = true;
  for (int64_t r = 0; r < n && sym; ++r)
    for (int j = 0; j < W; ++j) {
      const int64_t c = cols[r * W + j];
      const float v = vals[r * W + j];
      if (v == 0.f || c == r) continue;  // (padding entries carry zeros)
      if (c < 0 || c >= n) { sym = false; break; }
      // the entry (c, r) must hold the same value (every off-diagonal entry is checked from its own side)
      float back = 0.f;
      for (int i = 0; i < W; ++i)
        if (cols[c * W + i] == r) back += vals[c * W + i];
      // (to the last bits of fp32: a normalised Laplacian D^-1/2 A D^-1/2 evaluated in float64 and rounded may differ by one
      // unit in the last place between (r, c) and (c, r); that moves dW by 1e-7 of itself, far below the arithmetic's 4e-6)
      if (fabsf(back - v) > 2.4e-7f * fmaxf(fabsf(back), fabsf(v))) { sym = false; break; }
    }
  fp->symmetric = sym ? 1 : 0;
  return sym;
}

// The quad-strip weight gradient takes the strips' pixels of a K = 5, 64 -> 64 j layer in the three-term bf16 arithmetic when
// the forward of that shape would run on the quad strips (same tables, same cost rule) and L~ is symmetric; the BFS-tile
// kernel's weight-gradient mode takes the tiles the strips leave over.
bool fused_qwgrad_applies(const dsph_plan* plan, int64_t N, int32_t Fin, int32_t Fout, int32_t K, int32_t precision) {
  if (precision != DSPH_PREC_BF16X3 || !qwgrad_shape_ok(Fin, 64, K) || Fout % 64 != 0) return false;
  if (!plan->fused || plan->fused->wide || plan->n_rows != plan->n_cols || !plan->levels.empty()) return false;
  if (!fused_wgrad_supported(plan, Fin, Fout, K)) return false;
  const FusedTiles& ft = get_tiles(plan, K - 1, false);
  if (!ft.ok || !use_qstrips(plan, ft, Fin, 64, K) || !strips_apply(plan, ft, Fin, 64, K, DSPH_PREC_BF16X3, N, Fout)) return false;
  return fused_symmetric(plan);
}

size_t fused_qwgrad_workspace_bytes(const dsph_plan* plan) { return qwgrad_slab_bytes(plan->fused ? plan->fused->num_cu : 256); }

// workspace: [the BFS-tile kernel's slabs (fused_wgrad_workspace_bytes, 256-aligned) | the quad strips' slabs]
int launch_cheb_fused_qwgrad(const dsph_plan* plan, const float* x, const float* dy, float* dw, int64_t N, int32_t Fin, int32_t Fout,
                             int32_t K, float alpha_rest, float beta_rest, void* workspace, size_t bfs_slab_bytes, hipStream_t stream) {
  const FusedTiles& ft = get_tiles(plan, K - 1, false);
  for (int32_t cb = 0; cb < Fout; cb += 64) {
    if (ft.n_nonq > 0) {
      const int rc = launch_fused_common(plan, x, nullptr, nullptr, static_cast<float*>(workspace), nullptr, N, Fin, 64, K, DSPH_ACT_NONE,
                                         DSPH_PREC_BF16X3, alpha_rest, beta_rest, nullptr, 0, stream, dy + cb, dw + cb, Fout, 0, Fin, 0, false,
                                         nullptr, ft.d_nonq, ft.n_nonq);
      if (rc != DSPH_OK) return rc;
    }
    QWgradLaunch q;
    q.x = x; q.dy = dy + cb; q.dw = dw + cb;
    q.slabs = reinterpret_cast<float*>(static_cast<char*>(workspace) + bfs_slab_bytes);
    q.strips = ft.d_qstrips; q.prefix = ft.d_qprefix; q.tape_rows = ft.qtape_rows; q.tab = ft.d_qtab;
    q.gvals8 = plan->fused->d_gvals8; q.gdiag = plan->fused->d_gdiag;
    q.x_rows = plan->n_cols; q.dy_rows = plan->n_rows; q.N = N;
    q.nstrips = ft.n_qstrips; q.lddy = Fout; q.lddw = Fout; q.num_cu = plan->fused->num_cu;
    q.cheb = beta_rest != 0.f;
    q.accumulate = ft.n_nonq > 0;
    const int rc = launch_cheb_qwgrad(q, stream);
    if (rc != DSPH_OK) return rc;
  }
  return DSPH_OK;
}

static int launch_fused_common(const dsph_plan* plan, const float* x, const float* w, const float* bias,
                               float* y, float* planes_out, int64_t N, int32_t Fin, int32_t Fout, int32_t K,
                               int32_t act, int32_t precision, float alpha_rest, float beta_rest,
                               void* workspace, size_t workspace_bytes, hipStream_t stream, const float* dy,
                               float* dw, int32_t ld, int32_t part, int32_t Fin_w, int32_t only, bool keep_weights,
                               const FusedPool* pool, const int32_t* wg_tiles, int wg_ntiles) {
  // DSPH_PREC_F16X3 is the quad strips' arithmetic; every other kernel of the forward runs the six-term split (same accuracy)
  const bool f16 = precision == DSPH_PREC_F16X3;
  const int32_t strip_precision = precision;
  if (f16) precision = DSPH_PREC_BF16X6;
  if (ld <= 0) ld = Fout;
  if (Fin_w <= 0) Fin_w = Fin;  // channels of w; smaller than Fin when x is a zero-padded copy  // row stride of w, bias-less y / dy / dw: the layer's Fout when this is one column block
  const bool wgrad_mode = dy != nullptr;  // y then carries the slab workspace
  const bool planes_mode = planes_out != nullptr || wgrad_mode;
  if (!supported_impl(plan, Fin, Fout, K, planes_mode)) {
    set_error("cheb_fused: plan/shape not supported");
    return DSPH_E_UNSUPPORTED;
  }
  const FusedTiles& ft = get_tiles(plan, K - 1, want_full(plan, Fin, planes_mode));
  const size_t wb = planes_mode ? 0 : wfrag_bytes(Fin, Fout, K);
  const size_t wb_all = planes_mode ? 0 : all_frag_bytes(Fin, Fout, K);
  if (!planes_mode && (!workspace || workspace_bytes < wb_all)) {
    set_error("cheb_fused: workspace %zu < %zu", workspace_bytes, wb_all);
    return DSPH_E_WORKSPACE;
  }
  if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(workspace) & 15) ||
      (reinterpret_cast<uintptr_t>(planes_out) & 15) || (wgrad_mode && (reinterpret_cast<uintptr_t>(y) & 15))) {
    set_error("cheb_fused: x, workspace and planes must be 16-byte aligned");
    return DSPH_E_BADARG;
  }
  const int C = (Fin + FUSED_CH - 1) / FUSED_CH, NB = (Fout + 31) / 32;
  // ---- class-R and class-T tiles: the structured-tile kernel (forward only), without and with per-tile tables ----------
  if (!planes_mode && ft.n_r + ft.n_t > 0 && only != 2) {
    StructLaunch sl;
    sl.x = x; sl.w = w; sl.bias = bias; sl.y = y;
    sl.wfrag = static_cast<unsigned char*>(workspace) + wb;
    sl.gvals8 = plan->fused->d_gvals8;
    sl.gdiag = plan->fused->d_gdiag;
    sl.x_rows = plan->n_cols;
    sl.y_rows = plan->levels.empty() ? plan->n_rows : plan->levels[0];
    sl.N = N;
    sl.Fin = Fin; sl.Fin_w = Fin_w; sl.Fout = Fout; sl.K = K; sl.act = act; sl.precision = precision; sl.ld = ld;
    sl.num_cu = plan->fused->num_cu;
    sl.cheb = beta_rest != 0.f;
    bool struct_prep = true;  // the first structured launch of this call packs the fragments, if the block lacks them
    sl.allow_pack = plan->opt.pack;
    if (pool != nullptr) {
      sl.pool = pool->type;
      sl.ypool = pool->y;
      sl.ypool_rows = plan->n_rows / 4;
    }
    // rectangles of interior class-R tiles: the strip kernel, when it has this shape; the class-R list shrinks to the rest
    const bool strips = strips_apply(plan, ft, Fin, Fout, K, strip_precision, N, ld) && Fin_w == Fin && (reinterpret_cast<uintptr_t>(y) & 15) == 0;
    if (strips && part != 2 && !dbg_only('b') && use_qstrips(plan, ft, Fin, Fout, K)) {
      QStripLaunch qs;
      qs.x = x; qs.w = w; qs.bias = bias; qs.y = y;
      qs.wimg = static_cast<unsigned char*>(workspace) + wb + struct_wfrag_bytes(Fin, Fout, K) + strip_wimg_bytes(Fin, Fout, K) +
                2 * istrip_wimg_bytes(K, DSPH_PREC_BF16X6);
      qs.strips = ft.d_qstrips;
      qs.tab = ft.d_qtab;
      qs.prefix = ft.d_qprefix;
      qs.tape_rows = ft.qtape_rows;
      qs.gvals8 = plan->fused->d_gvals8;
      qs.gdiag = plan->fused->d_gdiag;
      qs.x_rows = sl.x_rows; qs.y_rows = sl.y_rows; qs.N = N;
      qs.nstrips = ft.n_qstrips; qs.Fin = Fin; qs.Fout = Fout; qs.act = act; qs.ld = ld;
      qs.num_cu = plan->fused->num_cu;
      qs.cheb = sl.cheb;
      qs.f16 = f16;
      qs.f16_xexp = plan->opt.f16_xexp;
      qs.prep_weights = fused_images_claim(plan, workspace, IMG_QSTRIP);
      const int rc = launch_cheb_qstrip(qs, stream);
      if (rc != DSPH_OK) return rc;
    } else if (strips && part != 2 && !dbg_only('b')) {
      StripLaunch st;
      st.x = x; st.w = w; st.bias = bias; st.y = y;
      st.wimg = static_cast<unsigned char*>(workspace) + wb + struct_wfrag_bytes(Fin, Fout, K);
      st.pairs = ft.d_pairs;
      st.gvals8 = plan->fused->d_gvals8;
      st.gdiag = plan->fused->d_gdiag;
      st.x_rows = sl.x_rows; st.y_rows = sl.y_rows; st.N = N;
      st.npairs = ft.n_pairs; st.Fin = Fin; st.Fout = Fout; st.K = K; st.act = act; st.precision = precision; st.ld = ld;
      st.num_cu = plan->fused->num_cu;
      st.cheb = sl.cheb;
      st.generic = plan->opt.strip_generic;
      st.prep_weights = fused_images_claim(plan, workspace, IMG_STRIP);
      const int rc = launch_cheb_strip(st, stream);
      if (rc != DSPH_OK) return rc;
    }
    const bool istrips = !strips && istrips_apply(plan, ft, Fin, K, Fout, ld) && (reinterpret_cast<uintptr_t>(y) & 15) == 0;
    if (istrips && part != 2 && !dbg_only('b')) {
      IStripLaunch is;
      is.x = x; is.w = w; is.bias = bias; is.y = y;
      is.wimg = static_cast<unsigned char*>(workspace) + wb + struct_wfrag_bytes(Fin, Fout, K) + strip_wimg_bytes(Fin, Fout, K);
      is.pairs = ft.d_ipairs;
      is.gvals8 = plan->fused->d_gvals8;
      is.gdiag = plan->fused->d_gdiag;
      is.x_rows = sl.x_rows; is.y_rows = sl.y_rows; is.N = N;
      is.npairs = ft.n_ipairs; is.Fin = Fin; is.Fin_w = Fin_w; is.Fout = Fout; is.K = K; is.act = act; is.precision = precision; is.ld = ld;
      is.num_cu = plan->fused->num_cu;
      // (row segments for the items the kernel will deal: maps, or pairs of maps in the one-channel kernel's pair mode)
      is.nseg = istrip_nseg(plan, ft, istrip_pairs(Fin_w, Fout) ? (N + 1) / 2 : N, K - 1, istrip_narrow(Fin_w));
      is.cheb = sl.cheb;
      is.prep_weights = fused_images_claim(plan, workspace, IMG_ISTRIP);
      if (pool != nullptr) {
        is.pool = pool->type;
        is.ypool = pool->y;
        is.ypool_rows = plan->n_rows / 4;
      }
      const int rc = launch_cheb_istrip(is, stream);
      if (rc != DSPH_OK) return rc;
    }
    const bool stripped = strips || istrips;
    const bool qrest = strips && use_qstrips(plan, ft, Fin, Fout, K);  // the quad strips took class-T tiles too: their own rest lists
    const int32_t* rl = qrest ? ft.d_qrrest : (stripped ? ft.d_rrest : ft.d_rlist);
    const int rl_n = qrest ? ft.n_qrrest : (stripped ? ft.n_rrest : ft.n_r), rl_ni = qrest ? ft.n_qrrest_interior : (stripped ? ft.n_rrest_interior : ft.n_r_interior);
    const int nr = part == 0 ? rl_n : (part == 1 ? rl_ni : rl_n - rl_ni);
    if (nr > 0 && !dbg_only('b')) {
      sl.tiles = part == 2 ? rl + rl_ni : rl;
      sl.tabrow = nullptr;
      sl.tabvals = nullptr;
      sl.ntiles = nr;
      sl.prep_weights = struct_prep && fused_images_claim(plan, workspace, IMG_STRUCT);
      struct_prep = false;
      const int rc = launch_cheb_struct(sl, stream);
      if (rc != DSPH_OK) return rc;
    }
    const int n_t = qrest ? ft.n_qt : ft.n_t, n_t_interior = qrest ? ft.n_qt_interior : ft.n_t_interior;
    const int nt = part == 0 ? n_t : (part == 1 ? n_t_interior : n_t - n_t_interior);
    if (nt > 0 && !dbg_only('b')) {
      const size_t first = part == 2 ? (size_t)n_t_interior : 0;
      sl.tiles = (qrest ? ft.d_qtlist : ft.d_tlist) + first;
      sl.tabrow = (qrest ? ft.d_qtabrow : ft.d_tabrow) + first * ST_CELLS;
      sl.tabvals = (qrest ? ft.d_qtabvals : ft.d_tabvals) + first * ST_CELLS * ST_TABV;
      sl.ntiles = nt;
      sl.prep_weights = struct_prep && fused_images_claim(plan, workspace, IMG_STRUCT);
      struct_prep = false;
      const int rc = launch_cheb_struct(sl, stream);
      if (rc != DSPH_OK) return rc;
    }
    const int ng = part == 0 ? ft.n_part : (part == 1 ? ft.n_interior : ft.n_part - ft.n_interior);
    if (ng == 0 || dbg_only('s')) return DSPH_OK;
  }
  if (only == 1) return DSPH_OK;
  // the BFS-tile kernel has two contraction arithmetics; the six-term split of the structured kernel is fp32-equivalent
  if (precision == DSPH_PREC_BF16X6) precision = DSPH_PREC_FP32;
  // four maps per item where the layer has at most four input channels and 16 output columns (FusedArgs::pack)
  const int pack = (plan->opt.pack && !planes_mode && !wgrad_mode && N >= 2) ? bfs_packs(Fin, Fout) : 0;  // (a single map gains nothing from two column blocks)
  const int NBb = pack ? 2 : NB;
  if (!planes_mode && fused_images_claim(plan, workspace, IMG_BFS)) {
    hipLaunchKernelGGL(fused_wprep_kernel, dim3(K * C * NBb), dim3(256), 0, stream, w,
                       static_cast<unsigned char*>(workspace), (int)Fin_w, (int)Fout, (int)K, C, NBb,
                       (int)precision, (int)ld, pack);
    DSPH_HIP(hipGetLastError());
  }

  // K = 8, 32 -> 32: the rectangles of depth-7 regular tiles on the quad strips, the rest of the tiles below
  const bool q8 = !planes_mode && part != 2 && pool == nullptr && only == 0 && Fin_w == Fin && beta_rest != 0.f &&
                  (act == DSPH_ACT_NONE || act == DSPH_ACT_RELU) && (reinterpret_cast<uintptr_t>(y) & 15) == 0 &&
                  q8_applies(plan, ft, Fin, Fout, K, strip_precision, N, ld);
  if (q8) {
    QStrip8Launch q;
    q.x = x; q.w = w; q.bias = bias; q.y = y;
    q.wimg = static_cast<unsigned char*>(workspace) + all_frag_bytes(Fin, Fout, K) - qstrip8_wimg_bytes();
    q.strips = ft.d_q8strips; q.tab = ft.d_q8tab; q.prefix = ft.d_q8prefix; q.tape_rows = ft.q8tape_rows;
    q.gvals8 = plan->fused->d_gvals8; q.gdiag = plan->fused->d_gdiag;
    q.x_rows = plan->n_cols; q.y_rows = plan->levels.empty() ? plan->n_rows : plan->levels[0]; q.N = N;
    q.nstrips = ft.n_q8strips; q.act = act; q.ld = ld; q.ld_w = ld; q.num_cu = plan->fused->num_cu;
    q.f16 = f16;
    q.f16_xexp = plan->opt.f16_xexp;
    q.prep_weights = fused_images_claim(plan, workspace, IMG_Q8);
    const int rc = launch_cheb_qstrip8(q, stream);
    if (rc != DSPH_OK) return rc;
    if ((part == 0 ? ft.n_q8rest : ft.n_q8rest_interior) == 0) return DSPH_OK;
  }

  FusedArgs args;
  args.planes_out = planes_out;
  args.dy = dy;
  args.slabs = wgrad_mode ? y : nullptr;
  args.c_begin = 0;
  args.c_count = C;
  args.prow_stride = plan->n_cols;
  args.plane_stride = N * plan->n_cols * (int64_t)Fin;
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
  args.N = pack ? (int)((N + pack - 1) / pack) : (int)N;  // (packed: groups of P maps)
  args.n_maps = (int)N;
  args.pack = pack;
  args.num_cu = plan->fused->num_cu;
  args.pool = pool ? pool->type : 0;
  args.ypool = pool ? pool->y : nullptr;
  args.ypool_rows = plan->n_rows / 4;
  args.Fin = Fin;
  args.Fout = Fout;
  args.ld = ld;
  args.K = K;
  // part: 0 all tiles, 1 interior tiles (no row of another rank in their region), 2 boundary tiles
  // the BFS-tile kernel handles the tiles of d_part (every tile of a full table, the class-G ones otherwise)
  args.tile_list = part == 0 ? (ft.n_part == ft.ntiles ? nullptr : ft.d_part) : (part == 1 ? ft.d_part : ft.d_part + ft.n_interior);
  args.ntiles = part == 0 ? ft.n_part : (part == 1 ? ft.n_interior : ft.n_part - ft.n_interior);
  if (q8) {  // (part 0: every tile the strips leave; part 1: the interior ones of them -- the boundary tiles are never the strips')
    args.tile_list = ft.d_q8rest;
    args.ntiles = part == 0 ? ft.n_q8rest : ft.n_q8rest_interior;
  }
  if (wgrad_mode && wg_ntiles >= 0) {  // (the strips' pixels go to cheb_qwgrad.hip: launch_cheb_fused_qwgrad)
    args.tile_list = wg_tiles;
    args.ntiles = wg_ntiles;
  }
  if (args.ntiles == 0) return DSPH_OK;
  args.nchunks = C;
  args.act = act;
  args.alpha_rest = alpha_rest;
  args.beta_rest = beta_rest;
  args.wfrag_bytes = (int)wb;
#ifdef DSPH_ABLATE  // diagnostic build only (make ABLATE=1): the shipped library never skips work
  const char* dbg = getenv("DSPH_FUSED_DEBUG");
  args.dbg = dbg ? atoi(dbg) : 0;
#else
  args.dbg = 0;
#endif
  const int pr = plane_rows_for(ft.rmax, ft.emax, ft.width);
  const size_t lds = (size_t)2 * pr * FUSED_CH * 4 + wb;
  const int grid = (wgrad_mode && wg_ntiles < 0) ? fused_grid(plan, ft) : fused_grid(plan, ft, args.ntiles);
  if (wgrad_mode) {
    // as many slices per launch as fit the wave's WG_TILES accumulator tiles; every launch runs the
    // recurrence for its own slices only, so the split costs nothing but a second read of dy
    auto launch_one = [&](const FusedArgs& la) -> int {
#define DSPH_FUSED_CASE(PR, WT) \
  if (pr == PR && ft.width == WT) return launch_fused_##PR##_##WT(la, NB, precision, grid, lds, stream);
      DSPH_FUSED_CASE(576, 9)
      DSPH_FUSED_CASE(768, 9)
      DSPH_FUSED_CASE(928, 9)
      DSPH_FUSED_CASE(1024, 9)
      DSPH_FUSED_CASE(1168, 9)
      DSPH_FUSED_CASE(576, 12)
      DSPH_FUSED_CASE(768, 12)
      DSPH_FUSED_CASE(928, 12)
      DSPH_FUSED_CASE(1024, 12)
#undef DSPH_FUSED_CASE
      set_error("cheb_fused: no kernel for plane rows %d, width %d", pr, ft.width);
      return DSPH_E_UNSUPPORTED;
    };
    const int per = wgrad_slices_per_launch(plan, K);
    for (int c0 = 0; c0 < C; c0 += per) {
      FusedArgs la = args;
      la.c_begin = c0;
      la.c_count = std::min(per, C - c0);
      const int rc = launch_one(la);
      if (rc != DSPH_OK) return rc;
    }
    // (Fin_w < Fin: x is a zero-padded copy, only the rows of the real channels exist in dw)
    const int total = Fin_w * K * Fout;
    hipLaunchKernelGGL(fused_wgrad_reduce_kernel, dim3((total + 15) / 16), dim3(256), 0, stream, args.slabs, dw,
                       2 * grid * fused_wgrad_gy(args.N, grid, args.num_cu), (int)Fin_w, (int)Fout, (int)K, C, (int)ld,
                       precision == DSPH_PREC_BF16X3 ? grid : 0);
    DSPH_HIP(hipGetLastError());
    return DSPH_OK;
  }
#ifdef DSPH_STAMPS
  static unsigned long long* d_stamps = nullptr;
  constexpr size_t NST = 8 * 8 * 32;
  if (!d_stamps) DSPH_HIP(hipMalloc(&d_stamps, NST * 8));
  DSPH_HIP(hipMemsetAsync(d_stamps, 0, NST * 8, stream));
  args.stamps = d_stamps;
  auto dump_stamps = [&](int rc) {
    if (rc != DSPH_OK || !getenv("DSPH_STAMPS_DUMP")) return rc;
    std::vector<unsigned long long> h(NST);
    if (hipStreamSynchronize(stream) != hipSuccess) return rc;
    if (hipMemcpy(h.data(), d_stamps, NST * 8, hipMemcpyDeviceToHost) != hipSuccess) return rc;
    for (int w = 0; w < 8; ++w)
      for (int it = 0; it < 8; ++it) {
        fprintf(stderr, "STAMP wave %d item %d:", w, it + 4);
        const unsigned long long* r = &h[((size_t)w * 8 + it) * 32];
        for (int i = 1; i < 32; ++i) fprintf(stderr, " %lld", r[i] && r[i - 1] ? (long long)(r[i] - r[i - 1]) : -1LL);
        fprintf(stderr, " | t0 %llu\n", r[0]);
      }
    return rc;
  };
#define DSPH_FUSED_CASE(PR, WT) \
  if (pr == PR && ft.width == WT) return dump_stamps(launch_fused_##PR##_##WT(args, NBb, precision, grid, lds, stream));
#else
#define DSPH_FUSED_CASE(PR, WT) \
  if (pr == PR && ft.width == WT) return launch_fused_##PR##_##WT(args, NBb, precision, grid, lds, stream);
#endif
  DSPH_FUSED_CASE(576, 9)
  DSPH_FUSED_CASE(768, 9)
  DSPH_FUSED_CASE(928, 9)
  DSPH_FUSED_CASE(1024, 9)
  DSPH_FUSED_CASE(1168, 9)
  DSPH_FUSED_CASE(576, 12)
  DSPH_FUSED_CASE(768, 12)
  DSPH_FUSED_CASE(928, 12)
  DSPH_FUSED_CASE(1024, 12)
#undef DSPH_FUSED_CASE
  set_error("cheb_fused: no kernel for plane rows %d, width %d", pr, ft.width);
  return DSPH_E_UNSUPPORTED;
}

}  // namespace dsph
