#!/usr/bin/env python3
"""Drives the host-side plan builders of libdsphere (tile classification, class-T embedding, BFS ring tables, rectangle merge,
strip / quad-strip lists, tape split, shard levels, option changes) through the C ABI of the SANITIZER build
(`make -C deepsphere-cosmo-tf2_amd/csrc asan` -> build_asan/libdsphere_asan.so: host code under AddressSanitizer + UBSan on a stub
HIP runtime, no GPU).  Run with the sanitizer runtime preloaded (tests/test_host.py does):

    LD_PRELOAD=$(hipcc -print-file-name=libclang_rt.asan-x86_64.so) ASAN_OPTIONS=detect_leaks=0 python3 tools/asan/run_plan_builders.py

Prints one line per plan and "ASAN-DRIVER-OK" at the end; any sanitizer report aborts the process (non-zero exit).
numpy / scipy only: torch is not imported (its allocator and the preloaded sanitizer runtime do not mix)."""
import ctypes
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "deepsphere-cosmo-tf2_amd", "deepsphere")
pkg = types.ModuleType("deepsphere")  # the package without its __init__ (which imports the torch layers)
pkg.__path__ = [PKG]
sys.modules["deepsphere"] = pkg
from deepsphere import healpix, utils  # noqa: E402

LIB = ctypes.CDLL(os.path.join(ROOT, "deepsphere-cosmo-tf2_amd", "csrc", "build_asan", "libdsphere_asan.so"))
i32, i64, vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p
LIB.dsph_plan_create.argtypes = [ctypes.POINTER(vp), i64, i64, i32, vp, vp, ctypes.c_int]
LIB.dsph_plan_destroy.argtypes = [vp]
LIB.dsph_plan_destroy.restype = None
LIB.dsph_plan_set_levels.argtypes = [vp, i32, vp]
LIB.dsph_plan_set_option.argtypes = [vp, i32, i64]
LIB.dsph_plan_prepare_layer.argtypes = [vp, i32, i32, i32, i32]
LIB.dsph_plan_tile_counts.argtypes = [vp, i32, ctypes.POINTER(i64), ctypes.POINTER(i64)]
LIB.dsph_plan_strip_tiles.argtypes = [vp, i64, i32, i32, i32, i32, ctypes.POINTER(i64)]
LIB.dsph_plan_strip_pairs.argtypes = [vp, i32, vp, i64, ctypes.POINTER(i64)]
LIB.dsph_plan_strip_split.argtypes = [vp, i64, ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(i64)]
LIB.dsph_plan_strip_rows.argtypes = [vp, i32, i64, i64, vp, vp]
LIB.dsph_plan_fused_ok.argtypes = [vp, i32, i32, i32]
LIB.dsph_workspace_bytes.argtypes = [vp, i64, i32, i32, i32, i32, i32]
LIB.dsph_workspace_bytes.restype = ctypes.c_size_t
LIB.dsph_backward_weights_workspace_bytes.argtypes = [vp, i64, i32, i32, i32, i32]
LIB.dsph_backward_weights_workspace_bytes.restype = ctypes.c_size_t
LIB.dsph_last_error.restype = ctypes.c_char_p


def ell_of(L, scale=0.75):
    Lt, _ = utils.prepare_L(L, scale=scale)
    cols, vals = utils.csr_to_ell(Lt)
    return np.ascontiguousarray(cols, np.int32), np.ascontiguousarray(vals, np.float32)


def plan_of(cols, vals, n_cols=None, levels=None, options=None):
    h = vp()
    rc = LIB.dsph_plan_create(ctypes.byref(h), cols.shape[0], cols.shape[0] if n_cols is None else n_cols, cols.shape[1],
                              cols.ctypes.data, vals.ctypes.data, 0)
    assert rc == 0, LIB.dsph_last_error()
    if levels is not None:
        lv = np.ascontiguousarray(levels, np.int64)
        assert LIB.dsph_plan_set_levels(h, len(lv), lv.ctypes.data) == 0, LIB.dsph_last_error()
    for k, v in (options or {}).items():
        assert LIB.dsph_plan_set_option(h, k, v) == 0, LIB.dsph_last_error()
    return h


def look(h, K, Fin, Fout, name, flags=1, N=(1, 3, 16)):
    assert LIB.dsph_plan_prepare_layer(h, K, Fin, Fout, flags) == 0, LIB.dsph_last_error()
    a, b, n = i64(0), i64(0), i64(0)
    rc = LIB.dsph_plan_tile_counts(h, K, ctypes.byref(a), ctypes.byref(b))
    ok = LIB.dsph_plan_fused_ok(h, Fin, Fout, K)
    strips = []
    for nn in N:
        t = i64(0)
        LIB.dsph_plan_strip_tiles(h, nn, Fin, Fout, K, 1, ctypes.byref(t))
        strips.append(t.value)
        LIB.dsph_workspace_bytes(h, nn, Fin, Fout, K, 1, 0)
        LIB.dsph_backward_weights_workspace_bytes(h, nn, Fin, Fout, K, 0)
        g, p, w, r = i32(0), i32(0), i32(0), i64(0)
        LIB.dsph_plan_strip_split(h, nn, ctypes.byref(g), ctypes.byref(p), ctypes.byref(w), ctypes.byref(r))
    LIB.dsph_plan_strip_pairs(h, K, None, 0, ctypes.byref(n))
    rec = np.zeros((max(n.value, 1), 12), np.int32)
    if n.value:
        assert LIB.dsph_plan_strip_pairs(h, K, rec.ctypes.data, n.value, ctypes.byref(n)) == 0
    print(f"{name}: K {K} {Fin}->{Fout}: tile_counts rc {rc} struct {a.value} bfs {b.value}, fused_ok {ok}, strip tiles {strips}, records {n.value}", flush=True)
    look.records = rec[: n.value]
    return a.value, b.value, strips


def check_strip_tables(h, K, cols, vals, name, D=4):
    """Every quad strip's table against the graph itself: a pixel's row, looked up through dsph_plan_strip_rows, must have its
    non-zeros of L~ exactly on the rows of the eight neighbouring pixels of the strip's plane (and itself) -- for every output
    pixel and every halo pixel a level is evaluated on (rings 0 .. D - 1 of the rectangle); output pixels are covered once."""
    rec = look.records
    seen = np.zeros(cols.shape[0], np.int32)
    for s, r in enumerate(rec):
        x0, w, y0, y1, xlo, xhi, ylo, yhi = int(r[0]), int(r[2]), int(r[6]), int(r[7]), int(r[8]), int(r[9]), int(r[10]), int(r[11])
        xa, xb = max(x0 - (D - 1), xlo + 1), min(x0 + w + (D - 1), xhi)  # evaluated columns of this strip (its neighbours cover the rest)
        ys = np.arange(ylo, yhi + 1)
        xs = np.arange(xa - 1, xb + 1)
        X, Y = np.meshgrid(xs, ys, indexing="ij")
        rows = np.zeros(X.size, np.int64)
        xy = np.ascontiguousarray(np.stack([X.ravel(), Y.ravel()], 1), np.int32)
        assert LIB.dsph_plan_strip_rows(h, K, s, xy.shape[0], xy.ctypes.data, rows.ctypes.data) == 0
        R = rows.reshape(X.shape)
        assert R.min() >= 0 and R.max() < cols.shape[0]
        inner = R[1:-1, 1:-1]                                          # rings 0 .. D - 1
        nb = np.stack([R[1 + dx: R.shape[0] - 1 + dx, 1 + dy: R.shape[1] - 1 + dy] for dx in (-1, 0, 1) for dy in (-1, 0, 1)], -1)
        c = cols[inner.ravel()]
        v = vals[inner.ravel()]
        ok = (v == 0) | (c[:, :, None] == nb.reshape(-1, 9)[:, None, :]).any(-1)
        assert ok.all(), f"{name}: strip {s}: a non-zero of L~ off the 3 x 3 window of the strip's plane"
        out = R[(x0 - (xa - 1)): (x0 - (xa - 1)) + w, (y0 - ylo): (y0 - ylo) + (y1 - y0)]
        np.add.at(seen, out.ravel(), 1)
        if K == 5:  # the rows a run of steps reads past the halo (the kernel does not clamp y): they must exist
            xs2, ys2 = np.arange(xlo, xhi + 1), np.concatenate([np.arange(ylo - 1, ylo), np.arange(yhi + 1, yhi + 7)])
            X2, Y2 = np.meshgrid(xs2, ys2, indexing="ij")
            xy2 = np.ascontiguousarray(np.stack([X2.ravel(), Y2.ravel()], 1), np.int32)
            rows2 = np.zeros(xy2.shape[0], np.int64)
            assert LIB.dsph_plan_strip_rows(h, K, s, xy2.shape[0], xy2.ctypes.data, rows2.ctypes.data) == 0
            assert rows2.min() >= 0 and rows2.max() < cols.shape[0], f"{name}: strip {s}: a row past the halo does not exist ({rows2.max()} of {cols.shape[0]})"
    assert seen.max() <= 1, f"{name}: an output pixel in two strips"
    print(f"{name}: {len(rec)} strips, {int(seen.sum())} output pixels = {int(seen.sum()) // 256} tiles, tables verified against the graph", flush=True)
    return int(seen.sum()) // 256


def shard_of(cols, vals, lo, hi, K):
    """Rows [lo, hi) of a graph as a plan with halo columns ordered by hop distance and a shrinking schedule (what
    deepsphere/sharding.ShardLayout builds, restated without torch)."""
    M, W = cols.shape
    dist = np.full(M, -1, np.int64)
    dist[lo:hi] = 0
    frontier = np.arange(lo, hi)
    order = [frontier]
    for d in range(1, K):
        nb = np.unique(cols[frontier][vals[frontier] != 0])
        nb = nb[dist[nb] < 0]
        dist[nb] = d
        order.append(nb)
        frontier = nb
    ids = np.concatenate(order)
    local = np.full(M, -1, np.int64)
    local[ids] = np.arange(ids.size)
    n_rows = int(sum(len(o) for o in order[: K - 1]))  # rows within K-2 hops carry a matrix row
    lc = local[cols[ids[:n_rows]]]
    lv = vals[ids[:n_rows]].copy()
    lv[lc < 0] = 0
    lc[lc < 0] = np.arange(n_rows)[:, None].repeat(W, 1)[lc < 0]
    levels = np.cumsum([len(o) for o in order[: K - 1]])
    return np.ascontiguousarray(lc, np.int32), np.ascontiguousarray(lv, np.float32), ids.size, levels


def main():
    OPT_STRIPS, OPT_STRUCT, OPT_TABLES, OPT_SPLIT, OPT_STRIP_FORM, OPT_MINROWS = 1, 2, 3, 8, 11, 6
    # full-sphere grid stencil: class R interiors (strips at 64 -> 64), class T face borders, class G corners
    for nside in (32, 64, 128):
        cols, vals = ell_of(healpix.healpix_laplacian(nside, mode="grid"))
        h = plan_of(cols, vals, options={OPT_STRIPS: 1})
        s, b, st = look(h, 5, 64, 64, f"grid nside {nside}")
        assert s + b == cols.shape[0] // 256 and b == 24
        if st[0]:
            assert check_strip_tables(h, 5, cols, vals, f"grid nside {nside}") == st[0]
        look(h, 5, 16, 32, f"grid nside {nside}")
        look(h, 3, 1, 16, f"grid nside {nside}")
        if nside == 128:  # the K = 8, 32 -> 32 quad strips: rectangles of the tiles whose 7-ring region stays inside a base pixel
            _, _, st8 = look(h, 8, 32, 32, f"grid nside {nside}")
            assert st8[0] >= 12 * 36 and check_strip_tables(h, 8, cols, vals, f"grid nside {nside} K 8", D=7) == st8[0]
        if nside == 64:
            look(h, 8, 32, 32, f"grid nside {nside}")          # 7-ring BFS tables
            look(h, 10, 16, 32, f"grid nside {nside}")         # the chain of passes
            for opt, v in ((OPT_STRIP_FORM, 1), (OPT_TABLES, 0), (OPT_STRUCT, 0), (OPT_SPLIT, 1), (OPT_MINROWS, 6)):
                assert LIB.dsph_plan_set_option(h, opt, v) == 0  # drops the tables: rebuilt below
                look(h, 5, 64, 64, f"grid nside {nside} option {opt}={v}")
            assert LIB.dsph_plan_set_option(h, 99, 0) != 0
        look(h, 5, 64, 64, f"grid nside {nside} release", flags=3)  # DSPH_PREPARE_RELEASE_HOST: further K fail cleanly
        assert LIB.dsph_plan_prepare_layer(h, 4, 64, 64, 0) in (0, -3)
        LIB.dsph_plan_destroy(h)
    # partial sky: a cap padded to superpixels (ragged rectangles, compacted rows: class T rings around every superpixel)
    for nside, sup in ((64, 8), (128, 8), (64, 2)):
        idx = healpix.extend_indices(healpix.cap_indices(nside, fraction=1.0 / 3.0), nside, sup)
        cols, vals = ell_of(healpix.healpix_laplacian(nside, indices=idx, mode="grid"))
        h = plan_of(cols, vals, options={OPT_STRIPS: 1})
        _, _, st = look(h, 5, 64, 64, f"cap nside {nside} superpixels {sup} ({cols.shape[0]} rows)")
        if st[0]:
            assert check_strip_tables(h, 5, cols, vals, f"cap nside {nside} superpixels {sup}") == st[0]
        LIB.dsph_plan_destroy(h)
    # the reference's graphs: k nearest neighbours (ELL width 11: BFS tiles; 23: the tiled step's depth-1 tables)
    for k in (8, 20):
        cols, vals = ell_of(healpix.healpix_laplacian(32, n_neighbors=k, mode="knn"))
        h = plan_of(cols, vals)
        look(h, 5, 16, 32, f"knn{k} nside 32 (width {cols.shape[1]})")
        LIB.dsph_plan_destroy(h)
    # a shard: half the sphere with its (K-1)-ring halo as trailing columns and a shrinking schedule
    cols, vals = ell_of(healpix.healpix_laplacian(64, mode="grid"))
    M = cols.shape[0]
    for K in (5, 8):
        for lo, hi in ((0, M // 2), (M // 4, M // 2), (M - M // 8, M)):
            lc, lv, n_cols, levels = shard_of(cols, vals, lo, hi, K)
            h = plan_of(lc, lv, n_cols=n_cols, levels=levels)
            look(h, K, 32, 32, f"shard rows [{lo}, {hi}) K {K}: {lc.shape[0]} rows, {n_cols} columns")
            LIB.dsph_plan_destroy(h)
    # a map whose last tile is incomplete, with strips: no rectangle may have that tile in its ring (the kernel reads rows of the
    # ring tiles past the halo)
    M = 12 * 64 * 64
    for drop in (100, 200, 256 + 37):
        cols, vals = ell_of(healpix.healpix_laplacian(64, indices=np.arange(M - drop), mode="grid"))
        h = plan_of(cols, vals, options={OPT_STRIPS: 1})
        _, _, st = look(h, 5, 64, 64, f"grid nside 64 without its last {drop} pixels ({cols.shape[0]} rows)")
        assert st[0] > 0 and check_strip_tables(h, 5, cols, vals, f"grid nside 64 without its last {drop} pixels") == st[0]
        LIB.dsph_plan_destroy(h)
    # a ragged tail (rows not a multiple of 256) and a tiny graph
    cols, vals = ell_of(healpix.healpix_laplacian(8, mode="grid"))
    h = plan_of(cols[:700].clip(max=699), vals[:700])
    look(h, 5, 4, 8, "ragged 700 rows")
    LIB.dsph_plan_destroy(h)
    print("ASAN-DRIVER-OK", flush=True)


if __name__ == "__main__":
    main()
