"""ctypes binding of ``libdsphere_hip.so`` (the C ABI declared in ``include/dsphere.h``).

There is no CPU fallback: if the shared library is missing, or a forward is requested
without a HIP device, this module raises.  Build the library with
``make -C deepsphere-cosmo-tf2_amd/csrc`` (or ``__graft_entry__.build()``).
"""

import ctypes
import os

import numpy as np

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_lib", "libdsphere_hip.so")
_lib = None

OK = 0
ACT_NONE, ACT_RELU, ACT_ELU, ACT_SIGMOID, ACT_TANH = 0, 1, 2, 3, 4
PREC_FP32, PREC_BF16X3, PREC_BF16X6, PREC_F16X3 = 0, 1, 2, 3
ALGO_AUTO, ALGO_UNFUSED, ALGO_FUSED = 0, 1, 2
PART_ALL, PART_INTERIOR, PART_BOUNDARY = 0, 1, 2
FWD_KEEP_WEIGHTS = 1
ABI_VERSION = 3  # DSPH_ABI_VERSION of include/dsphere.h this binding was written against
POOL_MAX, POOL_AVG = 0, 1
PREPARE_BACKWARD, PREPARE_RELEASE_HOST = 1, 2
BASIS_CHEBYSHEV, BASIS_MONOMIAL = 0, 1
# dsph_plan_set_option (include/dsphere.h: DSPH_OPT_*)
OPT_STRIPS, OPT_STRUCT, OPT_TABLES, OPT_FORK, OPT_STRIP_SEG, OPT_STRIP_MINROWS, OPT_STRIP_GENERIC, OPT_SPLIT, OPT_TSTEP = 1, 2, 3, 4, 5, 6, 7, 8, 9
OPT_PACK = 10
OPT_STRIP_FORM = 11
OPT_F16_XEXP = 12
STRIP_FORM_QUAD, STRIP_FORM_PAIRS = 0, 1
STRIPS_AUTO, STRIPS_ALWAYS, STRIPS_NEVER = 0, 1, 2
SPLIT_AUTO, SPLIT_ALWAYS, SPLIT_NEVER = 0, 1, 2

_c_i64 = ctypes.c_int64
_c_i32 = ctypes.c_int32
_c_vp = ctypes.c_void_p

# name -> (restype, argtypes); one entry per function declared in include/dsphere.h
SIGNATURES = {
    "dsph_abi_version": (ctypes.c_int, []),
    "dsph_last_error": (ctypes.c_char_p, []),
    "dsph_plan_create": (ctypes.c_int, [ctypes.POINTER(_c_vp), _c_i64, _c_i64, _c_i32, _c_vp, _c_vp, ctypes.c_int]),
    "dsph_plan_destroy": (None, [_c_vp]),
    "dsph_plan_set_levels": (ctypes.c_int, [_c_vp, _c_i32, _c_vp]),
    "dsph_plan_set_option": (ctypes.c_int, [_c_vp, _c_i32, _c_i64]),
    "dsph_plan_strip_pairs": (ctypes.c_int, [_c_vp, _c_i32, _c_vp, _c_i64, ctypes.POINTER(_c_i64)]),
    "dsph_plan_strip_split": (ctypes.c_int, [_c_vp, _c_i64, ctypes.POINTER(_c_i32), ctypes.POINTER(_c_i32), ctypes.POINTER(_c_i32),
                                             ctypes.POINTER(_c_i64)]),
    "dsph_plan_strip_rows": (ctypes.c_int, [_c_vp, _c_i32, _c_i64, _c_i64, _c_vp, _c_vp]),
    "dsph_plan_rows": (_c_i64, [_c_vp]),
    "dsph_plan_cols": (_c_i64, [_c_vp]),
    "dsph_plan_ell_width": (_c_i32, [_c_vp]),
    "dsph_plan_out_rows": (_c_i64, [_c_vp, _c_i32]),
    "dsph_plan_fused_ok": (ctypes.c_int, [_c_vp, _c_i32, _c_i32, _c_i32]),
    "dsph_plan_uses_chain": (ctypes.c_int, [_c_vp, _c_i32, _c_i32, _c_i32]),
    "dsph_plan_prepare": (ctypes.c_int, [_c_vp, _c_i32, _c_i32, _c_i32]),
    "dsph_plan_prepare_layer": (ctypes.c_int, [_c_vp, _c_i32, _c_i32, _c_i32, _c_i32]),
    "dsph_plan_tile_counts": (ctypes.c_int, [_c_vp, _c_i32, ctypes.POINTER(_c_i64), ctypes.POINTER(_c_i64)]),
    "dsph_plan_strip_tiles": (ctypes.c_int, [_c_vp, _c_i64, _c_i32, _c_i32, _c_i32, _c_i32, ctypes.POINTER(_c_i64)]),
    "dsph_workspace_bytes": (ctypes.c_size_t, [_c_vp, _c_i64, _c_i32, _c_i32, _c_i32, _c_i32, _c_i32]),
    "dsph_cheb_forward": (
        ctypes.c_int,
        [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i32, _c_i32, _c_i32, _c_i32, _c_i32, _c_i32, _c_vp,
         ctypes.c_size_t, _c_vp],
    ),
    "dsph_poly_forward": (
        ctypes.c_int,
        [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i32, _c_i32, _c_i32, _c_i32, _c_i32, _c_i32, _c_i32, _c_vp,
         ctypes.c_size_t, _c_vp],
    ),
    "dsph_poly_forward_part": (
        ctypes.c_int,
        [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i32, _c_i32, _c_i32, _c_i32, _c_i32, _c_i32, _c_i32, _c_i32,
         _c_vp, ctypes.c_size_t, _c_vp],
    ),
    "dsph_poly_forward_ex": (
        ctypes.c_int,
        [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i32, _c_i32, _c_i32, _c_i32, _c_i32, _c_i32, _c_i32, _c_i32, _c_i32,
         _c_vp, ctypes.c_size_t, _c_vp],
    ),
    "dsph_cheb_step": (
        ctypes.c_int,
        [_c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i32, ctypes.c_float, ctypes.c_float, _c_i64, _c_vp],
    ),
    "dsph_cheb_contract": (
        ctypes.c_int,
        [_c_vp, _c_i64, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i32, _c_i32, _c_i32, _c_i32, _c_i32, ctypes.c_int,
         _c_vp],
    ),
    "dsph_wgrad_workspace_bytes": (ctypes.c_size_t, [_c_i64, _c_i64, _c_i32, _c_i32, _c_i32]),
    "dsph_cheb_planes": (
        ctypes.c_int,
        [_c_vp, _c_vp, _c_vp, _c_i64, _c_i32, _c_i32, _c_i32, _c_i32, _c_vp],
    ),
    "dsph_backward_weights_workspace_bytes": (ctypes.c_size_t, [_c_vp, _c_i64, _c_i32, _c_i32, _c_i32, _c_i32]),
    "dsph_cheb_backward_weights": (
        ctypes.c_int,
        [_c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i32, _c_i32, _c_i32, _c_i32, _c_i32, _c_i32, _c_vp, ctypes.c_size_t,
         _c_vp],
    ),
    "dsph_cheb_wgrad": (
        ctypes.c_int,
        [_c_vp, _c_i64, _c_vp, _c_vp, _c_i64, _c_i64, _c_i32, _c_i32, _c_i32, _c_vp, ctypes.c_size_t, ctypes.c_int,
         _c_vp],
    ),
    "dsph_rows_pack": (ctypes.c_int, [_c_vp, _c_i64, _c_vp, _c_i64, _c_vp, _c_i64, _c_i32, ctypes.c_int, _c_vp]),
    "dsph_rows_unpack": (ctypes.c_int, [_c_vp, _c_i64, _c_vp, _c_i64, _c_vp, _c_i64, _c_i32, ctypes.c_int, _c_vp]),
    "dsph_plan_pool_fusable": (ctypes.c_int, [_c_vp, _c_i64, _c_i32, _c_i32, _c_i32, _c_i32]),
    "dsph_poly_forward_pool": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i32, _c_i32, _c_i32, _c_i32, _c_i32,
                                               _c_i32, _c_i32, _c_i32, _c_vp, ctypes.c_size_t, _c_vp]),
    "dsph_healpix_pool": (ctypes.c_int, [_c_vp, _c_vp, _c_i64, _c_i64, _c_i32, _c_i32, _c_i32, ctypes.c_int, _c_vp]),
    "dsph_healpix_pool_backward": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i32, _c_i32, _c_i32, ctypes.c_int, _c_vp]),
    "dsph_residual_epilogue": (ctypes.c_int, [_c_vp, _c_vp, _c_i64, ctypes.c_float, _c_i32, _c_i32, ctypes.c_int, _c_vp]),
}


def library_path():
    return _LIB_PATH


def lib():
    """The loaded library; raises RuntimeError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(
                f"HIP library not found at {_LIB_PATH}: build it with "
                "`make -C deepsphere-cosmo-tf2_amd/csrc` (there is no CPU fallback)"
            )
        handle = ctypes.CDLL(_LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if handle.dsph_abi_version() != ABI_VERSION:
            raise RuntimeError("libdsphere_hip.so has an unexpected ABI version")
        _lib = handle
    return _lib


def last_error():
    msg = lib().dsph_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc, what):
    if rc == OK:
        return
    msg = f"{what} failed ({rc}): {last_error()}"
    if rc == -1:
        raise ValueError(msg)
    raise RuntimeError(msg)


def require_gpu():
    import torch

    if not torch.cuda.is_available():
        raise RuntimeError(
            "the Chebyshev forward runs only on a HIP device (MI355X); no GPU is visible and there is no CPU fallback"
        )


class LaplacianPlan:
    """Owner of one ``dsph_plan``: the rescaled Laplacian, padded ELL, resident on one GPU."""

    def __init__(self, ell_cols, ell_vals, n_cols=None, device=0, levels=None, options=None):
        cols = np.ascontiguousarray(ell_cols, dtype=np.int32)
        vals = np.ascontiguousarray(ell_vals, dtype=np.float32)
        if cols.ndim != 2 or cols.shape != vals.shape:
            raise ValueError("ELL cols/vals must be 2-D arrays of equal shape [rows, width]")
        self.n_rows, self.width = int(cols.shape[0]), int(cols.shape[1])
        self.n_cols = int(n_cols) if n_cols is not None else self.n_rows
        self.device = int(device)
        self._h = _c_vp()
        require_gpu()
        rc = lib().dsph_plan_create(
            ctypes.byref(self._h), self.n_rows, self.n_cols, self.width, cols.ctypes.data, vals.ctypes.data,
            self.device,
        )
        check(rc, "dsph_plan_create")
        self.levels = None
        if levels is not None:
            self.set_levels(levels)
        for opt, value in (options or {}).items():
            self.set_option(opt, value)

    def set_option(self, option, value):
        """``dsph_plan_set_option``: a per-plan choice (OPT_*), to be made before the tables of a K are built."""
        check(lib().dsph_plan_set_option(self.handle, int(option), int(value)), "dsph_plan_set_option")

    def set_levels(self, levels):
        lv = np.ascontiguousarray(levels, dtype=np.int64)
        check(lib().dsph_plan_set_levels(self._h, int(lv.shape[0]), lv.ctypes.data), "dsph_plan_set_levels")
        self.levels = lv.copy()

    @property
    def handle(self):
        if not self._h:
            raise RuntimeError("plan already destroyed")
        return self._h

    @property
    def out_rows(self):
        return int(lib().dsph_plan_out_rows(self.handle, 1))

    def fused_ok(self, Fin, Fout, K):
        return bool(lib().dsph_plan_fused_ok(self.handle, int(Fin), int(Fout), int(K)))

    def uses_chain(self, Fin, Fout, K):
        """``dsph_plan_uses_chain``: does a forward of this shape run as the chain of <= 5-term passes (K > 5)."""
        return bool(lib().dsph_plan_uses_chain(self.handle, int(Fin), int(Fout), int(K)))

    def prepare(self, K, Fin, backward=False, release_host=False, Fout=None):
        """Build the fused kernels' tables for a K-term layer now (``dsph_plan_prepare_layer``; without ``Fout`` the width is
        taken as ``Fin``): afterwards a forward neither allocates nor synchronises, so it can be timed and captured into a graph."""
        flags = (PREPARE_BACKWARD if backward else 0) | (PREPARE_RELEASE_HOST if release_host else 0)
        check(lib().dsph_plan_prepare_layer(self.handle, int(K), int(Fin), int(Fin if Fout is None else Fout), flags),
              "dsph_plan_prepare_layer")

    def tile_counts(self, K):
        """(tiles run by the structured-tile kernel, tiles run by the BFS-tile kernel) of a K-term fused forward."""
        a, b = _c_i64(0), _c_i64(0)
        check(lib().dsph_plan_tile_counts(self.handle, int(K), ctypes.byref(a), ctypes.byref(b)), "dsph_plan_tile_counts")
        return int(a.value), int(b.value)

    def strip_tiles(self, Fin, Fout, K, precision=PREC_BF16X3, N=1):
        """How many of the structured tiles a fused forward of this shape, on a batch of N maps, hands to the strip kernel
        (``dsph_plan_strip_tiles``)."""
        n = _c_i64(0)
        check(lib().dsph_plan_strip_tiles(self.handle, int(N), int(Fin), int(Fout), int(K), int(precision), ctypes.byref(n)),
              "dsph_plan_strip_tiles")
        return int(n.value)

    def strip_pairs(self, K):
        """The strip kernel's work list for K terms (``dsph_plan_strip_pairs``): an int32 array [n_pairs, 12] of
        x0[2], w[2], xs[2], y0, y1, xlo, xhi, ylo, yhi in the virtual Z-order plane of the row index."""
        n = _c_i64(0)
        check(lib().dsph_plan_strip_pairs(self.handle, int(K), _c_vp(), 0, ctypes.byref(n)), "dsph_plan_strip_pairs")
        out = np.zeros((int(n.value), 12), dtype=np.int32)
        if n.value:
            check(lib().dsph_plan_strip_pairs(self.handle, int(K), out.ctypes.data, int(n.value), ctypes.byref(n)),
                  "dsph_plan_strip_pairs")
        return out

    def strip_rows(self, K, strip, xs, ys):
        """Row numbers of the pixels (xs[i], ys[i]) of the plane of strip record ``strip`` (``dsph_plan_strip_rows``): through the
        rectangle's table of tile bases for the quad strips, the Z-order plane of the row index for the strip pairs."""
        xy = np.ascontiguousarray(np.stack([np.asarray(xs), np.asarray(ys)], axis=1), dtype=np.int32)
        rows = np.zeros(xy.shape[0], dtype=np.int64)
        check(lib().dsph_plan_strip_rows(self.handle, int(K), int(strip), int(xy.shape[0]), xy.ctypes.data, rows.ctypes.data),
              "dsph_plan_strip_rows")
        return rows

    def strip_split(self, N):
        """How a quad-strip forward of ``N`` maps cuts its work (``dsph_plan_strip_split``): (grid, pieces, workgroups per piece,
        rows of the tape of one map)."""
        g, p, w, r = _c_i32(0), _c_i32(0), _c_i32(0), _c_i64(0)
        check(lib().dsph_plan_strip_split(self.handle, int(N), ctypes.byref(g), ctypes.byref(p), ctypes.byref(w), ctypes.byref(r)),
              "dsph_plan_strip_split")
        return int(g.value), int(p.value), int(w.value), int(r.value)

    def workspace_bytes(self, N, Fin, Fout, K, precision=PREC_FP32, algo=ALGO_AUTO):
        return int(lib().dsph_workspace_bytes(self.handle, int(N), int(Fin), int(Fout), int(K), int(precision),
                                              int(algo)))

    def close(self):
        if getattr(self, "_h", None):
            lib().dsph_plan_destroy(self._h)
            self._h = _c_vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _ptr(t):
    return _c_vp(t.data_ptr()) if t is not None else _c_vp()


def _stream_ptr(device):
    import torch

    return _c_vp(torch.cuda.current_stream(device).cuda_stream)


def _check_dev(t, plan, name):
    import torch

    if not t.is_cuda or t.device.index != plan.device:
        raise ValueError(f"{name} must live on cuda:{plan.device}")
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise ValueError(f"{name} must be a contiguous float32 tensor")


def cheb_forward(plan, x, w, bias, K, act=ACT_NONE, precision=PREC_FP32, algo=ALGO_AUTO, workspace=None, out=None,
                 basis=BASIS_CHEBYSHEV, part=PART_ALL, keep_weights=False):
    """y = dsph_poly_forward_ex(...) on torch CUDA tensors; x (N, n_cols, Fin), w (Fin*K, Fout).
    ``part``: PART_ALL, or PART_INTERIOR / PART_BOUNDARY (fused kernel only) to write only the tiles that do not /
    do touch halo rows -- pass the same ``out`` to both calls.
    ``keep_weights``: the caller vouches that ``workspace`` was last used by a call with the same weight values, shape,
    basis and precision (DSPH_FWD_KEEP_WEIGHTS: no weight-preparation launches); dropped when the workspace is replaced."""
    import torch

    _check_dev(x, plan, "x")
    _check_dev(w, plan, "w")
    N, rows, Fin = x.shape
    if rows != plan.n_cols:
        raise ValueError(f"x has {rows} rows, the plan multiplies vectors of {plan.n_cols} rows")
    if w.shape[0] != Fin * K:
        raise ValueError(f"w has {w.shape[0]} rows, expected Fin*K = {Fin * K}")
    Fout = int(w.shape[1])
    if bias is not None:
        _check_dev(bias, plan, "bias")
        if bias.numel() != Fout:
            raise ValueError("bias must have Fout elements")
    need = plan.workspace_bytes(N, Fin, Fout, K, precision, algo)
    if need > 0 and (workspace is None or workspace.numel() * workspace.element_size() < need):
        workspace = torch.empty(need, dtype=torch.uint8, device=x.device)
        keep_weights = False
    orows = plan.out_rows
    if out is None:
        out = torch.empty((N, orows, Fout), dtype=torch.float32, device=x.device)
    else:
        _check_dev(out, plan, "out")
        if tuple(out.shape) != (N, orows, Fout):
            raise ValueError("out has the wrong shape")
    rc = lib().dsph_poly_forward_ex(
        plan.handle, _ptr(x), _ptr(w), _ptr(bias), _ptr(out), int(N), int(Fin), Fout, int(K), int(basis), int(act),
        int(precision), int(algo), int(part), FWD_KEEP_WEIGHTS if (keep_weights and need > 0) else 0,
        _ptr(workspace) if need > 0 else _c_vp(),
        (workspace.numel() * workspace.element_size()) if need > 0 else 0, _stream_ptr(x.device),
    )
    check(rc, "dsph_poly_forward_ex")
    return out, workspace


def pool_fusable(plan, N, Fin, Fout, K, act=ACT_NONE):
    """Whether ``cheb_forward_pool`` can run this layer on this plan (``dsph_plan_pool_fusable``)."""
    return bool(lib().dsph_plan_pool_fusable(plan.handle, int(N), int(Fin), int(Fout), int(K), int(act)))


def cheb_forward_pool(plan, x, w, bias, K, pool_type=POOL_MAX, act=ACT_NONE, precision=PREC_FP32, workspace=None,
                      basis=BASIS_CHEBYSHEV, keep_weights=False):
    """pool(act(conv(x) + bias)) with HealpyPool(p = 1) reduced in the kernels' store step (``dsph_poly_forward_pool``):
    -> (y_pooled (N, rows / 4, Fout), workspace).  The full-resolution output is never written."""
    import torch

    _check_dev(x, plan, "x")
    _check_dev(w, plan, "w")
    N, rows, Fin = x.shape
    Fout = int(w.shape[1])
    need = plan.workspace_bytes(N, Fin, Fout, K, precision, ALGO_FUSED)
    if need > 0 and (workspace is None or workspace.numel() * workspace.element_size() < need):
        workspace = torch.empty(need, dtype=torch.uint8, device=x.device)
        keep_weights = False
    out = torch.empty((N, rows // 4, Fout), dtype=torch.float32, device=x.device)
    rc = lib().dsph_poly_forward_pool(
        plan.handle, _ptr(x), _ptr(w), _ptr(bias), _c_vp(), _ptr(out), int(N), int(Fin), Fout, int(K), int(basis), int(act),
        int(precision), int(pool_type), FWD_KEEP_WEIGHTS if (keep_weights and need > 0) else 0,
        _ptr(workspace) if need > 0 else _c_vp(), (workspace.numel() * workspace.element_size()) if need > 0 else 0,
        _stream_ptr(x.device))
    check(rc, "dsph_poly_forward_pool")
    return out, workspace


def cheb_step(plan, inp, prev, alpha, beta, rows=0, out=None):
    """out = alpha * (L~ @ inp) - beta * prev on (N, n_cols, F) planes."""
    import torch

    _check_dev(inp, plan, "in")
    N, r, F = inp.shape
    if r != plan.n_cols:
        raise ValueError("plane row count must equal the plan's n_cols")
    if prev is not None:
        _check_dev(prev, plan, "prev")
    if out is None:
        # rows the step does not produce (halo rows of a shard, rows beyond `rows`) stay zero
        full = plan.n_cols == plan.n_rows and (rows <= 0 or rows == plan.n_rows)
        out = torch.empty_like(inp) if full else torch.zeros_like(inp)
    rc = lib().dsph_cheb_step(plan.handle, _ptr(inp), _ptr(prev), _ptr(out), int(N), int(F), float(alpha),
                              float(beta), int(rows), _stream_ptr(inp.device))
    check(rc, "dsph_cheb_step")
    return out


def cheb_contract(planes, w, bias, rows, K, act=ACT_NONE, precision=PREC_FP32):
    """Contraction over a list of K (N, plane_rows, Fin) planes -> (N, rows, Fout)."""
    import torch

    p0 = planes[0]
    N, plane_rows, Fin = p0.shape
    Fout = int(w.shape[1])
    arr = (_c_vp * K)(*[p.data_ptr() for p in planes])
    out = torch.empty((N, rows, Fout), dtype=torch.float32, device=p0.device)
    rc = lib().dsph_cheb_contract(ctypes.cast(arr, _c_vp), int(plane_rows), _ptr(w), _ptr(bias), _ptr(out), int(N),
                                  int(rows), int(Fin), Fout, int(K), int(act), int(precision), p0.device.index,
                                  _stream_ptr(p0.device))
    check(rc, "dsph_cheb_contract")
    return out


def cheb_planes(plan, x, K, basis=BASIS_CHEBYSHEV, algo=ALGO_AUTO):
    """[x, T_1 x, ..., T_{K-1} x]: the recurrence without the contraction (``dsph_cheb_planes``).
    Every plane has x's shape (N, n_cols, Fin) and is valid on the plan's output rows."""
    import torch

    _check_dev(x, plan, "x")
    if x.dim() != 3 or x.shape[1] != plan.n_cols:
        raise ValueError(f"x must be (N, {plan.n_cols}, Fin), got {tuple(x.shape)}")
    N, M, Fin = x.shape
    if K <= 1:
        return [x]
    out = torch.empty((K - 1, N, M, Fin), dtype=torch.float32, device=x.device)
    rc = lib().dsph_cheb_planes(plan.handle, _ptr(x), _ptr(out), int(N), int(Fin), int(K), int(basis), int(algo),
                                _stream_ptr(x.device))
    check(rc, "dsph_cheb_planes")
    return [x] + [out[k] for k in range(K - 1)]


def cheb_backward_weights(plan, x, dy, K, basis=BASIS_CHEBYSHEV, algo=ALGO_AUTO, workspace=None, precision=PREC_FP32):
    """dkernel[f*K + k, o] = sum_{n,m} (T_k x)[n,m,f] dy[n,m,o] (``dsph_cheb_backward_weights``).
    Returns (dkernel, workspace)."""
    import torch

    _check_dev(x, plan, "x")
    _check_dev(dy, plan, "dy")
    N, M, Fin = x.shape
    if M != plan.n_cols or dy.dim() != 3 or dy.shape[0] != N or dy.shape[1] != plan.out_rows:
        raise ValueError(f"x must be (N, {plan.n_cols}, Fin) and dy (N, {plan.out_rows}, Fout)")
    Fout = int(dy.shape[2])
    need = int(lib().dsph_backward_weights_workspace_bytes(plan.handle, int(N), int(Fin), Fout, int(K), int(algo)))
    if workspace is None or workspace.numel() * workspace.element_size() < need or workspace.device != x.device:
        workspace = torch.empty(max(need, 16), dtype=torch.uint8, device=x.device)
    dw = torch.empty((Fin * K, Fout), dtype=torch.float32, device=x.device)
    rc = lib().dsph_cheb_backward_weights(plan.handle, _ptr(x), _ptr(dy), _ptr(dw), int(N), int(Fin), Fout, int(K),
                                          int(basis), int(precision), int(algo), _ptr(workspace),
                                          workspace.numel() * workspace.element_size(), _stream_ptr(x.device))
    check(rc, "dsph_cheb_backward_weights")
    return dw, workspace


def cheb_wgrad(planes, dy, rows=None, workspace=None):
    """dw[f*K + k, o] = sum_{n,m} planes[k][n,m,f] * dy[n,m,o] for a list of K (N, plane_rows, Fin) planes."""
    import torch

    p0 = planes[0]
    K = len(planes)
    N, plane_rows, Fin = p0.shape
    rows = int(dy.shape[1]) if rows is None else int(rows)
    Fout = int(dy.shape[2])
    need = int(lib().dsph_wgrad_workspace_bytes(int(N), rows, int(Fin), Fout, K))
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=p0.device)
    dw = torch.empty((Fin * K, Fout), dtype=torch.float32, device=p0.device)
    arr = (_c_vp * K)(*[p.data_ptr() for p in planes])
    rc = lib().dsph_cheb_wgrad(ctypes.cast(arr, _c_vp), int(plane_rows), _ptr(dy), _ptr(dw), int(N), rows, int(Fin),
                               Fout, K, _ptr(workspace), workspace.numel() * workspace.element_size(),
                               p0.device.index, _stream_ptr(p0.device))
    check(rc, "dsph_cheb_wgrad")
    return dw, workspace


def rows_pack(src, idx, out=None):
    """buf[n, i, :] = src[n, idx[i], :]; idx is an int32 CUDA tensor."""
    import torch

    N, rows, F = src.shape
    n_idx = int(idx.numel())
    if out is None:
        out = torch.empty((N, n_idx, F), dtype=torch.float32, device=src.device)
    rc = lib().dsph_rows_pack(_ptr(src), int(rows), _ptr(idx), n_idx, _ptr(out), int(N), int(F), src.device.index,
                              _stream_ptr(src.device))
    check(rc, "dsph_rows_pack")
    return out


def rows_unpack(dst, idx, buf):
    """dst[n, idx[i], :] = buf[n, i, :] in place."""
    N, rows, F = dst.shape
    rc = lib().dsph_rows_unpack(_ptr(dst), int(rows), _ptr(idx), int(idx.numel()), _ptr(buf), int(N), int(F),
                                dst.device.index, _stream_ptr(dst.device))
    check(rc, "dsph_rows_unpack")
    return dst


def residual_epilogue(y, skip, alpha=1.0, act=ACT_NONE, act_before=False):
    """In place, one pass (``dsph_residual_epilogue``): y = act(y + alpha * skip), or act(y) + alpha * skip."""
    if not (y.is_cuda and skip.is_cuda and y.device == skip.device):
        raise ValueError("residual_epilogue works on HIP tensors of one device")
    if y.dtype != skip.dtype or str(y.dtype) != "torch.float32" or y.shape != skip.shape:
        raise ValueError("y and skip must be float32 tensors of one shape")
    if not (y.is_contiguous() and skip.is_contiguous()):
        raise ValueError("y and skip must be contiguous")
    rc = lib().dsph_residual_epilogue(_ptr(y), _ptr(skip), int(y.numel()), float(alpha), int(act), 1 if act_before else 0,
                                      int(y.device.index), _stream_ptr(y.device))
    check(rc, "dsph_residual_epilogue")
    return y


def healpix_pool(x, group, pool_type=POOL_MAX):
    """y[n, m, f] = max | mean over the ``group`` = 4^p consecutive (NEST children) rows of x (``dsph_healpix_pool``)."""
    import torch

    if not x.is_cuda or x.dtype != torch.float32 or not x.is_contiguous() or x.dim() != 3:
        raise ValueError("healpix_pool works on a contiguous float32 (N, rows, F) HIP tensor")
    N, M, F = x.shape
    if M % group != 0:
        raise ValueError(f"{M} rows are not a multiple of the group size {group}")
    y = torch.empty((N, M // group, F), dtype=torch.float32, device=x.device)
    rc = lib().dsph_healpix_pool(_ptr(x), _ptr(y), int(N), int(M // group), int(F), int(group), int(pool_type), x.device.index,
                                 _stream_ptr(x.device))
    check(rc, "dsph_healpix_pool")
    return y


def healpix_pool_backward(x, dy, group, pool_type=POOL_MAX):
    """Gradient of ``healpix_pool`` with respect to its input (``dsph_healpix_pool_backward``)."""
    import torch

    N, Mo, F = dy.shape
    dx = torch.empty((N, Mo * group, F), dtype=torch.float32, device=dy.device)
    rc = lib().dsph_healpix_pool_backward(_ptr(x), _ptr(dy.contiguous()), _ptr(dx), int(N), int(Mo), int(F), int(group),
                                          int(pool_type), dy.device.index, _stream_ptr(dy.device))
    check(rc, "dsph_healpix_pool_backward")
    return dx
