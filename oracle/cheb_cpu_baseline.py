"""Timed CPU baseline: the reference's op sequence in fp32 on the host cores.  TEST INFRASTRUCTURE.

Same materialisations as ``Chebyshev.call`` (``/root/reference/src/deepsphere/gnn_layers.py:131-150``):
transpose to M x Fin*N, (K-1) sparse @ dense products with the axpy temporaries, stack,
reshape/transpose to N*M x Fin*K, dense matmul.  TensorFlow is not installed on either
box, so its two kernels are stood in for by torch-CPU's (``torch.sparse.mm`` on CSR,
``torch.matmul``), multi-threaded over the host cores.  Used only by ``bench.py``'s
``cpu_baseline`` leg (kind "port") and by a test that checks it against ``cheb_oracle``.
"""

import time

import numpy as np
import torch


def to_torch_csr(Lt):
    Lt = Lt.tocsr()
    return torch.sparse_csr_tensor(
        torch.from_numpy(Lt.indptr.astype(np.int64)),
        torch.from_numpy(Lt.indices.astype(np.int64)),
        torch.from_numpy(Lt.data.astype(np.float32)),
        size=Lt.shape,
        dtype=torch.float32,
    )


def forward_fp32(L_csr, x, kernel, K):
    """x (N,M,Fin) float32 torch CPU tensor -> (N,M,Fout); op order of gnn_layers.py:131-150."""
    N, M, Fin = x.shape
    Fout = kernel.shape[1]
    x0 = x.permute(1, 2, 0).reshape(M, Fin * N)
    stack = [x0]
    if K > 1:
        x1 = torch.sparse.mm(L_csr, x0)
        stack.append(x1)
    for _k in range(2, K):
        x2 = 2 * torch.sparse.mm(L_csr, x1) - x0
        stack.append(x2)
        x0, x1 = x1, x2
    xs = torch.stack(stack, dim=0).reshape(K, M, Fin, N).permute(3, 1, 2, 0).reshape(N * M, Fin * K)
    return torch.matmul(xs, kernel).reshape(N, M, Fout)


def time_forward(Lt, x_np, kernel_np, K, budget_s=20.0, min_reps=1, threads=None):
    """Run the port on a bounded sample; returns dict(seconds per forward, reps, threads)."""
    if threads is not None:
        torch.set_num_threads(int(threads))
    L_csr = to_torch_csr(Lt)
    x = torch.from_numpy(np.ascontiguousarray(x_np, dtype=np.float32))
    w = torch.from_numpy(np.ascontiguousarray(kernel_np, dtype=np.float32))
    forward_fp32(L_csr, x[:1], w, K)  # warm-up on one map
    times = []
    t_start = time.perf_counter()
    while len(times) < min_reps or (time.perf_counter() - t_start) < budget_s:
        t0 = time.perf_counter()
        y = forward_fp32(L_csr, x, w, K)
        times.append(time.perf_counter() - t0)
        if len(times) >= 50:
            break
    return {"seconds": float(np.median(times)), "reps": len(times), "threads": torch.get_num_threads(), "y": y}


def time_forward_scipy_1thread(Lt, x_np, kernel_np, K, budget_s=10.0):
    """The same op sequence with scipy CSR @ dense (single-threaded) and a numpy GEMM restricted to the
    calling thread's BLAS default: the 'scalar port' figure of SURVEY 8(d).  Returns seconds per forward."""
    import contextlib
    import time

    try:
        from threadpoolctl import threadpool_limits
        one_thread = threadpool_limits(limits=1)
    except Exception:  # threadpoolctl missing: BLAS keeps its default thread count (reported by the caller)
        one_thread = contextlib.nullcontext()
    N, M, Fin = x_np.shape
    Lc = Lt.tocsr().astype(np.float32)
    times = []
    t_all = time.perf_counter()
    with one_thread:
      while True:
        t0 = time.perf_counter()
        x0 = np.ascontiguousarray(np.transpose(x_np, (1, 2, 0)).reshape(M, Fin * N))
        planes = [x0]
        if K > 1:
            planes.append(Lc @ x0)
        for _k in range(2, K):
            planes.append(2 * (Lc @ planes[-1]) - planes[-2])
        X = np.stack(planes, axis=0).reshape(K, M, Fin, N)
        X = np.transpose(X, (3, 1, 2, 0)).reshape(N * M, Fin * K)
        y = (X @ kernel_np).reshape(N, M, -1)
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_all > budget_s or len(times) >= 3:
            break
    return {"seconds": float(np.median(times)), "reps": len(times), "y": y}


# ---- multi-threaded port (oracle/cheb_port.c): the figure bench.py reports as cpu_baseline ----------------------

_PORT = None


def load_port():
    """Compile (gcc -O3 -march=native -fopenmp, on the box it runs on) and load oracle/cheb_port.c."""
    global _PORT
    if _PORT is not None:
        return _PORT
    import ctypes
    import os
    import subprocess

    here = os.path.dirname(os.path.abspath(__file__))
    out_dir = os.path.join(here, "_build")
    os.makedirs(out_dir, exist_ok=True)
    # -march=native code must not travel between hosts: one library per CPU model
    import hashlib

    try:
        with open("/proc/cpuinfo") as f:
            cpu = next((ln for ln in f if ln.startswith("flags")), "")
    except OSError:
        cpu = ""
    so = os.path.join(out_dir, "libcheb_port_%s.so" % hashlib.sha1(cpu.encode()).hexdigest()[:10])
    src = os.path.join(here, "cheb_port.c")
    cmd = ["gcc", "-O3", "-march=native", "-fopenmp", "-shared", "-fPIC", src, "-o", so]
    lib = None
    if os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(src):
        try:
            lib = ctypes.CDLL(so)
        except OSError:
            lib = None
    if lib is None:
        subprocess.run(cmd, check=True)
        lib = ctypes.CDLL(so)
    i64, vp, f32, i32 = ctypes.c_int64, ctypes.c_void_p, ctypes.c_float, ctypes.c_int
    lib.port_relayout_in.argtypes = [vp, vp, i64, i64, i64, i32]
    lib.port_spmm.argtypes = [vp, vp, i64, i64, vp, vp, vp, i64, f32, f32, i32]
    lib.port_relayout_out.argtypes = [vp, vp, i64, i64, i64, i64, i32]
    for fn in (lib.port_relayout_in, lib.port_spmm, lib.port_relayout_out):
        fn.restype = None
    _PORT = lib
    return lib


def forward_threaded(ell_cols, ell_vals, x_np, kernel_np, K, threads):
    """The reference op sequence (gnn_layers.py:131-150) with every op spread over `threads` host threads:
    relayout, K-1 sparse products with the axpy, stack + transposes, dense GEMM (numpy -> BLAS threads)."""
    import ctypes

    lib = load_port()
    N, M, Fin = x_np.shape
    W = ell_cols.shape[1]
    cols = np.ascontiguousarray(ell_cols, dtype=np.int32)
    vals = np.ascontiguousarray(ell_vals, dtype=np.float32)
    x = np.ascontiguousarray(x_np, dtype=np.float32)
    C = Fin * N
    p = lambda a: a.ctypes.data  # noqa: E731
    planes = [np.empty((M, C), dtype=np.float32)]
    lib.port_relayout_in(p(x), p(planes[0]), N, M, Fin, threads)
    if K > 1:
        planes.append(np.empty((M, C), dtype=np.float32))
        lib.port_spmm(p(cols), p(vals), M, W, p(planes[0]), None, p(planes[1]), C, 1.0, 0.0, threads)
    for k in range(2, K):
        planes.append(np.empty((M, C), dtype=np.float32))
        lib.port_spmm(p(cols), p(vals), M, W, p(planes[k - 1]), p(planes[k - 2]), p(planes[k]), C, 2.0, 1.0, threads)
    X = np.empty((N * M, Fin * K), dtype=np.float32)
    arr = (ctypes.c_void_p * K)(*[p(a) for a in planes])
    lib.port_relayout_out(arr, p(X), K, N, M, Fin, threads)
    return (X @ np.ascontiguousarray(kernel_np, dtype=np.float32)).reshape(N, M, -1)


def time_forward_threaded(ell_cols, ell_vals, x_np, kernel_np, K, threads, budget_s=20.0):
    import time

    try:
        from threadpoolctl import threadpool_limits
        ctx = threadpool_limits(limits=int(threads))
    except Exception:
        import contextlib
        ctx = contextlib.nullcontext()
    times = []
    t_all = time.perf_counter()
    with ctx:
        y = forward_threaded(ell_cols, ell_vals, x_np[:1], kernel_np, K, threads)  # warm-up, one map
        while True:
            t0 = time.perf_counter()
            y = forward_threaded(ell_cols, ell_vals, x_np, kernel_np, K, threads)
            times.append(time.perf_counter() - t0)
            if time.perf_counter() - t_all > budget_s or len(times) >= 5:
                break
    return {"seconds": float(np.median(times)), "reps": len(times), "threads": int(threads), "y": y}
