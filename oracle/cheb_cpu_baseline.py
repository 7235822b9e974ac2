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
