"""Host-side utilities of the Chebyshev path (scipy/numpy; nothing here touches the GPU
except ``split_sparse_dense_matmul``).  Mirrors the public names of the reference's
``deepsphere/utils.py``."""

import numpy as np
from scipy import sparse
from scipy.sparse.linalg import eigsh

from .healpix import extend_indices  # noqa: F401  (same public name as utils.py:9)

__all__ = ["extend_indices", "rescale_L", "prepare_L", "csr_to_ell", "split_sparse_dense_matmul", "lanczos_lmax",
           "rescale_ell"]


def rescale_L(L, lmax=2, scale=1):
    """Rescale the Laplacian eigenvalues into [-scale, scale]: ``L * (2*scale/lmax) - I``.

    Same signature and result as the reference (``utils.py:40-46``).  The reference scales
    the caller's ``.data`` in place as a side effect; this returns a new matrix and leaves
    the argument untouched (the layer result is unaffected: the preparation is invariant
    to a rescaling of L)."""
    L = sparse.csr_matrix(L, dtype=np.float64, copy=True)
    M = L.shape[0]
    L = L * (2.0 * scale / lmax) - sparse.identity(M, format="csr", dtype=np.float64)
    return L.tocsr()


def prepare_L(L, scale=0.75, tol=0):
    """One-time Laplacian preparation of ``Chebyshev.__init__`` (``gnn_layers.py:64-72``).

    csr_matrix(L) -> lmax = 1.02 * eigsh(k=1, which="LM") -> rescale_L(scale=0.75), float64
    arithmetic, values cast to float32 at the end (the reference stores floatx constants).
    Returns (Lt CSR float32 with sorted indices, lmax).  ``tol`` is ARPACK's tolerance
    (0 = machine precision, the reference's setting)."""
    Lc = sparse.csr_matrix(L, dtype=np.float64)
    if Lc.shape[0] != Lc.shape[1]:
        raise ValueError("the graph Laplacian must be square")
    if Lc.shape[0] <= 2:
        ev = np.linalg.eigvalsh(Lc.toarray())
        lam = ev[np.argmax(np.abs(ev))]
    else:
        lam = eigsh(Lc, k=1, which="LM", return_eigenvectors=False, tol=tol)[0]
    lmax = 1.02 * float(lam)
    Lt = rescale_L(Lc, lmax=lmax, scale=scale)
    Lt.sort_indices()
    return Lt.astype(np.float32), lmax


def csr_to_ell(A, width=None):
    """CSR -> padded ELL: (cols int32 [M, W], vals float32 [M, W]), W = max row length.

    Entries keep their CSR order inside a row (column-ascending after sort_indices), so the
    kernels sum a row in the same order as a CSR product.  Padding slots point at the row's
    own index with value 0."""
    A = sparse.csr_matrix(A)
    M = A.shape[0]
    lens = np.diff(A.indptr)
    W = int(lens.max()) if M > 0 and A.nnz > 0 else 1
    if width is not None:
        if width < W:
            raise ValueError(f"requested ELL width {width} < longest row {W}")
        W = int(width)
    rows = np.repeat(np.arange(M, dtype=np.int64), lens)
    slot = np.arange(A.nnz, dtype=np.int64) - np.repeat(A.indptr[:-1].astype(np.int64), lens)
    cols = np.repeat(np.arange(M, dtype=np.int32)[:, None], W, axis=1)
    cols = np.minimum(cols, A.shape[1] - 1).astype(np.int32)
    vals = np.zeros((M, W), dtype=np.float32)
    cols[rows, slot] = A.indices.astype(np.int32)
    vals[rows, slot] = A.data.astype(np.float32)
    return cols, vals


def split_sparse_dense_matmul(sparse_tensor, dense_tensor, n_splits=1):
    """Sparse @ dense on the GPU: ``sparse_tensor`` is a ``LaplacianPlan`` (the uploaded
    Laplacian), ``dense_tensor`` a float32 CUDA tensor of shape (M, C).

    Same name and arguments as the reference helper (``utils.py:49-78``), whose ``n_splits``
    only exists to dodge TensorFlow-GPU's ``nnz * ncols <= 2**31`` limit; the ELL kernel has
    no such limit, so the split count is validated (it must divide C, like ``tf.split``) and
    otherwise ignored."""
    from . import _native

    if dense_tensor.dim() != 2:
        raise ValueError("dense_tensor must have rank 2")
    if n_splits > 1 and dense_tensor.shape[1] % n_splits != 0:
        raise ValueError("n_splits must divide the number of dense columns")
    out = _native.cheb_step(sparse_tensor, dense_tensor.contiguous().unsqueeze(0), None, 1.0, 0.0)
    return out.squeeze(0)


def rescale_ell(cols, vals, lmax, scale=0.75):
    """``rescale_L`` on a padded-ELL Laplacian whose slot 0 is the diagonal (the layout of
    ``healpix.grid_laplacian_ell``): vals * (2*scale/lmax), minus 1 on the diagonal slot.
    Accepts numpy arrays or torch tensors; returns float32 values."""
    out = vals * (2.0 * scale / lmax)
    out[:, 0] -= 1.0
    if hasattr(out, "numpy") and not isinstance(out, np.ndarray):
        import torch

        return out.to(torch.float32)
    return out.astype(np.float32)


def lanczos_lmax(plan, iters=64, seed=0):
    """Largest-magnitude eigenvalue of the matrix held by ``plan`` from ``iters`` Lanczos steps
    with full re-orthogonalisation, entirely on the GPU (products through ``dsph_cheb_step``).

    ARPACK at machine precision -- what the reference calls (``gnn_layers.py:66``) -- takes ten
    minutes at nside 512 and hours at nside 1024 because the spectrum is dense near its top
    edge; this gives the edge to ~1e-4 in a second, enough to scale the benchmark Laplacian.
    The layer constructor keeps the reference's exact ``eigsh`` call."""
    import torch

    from . import _native

    M = plan.n_rows
    if plan.n_cols != M:
        raise ValueError("lanczos_lmax needs a square plan")
    dev = torch.device("cuda", plan.device)
    gen = torch.Generator(device=dev).manual_seed(seed)
    iters = int(min(iters, M))
    Q = torch.zeros((iters + 1, M), dtype=torch.float64, device=dev)
    q = torch.randn(M, dtype=torch.float64, device=dev, generator=gen)
    Q[0] = q / torch.linalg.norm(q)
    alphas, betas = [], []
    for j in range(iters):
        w = _native.cheb_step(plan, Q[j].to(torch.float32).reshape(1, M, 1).contiguous(), None, 1.0, 0.0)
        w = w.reshape(M).to(torch.float64)
        alphas.append(torch.dot(w, Q[j]).item())
        w = w - Q[: j + 1].T @ (Q[: j + 1] @ w)
        w = w - Q[: j + 1].T @ (Q[: j + 1] @ w)
        b = torch.linalg.norm(w).item()
        if b < 1e-12:
            break
        betas.append(b)
        Q[j + 1] = w / b
    n = len(alphas)
    T = np.diag(alphas) + np.diag(betas[: n - 1], 1) + np.diag(betas[: n - 1], -1)
    ev = np.linalg.eigvalsh(T)
    return float(ev[np.argmax(np.abs(ev))])
