"""Generates tests/golden/cheb_cases.npz from the float64 CPU oracle (oracle/cheb_oracle.py).

The reference cannot be imported here (TensorFlow/healpy/pygsp are absent) and holds no
golden vectors for this path, so these fixtures pin the *oracle's* outputs on seeded inputs:
they guard the oracle against regressions and give every backend one shared set of expected
values.  Shapes mirror the reference's own tests where it has any
(tests/test_gnn_layers.py:11-22: dense SPD 3x3 L, x (5,3,7), K=4, Fout=3; :94-114: L = I_192,
x (3,192,7), K=5).  Run from the repo root:  python tests/make_golden.py
"""

import os
import sys

import numpy as np
from scipy import sparse

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "deepsphere-cosmo-tf2_amd"))

from deepsphere import healpix  # noqa: E402  (graph producer only; not on the parity path)
from oracle import cheb_oracle as orc  # noqa: E402


def make_case(name, L, N, Fin, Fout, K, seed, bias=False, activation=None):
    rng = np.random.default_rng(seed)
    Lt, lmax = orc.prepare_L(L)
    M = Lt.shape[0]
    x = rng.standard_normal((N, M, Fin)).astype(np.float32)
    kernel = (rng.standard_normal((Fin * K, Fout)) * orc.default_kernel_stddev(Fin, K)).astype(np.float32)
    b = rng.standard_normal(Fout).astype(np.float32) if bias else None
    y = orc.chebyshev_forward(Lt, x, kernel, K, bias=b, activation=activation, dtype=np.float64)
    out = {
        f"{name}/L_data": Lt.data.astype(np.float32),
        f"{name}/L_indices": Lt.indices.astype(np.int32),
        f"{name}/L_indptr": Lt.indptr.astype(np.int32),
        f"{name}/lmax": np.float64(lmax),
        f"{name}/x": x,
        f"{name}/kernel": kernel,
        f"{name}/K": np.int32(K),
        f"{name}/y": y,
        f"{name}/activation": np.str_(activation or ""),
    }
    if b is not None:
        out[f"{name}/bias"] = b
    return out


def main():
    rng = np.random.default_rng(11)
    A = rng.standard_normal((3, 3))
    cases = {}
    cases.update(make_case("dense3", A @ A.T, 5, 7, 3, 4, seed=12))
    cases.update(make_case("eye192", np.eye(192), 3, 7, 7, 5, seed=13, activation="relu"))
    L4 = healpix.healpix_laplacian(4, n_neighbors=8, mode="knn")
    L8 = healpix.healpix_laplacian(8, n_neighbors=8, mode="knn")
    L8k20 = healpix.healpix_laplacian(8, n_neighbors=20, mode="knn")
    cases.update(make_case("n4_k5", L4, 2, 3, 4, 5, seed=14, bias=True, activation="relu"))
    cases.update(make_case("n4_k1", L4, 2, 3, 4, 1, seed=15))
    cases.update(make_case("n4_k2", L4, 2, 1, 16, 2, seed=16, bias=True))
    cases.update(make_case("n8_k5", L8, 2, 16, 32, 5, seed=17))
    cases.update(make_case("n8_nb20_k8", L8k20, 1, 3, 3, 8, seed=18, activation="elu"))
    idx = healpix.extend_indices(healpix.cap_indices(8, fraction=1.0 / 3.0), 8, 2)
    Lp = healpix.healpix_laplacian(8, indices=idx, n_neighbors=8, mode="knn")
    cases.update(make_case("n8_cap_k5", Lp, 1, 64, 64, 5, seed=19))
    cases["n8_cap_k5/indices"] = idx.astype(np.int64)
    Lg = healpix.healpix_laplacian(8, mode="grid")
    cases.update(make_case("n8_grid_k5", Lg, 2, 8, 8, 5, seed=20))
    path = os.path.join(ROOT, "tests", "golden", "cheb_cases.npz")
    np.savez_compressed(path, **cases)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB;", sorted({k.split('/')[0] for k in cases}))


if __name__ == "__main__":
    main()
