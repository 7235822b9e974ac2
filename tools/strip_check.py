"""Strip kernel (cheb_strip_kernel.h) against the float64 oracle on a whole map, with a per-tile error map.
usage: python tools/strip_check.py [nside] [N] [K] [basis]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "deepsphere-cosmo-tf2_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from deepsphere import _native  # noqa: E402
from oracle import cheb_oracle as orc  # noqa: E402
from scipy import sparse  # noqa: E402
import bench  # noqa: E402

nside = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2
K = int(sys.argv[3]) if len(sys.argv) > 3 else 5
basis = sys.argv[4] if len(sys.argv) > 4 else "chebyshev"
Fin = Fout = 64
cols, vals, _ = bench.build_laplacian(nside, torch.device("cuda", 0))
M, W = cols.shape
L = sparse.csr_matrix((vals.reshape(-1).astype(np.float64), cols.reshape(-1), np.arange(0, W * M + 1, W)), shape=(M, M))
plan = _native.LaplacianPlan(cols, vals, device=0)
print("tile counts", plan.tile_counts(K), "tiles", M // 256)
rng = np.random.default_rng(5)
x = rng.standard_normal((N, M, Fin)).astype(np.float32)
Wt = (rng.standard_normal((Fin * K, Fout)) * orc.default_kernel_stddev(Fin, K)).astype(np.float32)
b = rng.standard_normal(Fout).astype(np.float32)
fwd = orc.chebyshev_forward if basis == "chebyshev" else orc.monomial_forward
ref = fwd(L, x, Wt, K, bias=b, activation="relu")
B = {"chebyshev": _native.BASIS_CHEBYSHEV, "monomial": _native.BASIS_MONOMIAL}[basis]
d = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
y, _ = _native.cheb_forward(plan, d(x), d(Wt), d(b), K, act=_native.ACT_RELU, precision=_native.PREC_BF16X3,
                            algo=_native.ALGO_FUSED, basis=B)
torch.cuda.synchronize()
yn = y.cpu().numpy()
s = np.abs(ref).max()
err = np.abs(yn - ref).max(axis=(0, 2)) / s  # per pixel
print(f"nside {nside} N {N} K {K} {basis}: max rel err {err.max():.3e}")
te = err.reshape(-1, 256).max(axis=1)
bad = np.nonzero(te > 1e-5)[0]
print("tiles over 1e-5:", bad.size, "of", te.size, bad[:40])
if bad.size:
    t = bad[0]
    e = err[t * 256:(t + 1) * 256]
    print("first bad tile", t, "pixel errs (morton order) max at", int(e.argmax()), e.max())
    from deepsphere.healpix import nest2xyf
    px = np.nonzero(err > 1e-5)[0]
    xx, yy, ff = nest2xyf(nside, px)
    print("bad pixels x range", xx.min(), xx.max(), "y range", yy.min(), yy.max(), "faces", np.unique(ff))
    print("bad x mod 24 hist (x-16):", np.bincount((xx - 16) % 24, minlength=24))
    print("bad y hist:", np.bincount(yy)[:40])
y2, _ = _native.cheb_forward(plan, d(x), d(Wt), d(b), K, act=_native.ACT_RELU, precision=_native.PREC_BF16X3,
                             algo=_native.ALGO_FUSED, basis=B)
print("deterministic:", bool(torch.equal(y, y2)))
