"""Diagnostic, not a test (pytest does not collect it; it lives here because it checks against oracle/): rows of dW at a large
size against the float64 oracle, by order k, for the three weight-gradient routes -- max and mean SIGNED error, which is how the
matrix pipe's accumulation bias was found (profiles/r5_dw_error_by_order.txt).
    python tests/diag_dw_by_order.py [nside] [N]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "deepsphere-cosmo-tf2_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench  # noqa: E402
from deepsphere import _native  # noqa: E402
from oracle import cheb_oracle as orc  # noqa: E402
from scipy import sparse  # noqa: E402

nside = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4
K, Fin, Fout = 5, 64, 64
cols, vals, _ = bench.build_laplacian(nside, torch.device("cuda", 0))
M, W = cols.shape
Lc = sparse.csr_matrix((vals.reshape(-1).astype(np.float64), cols.reshape(-1), np.arange(0, W * M + 1, W)), shape=(M, M))
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.randn((N, M, Fin), device="cuda", generator=g)
dy = torch.randn((N, M, Fout), device="cuda", generator=g)
routes = {}
plan = _native.LaplacianPlan(cols, vals, device=0)
plan.prepare(K, Fin)
routes["quad bf16x3"] = _native.cheb_backward_weights(plan, x, dy, K, algo=_native.ALGO_FUSED, precision=_native.PREC_BF16X3)[0]
routes["bfs fp32"] = _native.cheb_backward_weights(plan, x, dy, K, algo=_native.ALGO_FUSED, precision=_native.PREC_FP32)[0]
plain = _native.LaplacianPlan(cols, vals, device=0, options={_native.OPT_STRIPS: _native.STRIPS_NEVER})
plain.prepare(K, Fin)
routes["bfs bf16x3"] = _native.cheb_backward_weights(plain, x, dy, K, algo=_native.ALGO_FUSED, precision=_native.PREC_BF16X3)[0]
scale = float(routes["bfs fp32"].abs().max())
fs, osub = (0, 29, 63), slice(8, 16)
dy_h = dy[:, :, osub].cpu().numpy().astype(np.float64)
err = {name: np.zeros(K) for name in routes}
mean = {name: np.zeros(K) for name in routes}
for f in fs:
    planes = orc.chebyshev_planes(Lc, x[:, :, f:f + 1].cpu().numpy(), K)
    ref = np.einsum("knm,nmo->ko", planes[..., 0], dy_h)
    for name, got in routes.items():
        d = got[f * K:(f + 1) * K, osub].cpu().numpy() - ref
        err[name] = np.maximum(err[name], np.abs(d).max(axis=1) / scale)
        mean[name] += d.mean(axis=1) / scale / len(fs)
print(f"nside {nside} N {N}: max |dW| {scale:.1f}")
for name in routes:
    print(f"  {name:12s} max err by order: " + " ".join(f"{e:.2e}" for e in err[name]) + "   mean signed err: " + " ".join(f"{e:+.1e}" for e in mean[name]))
