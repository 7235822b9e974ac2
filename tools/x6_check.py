import sys
sys.path.insert(0, "tools"); sys.path.insert(0, "."); sys.path.insert(0, "deepsphere-cosmo-tf2_amd")
import numpy as np, torch
import check_struct as cs
from deepsphere import _native
from oracle import cheb_oracle as orc
import bench
from scipy import sparse
for nside, Fin, Fout, K in [(64, 64, 64, 5), (64, 1, 16, 5), (64, 16, 32, 4), (64, 20, 40, 4), (64, 8, 64, 2), (64, 32, 130, 3), (128, 64, 64, 5)]:
    cols, vals, _ = bench.build_laplacian(nside, torch.device("cuda", 0))
    M, W = cols.shape
    L = sparse.csr_matrix((vals.reshape(-1).astype(np.float64), cols.reshape(-1), np.arange(0, W * M + 1, W)), shape=(M, M))
    plan = _native.LaplacianPlan(cols, vals, device=0)
    rng = np.random.default_rng(1)
    x = rng.standard_normal((3, M, Fin)).astype(np.float32)
    Wt = (rng.standard_normal((Fin * K, Fout)) * orc.default_kernel_stddev(Fin, K)).astype(np.float32)
    ref = orc.chebyshev_forward(L, x, Wt, K)
    for name, P in [("fp32", 0), ("bf16x3", 1), ("bf16x6", 2)]:
        y, _ = _native.cheb_forward(plan, torch.tensor(x).cuda(), torch.tensor(Wt).cuda(), None, K, precision=P, algo=_native.ALGO_FUSED)
        print(nside, Fin, Fout, K, name, "err %.2e" % (np.abs(y.cpu().numpy() - ref).max() / np.abs(ref).max()))
for prec in ["bf16x6", "fp32", "bf16x3"]:
    cs.timing(1024, 5, 64, 64, 4, prec, reps=6)
