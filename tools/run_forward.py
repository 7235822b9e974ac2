#!/usr/bin/env python3
"""Runs a few forwards of one benchmark config and nothing else: the target for rocprofv3
(--kernel-trace --stats, or one --pmc pass at a time).  Usage: run_forward.py c3 bf16x3 fused 3"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "deepsphere-cosmo-tf2_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from deepsphere import _native  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
prec = {"fp32": _native.PREC_FP32, "bf16x3": _native.PREC_BF16X3, "bf16x6": _native.PREC_BF16X6}[sys.argv[2] if len(sys.argv) > 2 else "bf16x3"]
algo = {"auto": 0, "unfused": 1, "fused": 2}[sys.argv[3] if len(sys.argv) > 3 else "auto"]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
nside, K, Fin, Fout, N = bench.CONFIGS[cfg]
dev = torch.device("cuda", 0)
if cfg in bench.KNN:
    cols, vals, lmax = bench.build_laplacian_knn(nside, dev, bench.KNN[cfg])
else:
    cols, vals, lmax = bench.build_laplacian_masked(nside, dev, nside_super=bench.MASKED[cfg]) if cfg in bench.MASKED else bench.build_laplacian(nside, dev)
plan = _native.LaplacianPlan(cols, vals, device=0)
M = cols.shape[0]
x = torch.randn((N, M, Fin), device=dev)
w = torch.randn((Fin * K, Fout), device=dev) / np.sqrt(Fin * (K + 0.5) / 2)
# calibration dispatch for the PMC passes: a float4 copy of x reads |x| and writes |x| bytes exactly
xc = torch.empty_like(x)
xc.copy_(x)
del xc
ws = None
for _ in range(reps):
    y, ws = _native.cheb_forward(plan, x, w, None, K, precision=prec, algo=algo, workspace=ws)
torch.cuda.synchronize()
print("done", cfg, float(y.abs().max()))
