#!/usr/bin/env python3
"""Forward time of layer shapes outside the headline (first layers, wide layers): nside K Fin Fout N ..."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "deepsphere-cosmo-tf2_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from deepsphere import gnn_layers  # noqa: E402

dev = torch.device("cuda", 0)
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(1024, 5, 1, 64, 4), (1024, 5, 64, 128, 4)]
cache = {}
for nside, K, Fin, Fout, N in shapes:
    if nside not in cache:
        kk = int(os.environ.get("BS_KNN", "0"))  # 0: the grid stencil; 8 / 20: the reference's k-nearest-neighbour graphs
        cache[nside] = bench.build_laplacian_knn(nside, dev, kk) if kk else bench.build_laplacian(nside, dev)
    cols, vals, lmax = cache[nside]
    from deepsphere import _native
    layer = gnn_layers.Chebyshev.from_prepared_ell(cols, vals, K, lmax=lmax, Fout=Fout, device=dev, precision=os.environ.get("BS_PREC", "bf16x3"),
                                                   plan_options={_native.OPT_STRIPS: int(os.environ.get("BS_STRIPS", "0"))})
    x = torch.randn((N, cols.shape[0], Fin), device=dev)
    with torch.no_grad():
        for _ in range(3):
            y = layer(x)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            y = layer(x)
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / 10 * 1e3
    M = cols.shape[0]
    b_alg = bench.algorithmic_bytes(N, M, Fin, Fout, K, cols.shape[1])
    fused = layer._get_plan().fused_ok(Fin, Fout, K)
    print(json.dumps({"nside": nside, "K": K, "Fin": Fin, "Fout": Fout, "batch": N, "ms": round(ms, 3), "fused": bool(fused),
                      "roofline_frac": round(b_alg / (ms * 1e-3) / 8e12, 4)}), flush=True)
    del layer, x, y
