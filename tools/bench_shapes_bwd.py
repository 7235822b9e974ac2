#!/usr/bin/env python3
"""Forward + backward time of layer shapes (the small maps at the end of a network included): nside,K,Fin,Fout,N ..."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "deepsphere-cosmo-tf2_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
from deepsphere import gnn_layers  # noqa: E402

dev = torch.device("cuda", 0)
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(32, 5, 64, 64, 8)]
cache = {}
for nside, K, Fin, Fout, N in shapes:
    if nside not in cache:
        cache[nside] = bench.build_laplacian(nside, dev)
    cols, vals, lmax = cache[nside]
    layer = gnn_layers.Chebyshev.from_prepared_ell(cols, vals, K, lmax=lmax, Fout=Fout, device=dev)
    x = torch.randn((N, cols.shape[0], Fin), device=dev, requires_grad=True)
    dy = torch.randn((N, cols.shape[0], Fout), device=dev)

    def step():
        y = layer(x)
        y.backward(dy)
        x.grad = None
        layer.kernel.grad = None

    def fwd():
        with torch.no_grad():
            layer(x)

    out = {}
    for name, fn in (("fwd_bwd", step), ("fwd", fwd)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        out[name + "_ms"] = round((time.perf_counter() - t) / 20 * 1e3, 3)
    print(json.dumps({"nside": nside, "K": K, "Fin": Fin, "Fout": Fout, "batch": N, **out}), flush=True)
