#!/usr/bin/env python3
"""A DeepSphere-style stack end to end (SURVEY 8 f: the callers either side of the path): Chebyshev layers with NEST max-pooling
between them, from nside 512 down to nside 8 -- the shapes a network has, first layers to the small maps at its end.
    python tools/bench_net.py [batch] [knn] [nofuse] [bn]   (knn: the reference's 8-neighbour k-NN graphs instead of the grid stencil;
                                                        nofuse: no conv + pool fusion; bn: every layer with use_bn=True, the
                                                        reference models' pattern -- inference, moving statistics folded)
Prints one JSON line: ms per layer (HIP events) and for the whole forward."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "deepsphere-cosmo-tf2_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from deepsphere import gnn_layers, healpy_layers  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
knn = "knn" in sys.argv[2:]
use_bn = "bn" in sys.argv[2:]
nofuse = "nofuse" in sys.argv[2:]  # every layer and every pooling on its own (default: conv + pool in one pass where the kernels can)
dev = torch.device("cuda", 0)
STACK = [(512, 1, 16), (256, 16, 32), (128, 32, 64), (64, 64, 64), (32, 64, 64), (16, 64, 128), (8, 128, 128)]
layers = []
for nside, Fin, Fout in STACK:
    cols, vals, lmax = bench.build_laplacian_knn(nside, dev, 8) if knn else bench.build_laplacian(nside, dev)
    layers.append(gnn_layers.Chebyshev.from_prepared_ell(cols, vals, 5, lmax=lmax, Fout=Fout, device=dev, use_bias=True, activation="relu", use_bn=use_bn))
pool = healpy_layers.HealpyPool(p=1, pool_type="MAX")
x0 = torch.randn((N, 12 * 512 * 512, 1), device=dev)


def forward(events=None):
    x = x0
    for i, layer in enumerate(layers):
        if events is not None:
            events[i][0].record()
        y = None if (nofuse or i + 1 == len(layers)) else layer.forward_pool(x, "MAX")
        if y is None:
            x = layer(x)
            if events is not None:
                events[i][1].record()
            if i + 1 < len(layers):
                x = pool(x)
        else:
            x = y
            if events is not None:
                events[i][1].record()
    return x


with torch.no_grad():
    for _ in range(3):
        forward()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in evs:
        a.record()
        forward()
        b.record()
    torch.cuda.synchronize()
    each = sorted(a.elapsed_time(b) for a, b in evs)
    total = each[len(each) // 2]  # median of 20 forwards (one event pair each)
    per = np.zeros(len(layers))
    for _ in range(10):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in layers]
        forward(ev)
        torch.cuda.synchronize()
        per += np.array([s.elapsed_time(e) for s, e in ev]) / 10
print(json.dumps({"stack": [f"nside {n}: {fi}->{fo}" for n, fi, fo in STACK], "K": 5, "batch": N, "graph": "knn8" if knn else "grid",
                  "layer_ms": [round(float(v), 3) for v in per], "forward_ms": round(total, 3),
                  "forward_ms_min_max": [round(each[0], 3), round(each[-1], 3)],
                  "fused_pooling": not nofuse, "use_bn": use_bn,
                  "note": "Chebyshev (bias, ReLU) + HealpyPool(MAX, p=1) between layers; layer_ms by one event pair per layer (a layer that pools in its own epilogue: incl. the pooling)"}))
