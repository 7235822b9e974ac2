#!/usr/bin/env python3
"""Forward + backward of the layer at a benchmark config (SURVEY 8 f1 measurement).

    python tools/bench_backward.py [c3|c2|c5|...] [steps] [bf16x3|bf16x6|fp32]
Prints one JSON line: ms per forward+backward step (HIP events), and its split."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "deepsphere-cosmo-tf2_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from deepsphere import gnn_layers  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
nside, K, Fin, Fout, N = bench.CONFIGS[cfg]
dev = torch.device("cuda", 0)
cols, vals, lmax = bench.build_laplacian_masked(nside, dev, nside_super=bench.MASKED[cfg]) if cfg in bench.MASKED else bench.build_laplacian(nside, dev)
M = cols.shape[0]
w_np = (np.random.default_rng(13).standard_normal((Fin * K, Fout)) / np.sqrt(Fin * (K + 0.5) / 2)).astype(np.float32)
prec = sys.argv[3] if len(sys.argv) > 3 else "bf16x3"
layer = gnn_layers.Chebyshev.from_prepared_ell(cols, vals, K, lmax=lmax, Fout=Fout, device=dev, precision=prec,
                                               initializer=lambda t: t.copy_(torch.from_numpy(w_np)))
x = torch.randn((N, M, Fin), device=dev).requires_grad_(True)
dy = torch.randn((N, M, Fout), device=dev)


def step():
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    y = layer(x)
    e[1].record()
    y.backward(dy)
    e[2].record()
    x.grad = None
    layer.kernel.grad = None
    return e


for _ in range(2):
    step()
evs = [step() for _ in range(steps)]
torch.cuda.synchronize()
fwd = float(np.mean([a.elapsed_time(b) for a, b, _ in evs]))
bwd = float(np.mean([b.elapsed_time(c) for _, b, c in evs]))
print(json.dumps({"config": cfg, "workload": f"nside={nside} K={K} Fin={Fin} Fout={Fout} batch={N}", "steps": steps,
                  "forward_ms": round(fwd, 3), "backward_ms": round(bwd, 3), "fwd_bwd_ms": round(fwd + bwd, 3),
                  "note": "dx: the fused forward kernels on dy (transposed plan, re-indexed weights); dkernel: "
                          "dsph_cheb_backward_weights -- K = 5, 64 -> 64 j on a symmetric L~ in the three-term bf16 arithmetic: the "
                          "strips' pixels on the quad-strip weight-gradient kernel (cheb_qwgrad_kernel.h), the other tiles on the "
                          "BFS-tile kernel's weight-gradient mode; otherwise that mode on every tile (planes stay in LDS, fixed-order "
                          "sums of per-workgroup slabs)"}))
