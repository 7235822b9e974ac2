"""Host-side cost of one small forward (BASELINE configs[0]): the layer call, the ctypes wrapper, the bare C call, each timed
over many asynchronous submissions (one synchronisation at the end), and a cProfile of the layer call."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "deepsphere-cosmo-tf2_amd"))
import numpy as np
import torch

import bench
from deepsphere import _native, gnn_layers

dev = torch.device("cuda", 0)
cols, vals, _ = bench.build_laplacian(64, dev)
M = cols.shape[0]
layer = gnn_layers.Chebyshev.from_prepared_ell(cols, vals, 5, Fout=16, device=dev)
x = torch.randn(1, M, 1, device=dev)
with torch.no_grad():
    for _ in range(20):
        y = layer(x)
    torch.cuda.synchronize()

    def rate(fn, n=3000):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        t_submit = time.perf_counter() - t
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t
        return 1e6 * t_submit / n, 1e6 * t_all / n

    print("layer(x):            submit %.1f us, with drain %.1f us" % rate(lambda: layer(x)))
    plan = layer._get_plan()
    w = layer.kernel.detach()
    ws = layer._workspace
    out = torch.empty_like(y)
    f = lambda: _native.cheb_forward(plan, x, w, None, 5, precision=_native.PREC_BF16X6, algo=_native.ALGO_AUTO, workspace=ws, out=out,
                                     keep_weights=True)
    print("_native.cheb_forward: submit %.1f us, with drain %.1f us" % rate(f))
    lib = _native.lib()
    import ctypes
    args = (plan.handle, _native._ptr(x), _native._ptr(w), _native._ptr(None), _native._ptr(out), 1, 1, 16, 5, 0, 0,
            int(_native.PREC_BF16X6), int(_native.ALGO_AUTO), 0, int(_native.FWD_KEEP_WEIGHTS), _native._ptr(ws),
            ws.numel(), _native._stream_ptr(dev))
    g = lambda: lib.dsph_poly_forward_ex(*args)
    print("bare C call:          submit %.1f us, with drain %.1f us" % rate(g))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(2000):
        layer(x)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
