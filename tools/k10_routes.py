#!/usr/bin/env python3
"""K = 10 (the reference tutorials' order) on the two routes: the chain of K <= 5 passes (csrc/cheb_split.hip, DSPH_OPT_SPLIT = always)
and one pass of the breadth-first tile kernel over 9-ring regions (1,156 rows in planes of 1,168: round 6, DSPH_OPT_SPLIT = never),
timed with HIP events on one GPU and checked against the float64 patch oracle on a few rows."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "deepsphere-cosmo-tf2_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from deepsphere import _native  # noqa: E402
from oracle import cheb_oracle as orc  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    K = 10
    for nside, N, Fin, Fout in ((256, 8, 16, 32), (256, 4, 64, 64), (512, 2, 32, 32), (128, 8, 8, 8)):
        cols, vals, _ = bench.build_laplacian(nside, dev)
        M = cols.shape[0]
        rng = np.random.default_rng(nside + Fin)
        x = torch.from_numpy(rng.standard_normal((N, M, Fin)).astype(np.float32)).to(dev)
        W = torch.from_numpy((rng.standard_normal((Fin * K, Fout)) * orc.default_kernel_stddev(Fin, K)).astype(np.float32)).to(dev)
        b = torch.from_numpy(rng.standard_normal(Fout).astype(np.float32)).to(dev)
        centres = rng.integers(0, M, size=24)
        ref, s = None, None
        for route, opt in (("chain", _native.SPLIT_ALWAYS), ("one pass", _native.SPLIT_NEVER)):
            for prec, P in (("bf16x3", _native.PREC_BF16X3), ("bf16x6", _native.PREC_BF16X6)):
                plan = _native.LaplacianPlan(cols, vals, device=0, options={_native.OPT_SPLIT: opt})
                y, ws = _native.cheb_forward(plan, x, W, b, K, act=_native.ACT_RELU, precision=P)
                for _ in range(3):
                    _native.cheb_forward(plan, x, W, b, K, act=_native.ACT_RELU, precision=P, workspace=ws, out=y)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                steps = 20
                e0.record()
                for _ in range(steps):
                    _native.cheb_forward(plan, x, W, b, K, act=_native.ACT_RELU, precision=P, workspace=ws, out=y)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / steps
                if ref is None:
                    from test_gpu_round2 import _patch_reference
                    ref = _patch_reference(cols, vals, x[:1], W.cpu().numpy(), K, centres, bias=b.cpu().numpy(), activation="relu")
                    s = float(np.abs(ref).max())
                err = float(np.abs(y[:1, torch.as_tensor(centres).to(dev)].cpu().numpy() - ref).max() / s)
                print(f"nside {nside} N {N} {Fin}->{Fout} K {K} {route:8s} {prec}: {ms:.3f} ms, err {err:.2e}", flush=True)
                del plan


if __name__ == "__main__":
    main()
