#!/usr/bin/env python3
"""Quick GPU check of the structured-tile kernel: fused (class-R tiles via cheb_struct_kernel) against the unfused
kernels and against the float64 oracle on grid-stencil graphs; then timings.  tools/check_struct.py [nside ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deepsphere-cosmo-tf2_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch

from deepsphere import _native, healpix, utils
import bench


def case(nside, K, Fin, Fout, N, prec, oracle=True, act=0, bias=False):
    dev = torch.device("cuda", 0)
    cols, vals, lmax = bench.build_laplacian(nside, dev)
    M = cols.shape[0]
    plan = _native.LaplacianPlan(cols, vals, device=0)
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn((N, M, Fin), device=dev, generator=g)
    w = torch.randn((Fin * K, Fout), device=dev, generator=g) / np.sqrt(Fin * (K + 0.5) / 2)
    b = torch.randn((Fout,), device=dev, generator=g) if bias else None
    P = {"bf16x3": _native.PREC_BF16X3, "bf16x6": _native.PREC_BF16X6}.get(prec, _native.PREC_FP32)
    ok = (plan.fused_ok(Fin, Fout, K), plan.tile_counts(K))
    y_f, _ = _native.cheb_forward(plan, x, w, b, K, act=act, precision=P, algo=_native.ALGO_FUSED)
    y_u, _ = _native.cheb_forward(plan, x, w, b, K, act=act, precision=_native.PREC_FP32, algo=_native.ALGO_UNFUSED)
    torch.cuda.synchronize()
    s = float(y_u.abs().max())
    err = float((y_f - y_u).abs().max()) / s
    bad = int(((y_f - y_u).abs() > 1e-4 * s).sum())
    msg = f"nside {nside} K {K} {Fin}->{Fout} N {N} {prec} act {act} bias {bias}: fused_ok {ok} max|fused-unfused|/s = {err:.3e}  bad {bad}"
    if oracle and M <= 200000:
        from scipy import sparse
        from oracle import cheb_oracle as orc
        Wd = cols.shape[1]
        Lt = sparse.csr_matrix((vals.reshape(-1).astype(np.float64), cols.reshape(-1), np.arange(0, Wd * M + 1, Wd)), shape=(M, M))
        y_o = orc.chebyshev_forward(Lt, x.cpu().numpy().astype(np.float64), w.cpu().numpy().astype(np.float64), K,
                                    bias=None if b is None else b.cpu().numpy().astype(np.float64),
                                    activation={0: None, 1: "relu"}[act])
        so = np.abs(y_o).max()
        msg += f"  vs oracle {np.abs(y_f.cpu().numpy() - y_o).max() / so:.3e}"
    print(msg, flush=True)
    if bad:
        d = (y_f - y_u).abs().amax(dim=(0, 2)).cpu().numpy()
        rows = np.nonzero(d > 1e-4 * s)[0]
        print("   bad rows:", rows[:20], "... count", len(rows), " tiles:", np.unique(rows // 256)[:20])
    plan.close()
    return err


def timing(nside, K, Fin, Fout, N, prec, reps=10):
    dev = torch.device("cuda", 0)
    cols, vals, lmax = bench.build_laplacian(nside, dev)
    M = cols.shape[0]
    plan = _native.LaplacianPlan(cols, vals, device=0)
    x = torch.randn((N, M, Fin), device=dev)
    w = torch.randn((Fin * K, Fout), device=dev) / np.sqrt(Fin * (K + 0.5) / 2)
    P = {"bf16x3": _native.PREC_BF16X3, "bf16x6": _native.PREC_BF16X6}.get(prec, _native.PREC_FP32)
    ws = None
    out = None
    print("tiles (struct, bfs):", plan.tile_counts(K), flush=True)
    for _ in range(3):
        out, ws = _native.cheb_forward(plan, x, w, None, K, precision=P, algo=_native.ALGO_FUSED, workspace=ws, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        _native.cheb_forward(plan, x, w, None, K, precision=P, algo=_native.ALGO_FUSED, workspace=ws, out=out)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    balg = bench.algorithmic_bytes(N, M, Fin, Fout, K, 9)
    print(f"TIMING nside {nside} K {K} {Fin}->{Fout} N {N} {prec}: {ms:.3f} ms  roofline frac {balg / ms / 1e6 / 8000:.4f}", flush=True)
    plan.close()


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "all"
    if mode in ("all", "parity"):
        case(32, 5, 16, 32, 2, "fp32")
        case(32, 5, 16, 32, 2, "bf16x3")
        case(64, 5, 64, 64, 2, "fp32")
        case(64, 5, 64, 64, 3, "bf16x3", act=1, bias=True)
        case(64, 3, 32, 16, 2, "fp32")
        case(64, 2, 8, 64, 1, "fp32")
        case(64, 4, 20, 40, 2, "bf16x3")
        case(128, 5, 64, 64, 1, "bf16x3", oracle=False)
    if mode == "dbg":
        case(128, 5, 64, 64, 4, "fp32", oracle=False)
        case(256, 5, 64, 64, 2, "fp32", oracle=False)
        case(256, 5, 64, 64, 4, "fp32", oracle=False)
        case(256, 4, 64, 64, 4, "bf16x3", oracle=False)
        case(256, 5, 16, 32, 8, "bf16x3", oracle=False)
    if mode in ("all", "time"):
        timing(256, 5, 16, 32, 8, "bf16x3")
        timing(512, 5, 64, 64, 4, "bf16x3")
        timing(1024, 5, 64, 64, 4, "bf16x3", reps=10)
        timing(1024, 5, 64, 64, 4, "fp32", reps=5)
