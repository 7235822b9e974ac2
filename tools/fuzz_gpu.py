#!/usr/bin/env python3
"""Randomised cross-check on the GPU: fused vs unfused forward, planes, weight gradient (fp32 and bf16x3) on
random graphs / shapes.  Usage: fuzz_gpu.py [cases] [seed]
FUZZ_STRIPS=1: every case has the strip kernel's shape (K 5, 64 input channels, 64 / 128 / 192 output columns, nside 64 / 128)
and the plans are built with the option DSPH_OPT_STRIPS = always, so that the strip kernel runs whatever the cost gate says.
Tolerances (of max|y|, the ones DESIGN.md section 2 states): exact fp32 and the six-term split 2e-6; the three-term split 1e-5
with 16 or more input channels -- where the layers use it -- and 2e-5 below; behind tanh five times that (the reference scale
shrinks to <= 1 while the pre-activation's error passes through with slope <= 1)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "deepsphere-cosmo-tf2_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from deepsphere import _native, healpix, utils  # noqa: E402
from oracle import cheb_oracle as orc  # noqa: E402

STRIPS = os.environ.get("FUZZ_STRIPS") == "1"
# FUZZ_ISTRIPS=1 (round 4): shapes of the input-side strip kernel (at most 16 input channels, K 2..5, any width) on maps with
# strip rectangles (nside 128), and, every third case, K 6..13 through the chain of passes (DSPH_OPT_SPLIT = always)
ISTRIPS = os.environ.get("FUZZ_ISTRIPS") == "1"
PLAN_OPTIONS = {_native.OPT_STRIPS: _native.STRIPS_ALWAYS} if STRIPS else ({_native.OPT_SPLIT: _native.SPLIT_ALWAYS} if ISTRIPS else None)
NSIDES = [int(v) for v in os.environ.get("FUZZ_NSIDES", "64,128" if STRIPS else ("128" if ISTRIPS else "8,16,32")).split(",")]
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).cuda()  # noqa: E731
rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))  # noqa: E731
plans = {}
bad = with_strips = 0
for it in range(cases):
    nside = int(rng.choice(NSIDES))
    mode = str(rng.choice(["grid", "cap"] if (STRIPS or ISTRIPS) else ["grid", "knn", "cap"]))
    frac = float(rng.uniform(0.1, 0.9))
    basis = int(rng.choice([_native.BASIS_CHEBYSHEV, _native.BASIS_MONOMIAL]))
    key = (nside, mode, round(frac, 1) if mode == "cap" else 0, basis)
    if key not in plans:
        if mode == "cap":
            idx = healpix.extend_indices(healpix.cap_indices(nside, fraction=key[2]), nside, max(1, nside // 4))
            idx = idx[: len(idx) - int(rng.integers(0, 50))]
            L = healpix.healpix_laplacian(nside, indices=idx, mode="grid")
        else:
            L = healpix.healpix_laplacian(nside, mode=mode)
        Lt, _ = orc.prepare_L(L, scale=0.75 if basis == _native.BASIS_CHEBYSHEV else 1.0)
        cols, vals = utils.csr_to_ell(Lt)
        plans[key] = (Lt, _native.LaplacianPlan(cols, vals, device=0, options=PLAN_OPTIONS))
    Lt, plan = plans[key]
    M = Lt.shape[0]
    K = int(rng.integers(2, 11))  # (10: one pass over 9-ring regions on the grid since round 6, the chain of passes elsewhere)
    Fin = 4 * int(rng.integers(1, 19)) if rng.random() < 0.6 else int(rng.integers(1, 40))  # also channel counts that get padded
    Fout = int(rng.integers(1, 141))
    N = int(rng.integers(1, 4))
    if STRIPS:
        K, Fin, Fout = 5, 64, 64 * int(rng.integers(1, 4))
    if ISTRIPS:
        K = int(rng.integers(2, 6)) if it % 3 else int(rng.integers(6, 14))
        Fin = int(rng.integers(1, 17)) if rng.random() < 0.7 else int(rng.integers(1, 3))  # (1, 2: the level-packed kernel)
        Fout = 4 * int(rng.integers(1, 26)) if rng.random() < 0.8 else int(rng.integers(1, 100))
    if not plan.fused_ok(Fin, Fout, K):
        continue
    x = rng.standard_normal((N, M, Fin)).astype(np.float32)
    W = (rng.standard_normal((Fin * K, Fout)) / np.sqrt(Fin * K)).astype(np.float32)
    b = rng.standard_normal(Fout).astype(np.float32)
    dy = rng.standard_normal((N, M, Fout)).astype(np.float32)
    act = int(rng.choice([_native.ACT_NONE, _native.ACT_RELU, _native.ACT_TANH]))
    xd, Wd, bd, dyd = dev(x), dev(W), dev(b), dev(dy)
    yu, _ = _native.cheb_forward(plan, xd, Wd, bd, K, act=act, algo=_native.ALGO_UNFUSED, basis=basis)
    yf, _ = _native.cheb_forward(plan, xd, Wd, bd, K, act=act, algo=_native.ALGO_FUSED, basis=basis, precision=_native.PREC_FP32)
    yb, _ = _native.cheb_forward(plan, xd, Wd, bd, K, act=act, algo=_native.ALGO_FUSED, basis=basis, precision=_native.PREC_BF16X3)
    y6, _ = _native.cheb_forward(plan, xd, Wd, bd, K, act=act, algo=_native.ALGO_FUSED, basis=basis, precision=_native.PREC_BF16X6)
    e1, e2 = rel(yf.cpu().numpy(), yu.cpu().numpy()), rel(yb.cpu().numpy(), yu.cpu().numpy())
    e6 = rel(y6.cpu().numpy(), yu.cpu().numpy())
    same = True
    if Fin % 4 == 0 and K <= 10:  # (the planes / weight-gradient modes of the BFS kernel take whole 16-byte pieces only)
        try:
            pf = _native.cheb_planes(plan, xd, K, basis=basis, algo=_native.ALGO_FUSED)
            pu = _native.cheb_planes(plan, xd, K, basis=basis, algo=_native.ALGO_UNFUSED)
            same = all(torch.equal(a, b2) for a, b2 in zip(pf, pu))
        except RuntimeError as exc:  # a forward that runs as a chain of passes has no fused planes mode: refused loudly, fine
            if "cannot run" not in str(exc):
                raise
    du, _ = _native.cheb_backward_weights(plan, xd, dyd, K, basis=basis, algo=_native.ALGO_UNFUSED)
    e3 = e4 = -1.0
    try:
        if Fin % 4 != 0 or K > 10:
            raise RuntimeError("fused weight gradient cannot run: Fin % 4, or more than ten terms")
        df, _ = _native.cheb_backward_weights(plan, xd, dyd, K, basis=basis, algo=_native.ALGO_FUSED)
        dbf, _ = _native.cheb_backward_weights(plan, xd, dyd, K, basis=basis, algo=_native.ALGO_FUSED, precision=_native.PREC_BF16X3)
        e3, e4 = rel(df.cpu().numpy(), du.cpu().numpy()), rel(dbf.cpu().numpy(), du.cpu().numpy())
    except RuntimeError as exc:  # accumulators do not fit beside the planes: refused loudly, fine
        if "cannot run" not in str(exc):
            raise
    loose = 5.0 if act == _native.ACT_TANH else 1.0
    t_exact, t_x3 = 2e-6 * loose, (1e-5 if Fin >= 16 else 2e-5) * loose * (2.0 if K > 9 else 1.0)
    ok = e1 < t_exact and e6 < t_exact and e2 < t_x3 and same and e3 < 2e-5 and e4 < (2e-5 if Fin >= 16 else 1e-4)  # (dW in the three-term split: DESIGN 4.1)
    n_strip = plan.strip_tiles(Fin, Fout, K, _native.PREC_BF16X3)
    with_strips += n_strip > 0
    bad += not ok
    print(f"{'ok ' if ok else 'BAD'} nside={nside} {mode} M={M} K={K} {Fin}->{Fout} N={N} act={act} basis={basis} strip_tiles={n_strip}: fwd {e1:.1e} {e2:.1e} {e6:.1e} planes {same} dW {e3:.1e} {e4:.1e}",
          flush=True)
print("FAILED" if bad else "ALL OK", bad, f"({with_strips} cases with strip tiles)")
sys.exit(1 if bad else 0)
