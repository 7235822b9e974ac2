"""Round-6 GPU tests (all through the C ABI): the f16 three-term split at small input scales and in the backward,
inference batch norm folded into the kernel epilogue, `bench.py --gpus N` started as given (item 3)."""

import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from deepsphere import _native, gnn_layers
from helpers import rel_err
from oracle import cheb_oracle as orc
from test_gpu_round3 import _csr, _dev, _grid_ell

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL_FP32_EQUIV = 2e-6  # exact fp32, the six-term bf16 split and the f16 three-term split measure 2 - 7e-7


def _layer(cols, vals, K, Fout, W, **kw):
    return gnn_layers.Chebyshev.from_prepared_ell(
        cols, vals, K, Fout=Fout, device="cuda:0", initializer=lambda t: t.copy_(torch.from_numpy(W)),
        plan_options={_native.OPT_STRIPS: _native.STRIPS_ALWAYS}, **kw)


@pytest.mark.parametrize("scale", [1e-3, 1.0, 3e4])
def test_f16_split_is_fp32_equivalent_at_every_input_scale(scale):
    """DSPH_PREC_F16X3 with the input's power-of-two factor (DSPH_OPT_F16_XEXP): 3e-7 of max|y| whatever the scale of x --
    from the caller's bound (x_absmax) and from the layer's own reduction.  Without the factor an input of scale 1e-3 loses its
    lo halves to f16 subnormals (measured 1.8e-5: the lower range condition of include/dsphere.h), and one of scale 3e4 overflows."""
    K, Fin, Fout, nside, N = 5, 64, 64, 128, 2
    cols, vals = _grid_ell(nside)
    M = cols.shape[0]
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((N, M, Fin)) * scale).astype(np.float32)
    W = (rng.standard_normal((Fin * K, Fout)) * orc.default_kernel_stddev(Fin, K)).astype(np.float32)
    ref = orc.chebyshev_forward(_csr(cols, vals), x, W, K)
    for hint in (None, float(np.abs(x).max())):
        layer = _layer(cols, vals, K, Fout, W, precision="f16x3", x_absmax=hint)
        with torch.no_grad():
            y = layer(_dev(x))
        assert layer._get_plan().strip_tiles(Fin, Fout, K, _native.PREC_F16X3, N=N) > 0, "the quad strips run this"
        err = rel_err(y.cpu().numpy(), ref)
        print(f"f16x3 scale {scale:g} x_absmax {hint}: exponent {layer._f16_xexp}, rel err {err:.2e}")
        assert err < TOL_FP32_EQUIV
    # the C ABI without the factor: what the header says of the range condition
    plan = _native.LaplacianPlan(cols, vals, device=0, options={_native.OPT_STRIPS: _native.STRIPS_ALWAYS})
    plan.prepare(K, Fin)
    y0, _ = _native.cheb_forward(plan, _dev(x), _dev(W), None, K, precision=_native.PREC_F16X3, algo=_native.ALGO_FUSED)
    e0 = rel_err(y0.cpu().numpy(), ref) if torch.isfinite(y0).all() else float("inf")
    print(f"f16x3 scale {scale:g} without the factor: {e0:.2e}")
    if scale == 1e-3:
        assert e0 > 5e-6
    if scale == 3e4:
        assert not np.isfinite(e0), "beyond the f16 range: loud"
    plan.set_option(_native.OPT_F16_XEXP, 3)  # any exponent is exact to apply: same bits up to the range
    if scale == 1.0:
        y3, _ = _native.cheb_forward(plan, _dev(x), _dev(W), None, K, precision=_native.PREC_F16X3, algo=_native.ALGO_FUSED)
        assert rel_err(y3.cpu().numpy(), ref) < TOL_FP32_EQUIV
    with pytest.raises(ValueError):
        plan.set_option(_native.OPT_F16_XEXP, 1000)


def test_f16_layer_backward_with_tiny_upstream_gradients():
    """under precision="f16x3" the input gradient ran the quad strips on dy split into f16 as it is; a mean loss
    over millions of pixels gives dy ~ 1e-8 (below the smallest f16 subnormal): dx came out zero on the strips' pixels.  The dx
    contraction now runs the six-term bf16 split, the weight gradient exact fp32."""
    K, Fin, Fout, nside, N = 5, 64, 64, 128, 1
    cols, vals = _grid_ell(nside)
    M = cols.shape[0]
    rng = np.random.default_rng(6)
    x = rng.standard_normal((N, M, Fin)).astype(np.float32)
    W = (rng.standard_normal((Fin * K, Fout)) * orc.default_kernel_stddev(Fin, K)).astype(np.float32)
    dy = (rng.standard_normal((N, M, Fout)) * 1e-8).astype(np.float32)
    layer = _layer(cols, vals, K, Fout, W, precision="f16x3")
    xt = _dev(x).requires_grad_(True)
    y = layer(xt)
    y.backward(_dev(dy))
    dx_ref, dk_ref = orc.chebyshev_backward(_csr(cols, vals), x, W, K, dy)
    ex = rel_err(xt.grad.cpu().numpy(), dx_ref)
    ek = rel_err(layer.kernel.grad.cpu().numpy(), dk_ref)
    print(f"f16x3 backward, dy ~ 1e-8: dx {ex:.2e}, dkernel {ek:.2e}")
    assert ex < TOL_FP32_EQUIV and ek < TOL_FP32_EQUIV
    assert gnn_layers.resolve_dx_precision("f16x3", Fout, K) == "bf16x6"
    assert gnn_layers.resolve_wgrad_precision("f16x3", N * M) == "fp32"


@pytest.mark.parametrize("shape", [(32, 5, 16, 32, 3), (128, 5, 64, 64, 2)])
def test_inference_batch_norm_is_folded_into_the_kernel_epilogue(shape):
    """quick_start's layer (use_bn=True, use_bias=True, activation="elu", reference gnn_layers.py:152-159: BN -> bias -> act) in
    inference: ONE kernel call -- the moving statistics folded into weights and bias -- against the oracle's bn=(mean, var);
    neither the host framework's batch norm nor its activation runs.  A change of the statistics is noticed (version counters)."""
    nside, K, Fin, Fout, N = shape
    cols, vals = _grid_ell(nside)
    M = cols.shape[0]
    rng = np.random.default_rng(nside)
    x = rng.standard_normal((N, M, Fin)).astype(np.float32)
    W = (rng.standard_normal((Fin * K, Fout)) * orc.default_kernel_stddev(Fin, K)).astype(np.float32)
    layer = gnn_layers.Chebyshev.from_prepared_ell(cols, vals, K, Fout=Fout, device="cuda:0", use_bn=True, use_bias=True, activation="elu",
                                                   initializer=lambda t: t.copy_(torch.from_numpy(W)))
    layer.build((N, M, Fin))
    b = layer.bias.detach().cpu().numpy().reshape(-1)
    for trial in range(2):
        mean = rng.standard_normal(Fout).astype(np.float32) * 0.3
        var = (0.5 + rng.random(Fout)).astype(np.float32)
        with torch.no_grad():
            layer.bn.running_mean.copy_(torch.from_numpy(mean))
            layer.bn.running_var.copy_(torch.from_numpy(var))
        ref = orc.chebyshev_forward(_csr(cols, vals), x, W, K, bias=b, activation="elu", bn=(mean, var))

        def boom(*a, **k):
            raise AssertionError("the host framework's epilogue ran in inference")
        bn_forward, act = layer.bn.forward, layer.activation
        layer.bn.forward, layer.activation = boom, boom
        try:
            with torch.no_grad():
                y = layer(_dev(x))            # training=False: the reference's default call
                y2 = layer(_dev(x))           # steady state: kept weight images of the folded kernel
        finally:
            layer.bn.forward, layer.activation = bn_forward, act
        err = rel_err(y.cpu().numpy(), ref)
        print(f"BN folded, nside {nside} {Fin}->{Fout} trial {trial}: rel err {err:.2e}")
        assert err < 2e-5  # (the three-term bf16 split's tolerance, TOL_BF16X3 of the other tests: max|y| after the ELU is small)
        assert torch.equal(y, y2)
    # ... and with the pooling in the same pass (dsph_poly_forward_pool) where the kernels can: HealpyPool(MAX)(layer(x)), bit for bit
    if layer._act_code in (_native.ACT_NONE, _native.ACT_RELU):
        pass  # (this layer's ELU is not a pooled epilogue; the ReLU variant below)
    relu = gnn_layers.Chebyshev.from_prepared_ell(cols, vals, K, Fout=Fout, device="cuda:0", use_bn=True, use_bias=True, activation="relu",
                                                  initializer=lambda t: t.copy_(torch.from_numpy(W)))
    relu.build((N, M, Fin))
    with torch.no_grad():
        relu.bn.running_mean.copy_(layer.bn.running_mean)
        relu.bn.running_var.copy_(layer.bn.running_var)
        yp = relu.forward_pool(_dev(x), "MAX")
        yf = relu(_dev(x))
    if yp is not None:
        assert torch.equal(yp, yf.reshape(N, M // 4, 4, Fout).amax(dim=2)), "conv + BN + bias + ReLU + pool in one pass"
        print(f"BN folded, pooled epilogue: {tuple(yp.shape)}")
    # batch statistics (training=True) stay in the host framework: same layer, reference semantics
    with torch.no_grad():
        yt = layer(_dev(x), training=True)
    lin = orc.chebyshev_forward(_csr(cols, vals), x, W, K)
    mu, vv = lin.mean(axis=(0, 1)), lin.var(axis=(0, 1))
    ref_t = orc.chebyshev_forward(_csr(cols, vals), x, W, K, bias=b, activation="elu", bn=(mu, vv))
    assert rel_err(yt.cpu().numpy(), ref_t) < 2e-5
    # ... and moved the moving statistics: the next inference call folds the new ones
    with torch.no_grad():
        y3 = layer(_dev(x))
    m2, v2 = layer.bn.running_mean.cpu().numpy(), layer.bn.running_var.cpu().numpy()
    ref3 = orc.chebyshev_forward(_csr(cols, vals), x, W, K, bias=b, activation="elu", bn=(m2, v2))
    assert rel_err(y3.cpu().numpy(), ref3) < 2e-5


@pytest.mark.parametrize("nside,N,act", [(128, 1, "relu"), (256, 3, None)])
def test_k8_quad_strips_whole_map(nside, N, act):
    """K = 8, 32 -> 32 (BASELINE configs[3]'s shape) on the three-role quad strips (csrc/cheb_qstrip8_kernel.h):
    whole maps against the float64 oracle -- every row: the strips' seams in x, the run-time cuts of the tape, the rectangles' edges
    against the breadth-first tiles that keep the base-pixel borders -- and against the same plan with the strips switched off
    (the tile kernel sums in forward order: equal to rounding)."""
    K, Fin, Fout = 8, 32, 32
    cols, vals = _grid_ell(nside)
    M = cols.shape[0]
    rng = np.random.default_rng(nside + N)
    x = rng.standard_normal((N, M, Fin)).astype(np.float32)
    W = (rng.standard_normal((Fin * K, Fout)) * orc.default_kernel_stddev(Fin, K)).astype(np.float32)
    b = rng.standard_normal(Fout).astype(np.float32)
    ref = orc.chebyshev_forward(_csr(cols, vals), x, W, K, bias=b, activation=act)
    kw = dict(act=_native.ACT_RELU if act == "relu" else _native.ACT_NONE, precision=_native.PREC_BF16X3, algo=_native.ALGO_FUSED)
    ys = {}
    for strips in (_native.STRIPS_ALWAYS, _native.STRIPS_NEVER):
        plan = _native.LaplacianPlan(cols, vals, device=0, options={_native.OPT_STRIPS: strips})
        plan.prepare(K, Fin, Fout=Fout)
        n_strip = plan.strip_tiles(Fin, Fout, K, _native.PREC_BF16X3, N=N)
        nt = nside // 16
        if strips == _native.STRIPS_ALWAYS:
            assert 12 * (nt - 2) ** 2 <= n_strip <= 12 * nt * nt - 24, "the tiles whose 7-ring region stays inside their base pixel, and translated border tiles"
            rec = plan.strip_pairs(K)
            assert rec.shape[1] == 12 and np.all(rec[:, 2] <= 48) and np.all(rec[:, 0] - rec[:, 4] == 8), "48 output columns behind 8 of lead-in"
        else:
            assert n_strip == 0
        y, _ = _native.cheb_forward(plan, _dev(x), _dev(W), _dev(b), K, **kw)
        y2, _ = _native.cheb_forward(plan, _dev(x), _dev(W), _dev(b), K, **kw)
        assert torch.equal(y, y2), "two launches of the same inputs must agree bit for bit"
        err = rel_err(y.cpu().numpy(), ref)
        print(f"K 8 strips {strips} nside {nside} N {N}: {n_strip} strip tiles, rel err {err:.2e}")
        assert err < 1e-5
        ys[strips] = y
    assert rel_err(ys[_native.STRIPS_ALWAYS].cpu().numpy(), ys[_native.STRIPS_NEVER].cpu().numpy()) < 2e-5
    # the fp32-equivalent f16 three-term split on the same strips (x times 2^11: max |x| ~ 5 lands in [2^13, 2^14))
    plan = _native.LaplacianPlan(cols, vals, device=0, options={_native.OPT_STRIPS: _native.STRIPS_ALWAYS, _native.OPT_F16_XEXP: 11})
    plan.prepare(K, Fin, Fout=Fout)
    assert plan.strip_tiles(Fin, Fout, K, _native.PREC_F16X3, N=N) > 0
    kw["precision"] = _native.PREC_F16X3
    y16, _ = _native.cheb_forward(plan, _dev(x), _dev(W), _dev(b), K, **kw)
    e16 = rel_err(y16.cpu().numpy(), ref)
    print(f"K 8 f16x3 nside {nside} N {N}: rel err {e16:.2e}")
    assert e16 < TOL_FP32_EQUIV


def test_k8_layer_wider_than_the_strips_shape_stays_on_the_tile_kernel():
    """K = 8, 32 -> 96: the last 32 columns of a wider layer have the strips' shape but no room for their weight image in a
    workspace cut into 64-column blocks -- the whole layer runs on the tile kernel (and says so), correctly."""
    K, Fin, Fout, nside, N = 8, 32, 96, 128, 1
    cols, vals = _grid_ell(nside)
    M = cols.shape[0]
    rng = np.random.default_rng(96)
    x = rng.standard_normal((N, M, Fin)).astype(np.float32)
    W = (rng.standard_normal((Fin * K, Fout)) * orc.default_kernel_stddev(Fin, K)).astype(np.float32)
    plan = _native.LaplacianPlan(cols, vals, device=0, options={_native.OPT_STRIPS: _native.STRIPS_ALWAYS})
    plan.prepare(K, Fin, Fout=Fout)
    assert plan.strip_tiles(Fin, Fout, K, _native.PREC_BF16X3, N=N) == 0
    y, _ = _native.cheb_forward(plan, _dev(x), _dev(W), None, K, precision=_native.PREC_BF16X3, algo=_native.ALGO_FUSED)
    assert rel_err(y.cpu().numpy(), orc.chebyshev_forward(_csr(cols, vals), x, W, K)) < 1e-5


def test_finely_padded_mask_runs_on_the_strips():
    """A survey mask padded the way a network pads it (utils.extend_indices to the nside of its coarsest layer -- nside-32 superpixels:
    2 x 2 tiles each, no tile's 4-ring region inside its superpixel): until round 6 every tile of such a map was class T or G; the quad
    strips' rectangles on the logical tile grid now take most of them (bench.py --config c5s: 21.6 -> 15.7 ms).  The patch oracle on
    rows of the mask border, on the strips' seams (through dsph_plan_strip_rows: every one crosses superpixel borders) and at random."""
    import bench
    from test_gpu_round2 import _headline_check, _patch_reference, _strip_seam_rows, _tape_cut_rows

    nside, N, Fin, Fout, K = 1024, 4, 64, 64, 5
    cols, vals, _ = bench.build_laplacian_masked(nside, torch.device("cuda", 0), nside_super=32)
    M = cols.shape[0]
    plan, x, W, b, y, s = _headline_check(cols, vals, N, Fin, Fout, K, _native.PREC_BF16X3, seed=32)
    n_struct, n_bfs = plan.tile_counts(K)
    n_strip = plan.strip_tiles(Fin, Fout, K, _native.PREC_BF16X3, N=N)
    assert n_strip > 0.9 * n_struct, f"{n_strip} of {n_struct} structured tiles on the strips"
    pairs = plan.strip_pairs(K)
    deg = (vals != 0).sum(axis=1)
    border = np.nonzero(deg < 9)[0]
    rng = np.random.default_rng(32)
    seams = _strip_seam_rows(plan, K, pairs, stride=max(1, len(pairs) // 40))
    cuts = _tape_cut_rows(plan, K, pairs, N)[0][::4]
    centres = np.unique(np.concatenate([border[rng.integers(0, border.size, size=60)], rng.integers(0, M, size=60), seams[::2], cuts]))
    centres = centres[centres < M]
    ref = _patch_reference(cols, vals, x[:2], W, K, centres, bias=b, activation="relu")
    err = float(np.abs(y[:2, torch.as_tensor(centres).cuda()].cpu().numpy() - ref).max() / s)
    print(f"c5s: M {M}, tiles: {n_strip} strips on {len(pairs)} records, {n_struct - n_strip} structured, {n_bfs} class G; {centres.size} centres, err {err:.2e}")
    assert err < 1e-5


@pytest.mark.parametrize("drop", [100, 200, 256 + 37])
def test_quad_strips_on_a_map_with_an_incomplete_last_tile(drop):
    """The K = 5 quad-strip kernel reads a row as it comes (no clamp to the strip's halo: a run of steps touches a few rows of the
    table's ring tiles past the halo, for nothing) -- so those rows must exist, and the plan keeps every tile beside the map's last,
    incomplete tile off the strips (cheb_fused.hip, `barred`).  A sphere cut off `drop` pixels before its end: whole maps against
    the float64 oracle, every row."""
    nside, K, Fin, Fout, N = 128, 5, 64, 64, 2
    cols, vals = _grid_ell(nside)
    M = cols.shape[0] - drop
    cols, vals = cols[:M].copy(), vals[:M].copy()
    gone = cols >= M
    vals[gone] = 0.0
    cols[gone] = np.broadcast_to(np.arange(M, dtype=cols.dtype)[:, None], cols.shape)[gone]
    rng = np.random.default_rng(drop)
    x = rng.standard_normal((N, M, Fin)).astype(np.float32)
    W = (rng.standard_normal((Fin * K, Fout)) * orc.default_kernel_stddev(Fin, K)).astype(np.float32)
    b = rng.standard_normal(Fout).astype(np.float32)
    ref = orc.chebyshev_forward(_csr(cols, vals), x, W, K, bias=b, activation="relu")
    plan = _native.LaplacianPlan(cols, vals, device=0, options={_native.OPT_STRIPS: _native.STRIPS_ALWAYS})
    plan.prepare(K, Fin, Fout=Fout)
    n_strip = plan.strip_tiles(Fin, Fout, K, _native.PREC_BF16X3, N=N)
    assert n_strip > 0
    last = (M - 1) // 256  # the incomplete tile: no strip record's table may lead a pixel of its rectangle, halo or ring rows there ...
    rec = plan.strip_pairs(K)
    for s_ in range(rec.shape[0]):
        r = rec[s_]
        xs, ys = np.arange(int(r[8]), int(r[9]) + 1), np.arange(int(r[10]) - 1, int(r[11]) + 7)
        X, Y = np.meshgrid(xs, ys, indexing="ij")
        rows = plan.strip_rows(K, s_, X.ravel(), Y.ravel())
        assert rows.min() >= 0 and rows.max() < M, "a row the kernel reads does not exist"
    y, _ = _native.cheb_forward(plan, _dev(x), _dev(W), _dev(b), K, act=_native.ACT_RELU, precision=_native.PREC_BF16X3, algo=_native.ALGO_FUSED)
    err = rel_err(y.cpu().numpy(), ref)
    print(f"sphere without its last {drop} pixels: {n_strip} strip tiles on {rec.shape[0]} records (last tile {last}), rel err {err:.2e}")
    assert err < 1e-5


def test_k10_routes_by_graph():
    """K = 10 (the reference tutorials' order): ONE pass of the breadth-first tile kernel over 9-ring regions on the 8-neighbour grid
    (1,156 rows in planes of 1,168, instantiated for ELL width 9), the chain of <= 5-term passes on the reference's k-NN graph (width
    11 -> 12: its regions have no 1,168-row variant -- a fuzz case found the planes mode asking for one); both against the float64
    oracle, and the planes of the recurrence wherever the library offers them."""
    import bench

    K, Fin, Fout, N = 10, 16, 8, 2
    dev = torch.device("cuda", 0)
    for graph in ("grid", "knn"):
        cols, vals = (_grid_ell(32) if graph == "grid" else bench.build_laplacian_knn(32, dev, 8)[:2])
        M = cols.shape[0]
        rng = np.random.default_rng(10 + len(graph))
        x = rng.standard_normal((N, M, Fin)).astype(np.float32)
        W = (rng.standard_normal((Fin * K, Fout)) * orc.default_kernel_stddev(Fin, K)).astype(np.float32)
        plan = _native.LaplacianPlan(cols, vals, device=0)
        assert plan.fused_ok(Fin, Fout, K)
        assert plan.uses_chain(Fin, Fout, K) == (graph == "knn")
        ref = orc.chebyshev_forward(_csr(cols, vals), x, W, K)
        for prec, tol in ((_native.PREC_BF16X3, 1e-5 if graph == "grid" else 2e-5), (_native.PREC_BF16X6, TOL_FP32_EQUIV)):
            y, _ = _native.cheb_forward(plan, _dev(x), _dev(W), None, K, precision=prec)
            err = rel_err(y.cpu().numpy(), ref)
            print(f"K 10 on the {graph} graph (ELL width {cols.shape[1]}), precision {prec}: rel err {err:.2e}")
            assert err < tol
        pu = _native.cheb_planes(plan, _dev(x), K, algo=_native.ALGO_UNFUSED)
        try:
            pf = _native.cheb_planes(plan, _dev(x), K, algo=_native.ALGO_FUSED)
            assert all(torch.equal(a, b) for a, b in zip(pf, pu))
        except RuntimeError as exc:  # refused loudly where the fused planes mode does not exist (the chain's graph)
            assert graph == "knn" and "cannot run" in str(exc), str(exc)


def test_headline_shape_on_the_reference_graph_as_benchmarked():
    """the headline shape (K 5, 64 -> 64, three-term split) on the graph a user of the reference's HealpyGCNN
    gets (healpy_networks.py:110-118: symmetrised 8 nearest neighbours, ELL width 11) at the size bench.py --config knn8h times it
    (nside 512, batch 16): the patch oracle at the twelve base pixels' corners and borders, tile corners and random rows, in the
    first, a middle and the last map."""
    import bench
    from test_gpu_round2 import _patch_reference, _special_rows

    nside, K, Fin, Fout, N = bench.CONFIGS["knn8h"]
    dev = torch.device("cuda", 0)
    cols, vals, lmax = bench.build_laplacian_knn(nside, dev, bench.KNN["knn8h"])
    M = cols.shape[0]
    rng = np.random.default_rng(8)
    W = (rng.standard_normal((Fin * K, Fout)) * orc.default_kernel_stddev(Fin, K)).astype(np.float32)
    b = rng.standard_normal(Fout).astype(np.float32)
    plan = _native.LaplacianPlan(cols, vals, device=0)
    plan.prepare(K, Fin, Fout=Fout)
    x = torch.randn((N, M, Fin), device=dev, generator=torch.Generator(device=dev).manual_seed(8))
    y, _ = _native.cheb_forward(plan, x, _dev(W), _dev(b), K, act=_native.ACT_RELU, precision=_native.PREC_BF16X3, algo=_native.ALGO_FUSED)
    s = float(y.abs().max())
    n_struct, n_bfs = plan.tile_counts(K)
    n_strip = plan.strip_tiles(Fin, Fout, K, _native.PREC_BF16X3, N=N)
    centres = np.unique(np.concatenate([_special_rows(nside, M, rng), rng.integers(0, M, size=300)]))
    sel = [0, N // 2, N - 1]
    ref = _patch_reference(cols, vals, x[sel], W, K, centres, bias=b, activation="relu")
    got = y[sel][:, torch.as_tensor(centres).cuda()].cpu().numpy()
    err = float(np.abs(got - ref).max() / s)
    print(f"knn8h: M {M}, ELL width {cols.shape[1]}, tiles: {n_strip} on the quad strips, {n_struct - n_strip} structured, {n_bfs} class G "
          f"({100.0 * n_bfs / (M // 256):.1f} %); {centres.size} rows x {len(sel)} maps: err {err:.2e}")
    assert err < 1e-5
    assert n_struct + n_bfs == M // 256


def test_bench_gpus_2_as_given():
    """`python3 bench.py --gpus 2 ...` without a launcher (the form the driver uses): bench.py starts its own ranks as a child job
    and relays rank 0's line.  On a one-GPU box the two ranks share the GPU and stage halo rows through the host (--backend gloo)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--config", "c2", "--quick",
                        "--steps", "3", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "strong" and rec["value"] > 0
    assert "contiguous NEST ranges" in rec["config"]["sharding"]
