#!/usr/bin/env python3
"""Benchmark of the Chebyshev graph-convolution forward (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one forward of the layer over one batch of synthetic maps already resident in HBM
(headline workload: nside=1024 full sphere, K=5, Fin=Fout=64, batch 4, fp32).  Prints ONE JSON
line on rank 0: throughput in Mpix*channels/s (= N*M*Fout / t), the HBM-roofline figures of the
kernel(s) of one forward measured with HIP events, and a CPU baseline (the oracle's fp32 port
of the reference op sequence on a bounded sample, host cores counted).

With N > 1 the map is sharded over the ranks by contiguous NEST pixel ranges (base pixels and
their quarters) and the (K-1)-ring halo of x is exchanged over RCCL inside every timed step:
strong scaling on the fixed headline map.
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "deepsphere-cosmo-tf2_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

CONFIGS = {
    # name: (nside, K, Fin, Fout, batch)  -- BASELINE.json configs[0..2]
    "c1": (64, 5, 1, 16, 1),
    "c2": (256, 5, 16, 32, 8),
    "c3": (1024, 5, 64, 64, 4),
    # configs[3] run on ONE GPU (BASELINE quotes it sharded over 4): the 7-ring-halo variant of the fused kernel
    "c4": (2048, 8, 32, 32, 1),
    # configs[4] on ONE GPU (BASELINE quotes it on 8): partial sky, a spherical cap of 1/3 of the sphere padded to
    # nside-8 superpixels like utils.extend_indices, ragged tiles and border rows
    "c5": (1024, 5, 64, 64, 16),
    # configs[4]'s cap padded the way a network pads it (utils.extend_indices to the nside of its coarsest layer: nside-32 superpixels,
    # 32 x 32 pixels = 2 x 2 tiles each): no tile has its 4-ring region inside its superpixel -- until round 6 every tile of such a map
    # was class T; the quad strips' rectangles on the logical tile grid take them
    "c5s": (1024, 5, 64, 64, 16),
    # beyond BASELINE.json (side lines, never the headline): the order of the reference's tutorial layers
    # (examples/quick_start.ipynb:118-127, HealpyChebyshev(K=10, ...)) at configs[1]'s map and channel counts
    "k10": (256, 10, 16, 32, 8),
    # the first layer of a DeepSphere stack (tests/test_healpy_networks.py:96-107: one input channel) at nside 512
    "in1": (512, 5, 1, 16, 8),
    # the reference's own graphs (healpy_networks.py:38-41,108-118: k nearest neighbours, symmetrised; every shipped model uses
    # n_neighbors = 20) at configs[1]'s map and channel counts: ELL width 11 / 23
    "knn8": (256, 5, 16, 32, 8),
    "knn20": (256, 5, 16, 32, 8),
    # the headline SHAPE (K 5, 64 -> 64) on the reference's own kind of graph (healpy_networks.py:110-118: symmetrised 8 nearest
    # neighbours) at nside 512, batch 16: as many (map, pixel) units per forward as the headline config
    "knn8h": (512, 5, 64, 64, 16),
    # a layer of the reference's quick-start model (examples/quick_start.ipynb:118-127,142-147): K = 10, five channels,
    # 20 neighbours, batch 16 -- at nside 256
    "qs": (256, 10, 5, 5, 16),
    "qs1": (256, 10, 1, 5, 16),  # its first layer: one input channel
}
MASKED = {"c5": 8, "c5s": 32}  # config -> nside of the superpixels the mask is padded to
KNN = {"knn8": 8, "knn20": 20, "qs": 20, "qs1": 20, "knn8h": 8}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)


def algorithmic_bytes(N, M, Fin, Fout, K, W_ell, bias=False):
    """x once + y once + L~ (int32 col + fp32 val, padded ELL) once + weights."""
    return 4 * N * M * Fin + 4 * N * M * Fout + 8 * W_ell * M + 4 * K * Fin * Fout + (4 * Fout if bias else 0)


def build_laplacian(nside, device):
    """Full-sphere 8-neighbour normalised Laplacian, rescaled like Chebyshev.__init__: ELL arrays."""
    from deepsphere import _native, healpix, utils

    cols_t, vals_t = healpix.grid_laplacian_ell_torch(nside, device=device)
    cols = cols_t.cpu().numpy()
    plan_L = _native.LaplacianPlan(cols, vals_t.to(torch.float32).cpu().numpy(), device=device.index)
    lam = utils.lanczos_lmax(plan_L, iters=64)
    plan_L.close()
    lmax = 1.02 * lam
    vals = utils.rescale_ell(cols_t, vals_t, lmax=lmax, scale=0.75).cpu().numpy()
    return cols, vals, lmax


def build_laplacian_masked(nside, device, fraction=1.0 / 3.0, nside_super=8):
    """Partial-sky version: NEST indices of a cap around (1, 0, 0), padded to nside-`nside_super` superpixels, sorted."""
    from deepsphere import _native, healpix, utils

    idx = healpix.extend_indices(healpix.cap_indices(nside, fraction=fraction), nside, nside_super)
    L = healpix.healpix_laplacian(nside, indices=idx, mode="grid")
    cols, vals64 = utils.csr_to_ell(L)
    plan_L = _native.LaplacianPlan(cols, vals64.astype(np.float32), device=device.index)
    lam = utils.lanczos_lmax(plan_L, iters=64)
    plan_L.close()
    lmax = 1.02 * lam
    cols, vals = utils.csr_to_ell(utils.rescale_L(L, lmax=lmax, scale=0.75))  # diagonal wherever CSR order puts it
    return cols, vals.astype(np.float32), lmax


def build_laplacian_knn(nside, device, k):
    """The reference's kind of graph: symmetrised k-nearest-neighbour Gaussian-kernel graph on the pixel centres, normalised
    Laplacian (deepsphere/healpix.py; parity with the pygsp fork is unverifiable here), rescaled like Chebyshev.__init__."""
    from deepsphere import _native, healpix, utils

    L = healpix.healpix_laplacian(nside, n_neighbors=k, mode="knn")
    cols, vals64 = utils.csr_to_ell(L)
    plan_L = _native.LaplacianPlan(cols, vals64.astype(np.float32), device=device.index)
    lam = utils.lanczos_lmax(plan_L, iters=64)
    plan_L.close()
    lmax = 1.02 * lam
    cols, vals = utils.csr_to_ell(utils.rescale_L(L, lmax=lmax, scale=0.75))
    return cols, vals.astype(np.float32), lmax


def cpu_baseline(K, Fin, Fout, device, budget_s, nside_headline=1024):
    """The reference op sequence (gnn_layers.py:131-150) in fp32 on the host cores, every op multi-threaded
    (oracle/cheb_port.c: OpenMP relayouts and sparse products, BLAS GEMM).  Sample: ONE map of the headline nside
    when the host has the memory for the reference's materialisations (K + 2 copies of x plus the stacked matrix,
    ~40 GB at nside 1024), else nside 512; the op is linear in the batch, so Mpix*channels/s does not depend on it."""
    import psutil

    from oracle import cheb_cpu_baseline as cb

    cores = os.cpu_count() or 1
    avail_gb = psutil.virtual_memory().available / 2 ** 30
    nside_s = nside_headline if avail_gb > 96 else (512 if avail_gb > 28 else 256)
    N_s = 1
    cols, vals, _ = build_laplacian(nside_s, device)
    M = cols.shape[0]
    rng = np.random.default_rng(11)
    x = rng.standard_normal((N_s, M, Fin), dtype=np.float32)
    w = (np.random.default_rng(13).standard_normal((Fin * K, Fout)) / np.sqrt(Fin * (K + 0.5) / 2)).astype(np.float32)
    res = cb.time_forward_threaded(cols, vals, x, w, K, threads=cores, budget_s=budget_s * 0.7)
    v_thr = N_s * M * Fout / res["seconds"] / 1e6
    out = {
        "value": round(v_thr, 3),
        "unit": "Mpix*channels/s",
        "cores": int(res["threads"]),
        "kind": "port",
        "sample": f"nside={nside_s} full-sphere, batch={N_s} (1/{(nside_headline // nside_s) ** 2 * 4} of the headline pixel-batch), "
                  f"K={K}, Fin={Fin}, Fout={Fout}; OpenMP port of the reference op sequence + BLAS GEMM on {res['threads']} threads, "
                  f"median of {res['reps']} forwards, {res['seconds'] * 1e3:.0f} ms each; host RAM available {avail_gb:.0f} GiB",
    }
    # the single-thread scipy figure of round 1, on a small sample, for continuity
    try:
        from scipy import sparse

        c2, v2, _ = build_laplacian(256, device)
        M2, Wd = c2.shape
        Lt = sparse.csr_matrix((v2.reshape(-1), c2.reshape(-1), np.arange(0, Wd * M2 + 1, Wd)), shape=(M2, M2))
        x2 = rng.standard_normal((1, M2, Fin), dtype=np.float32)
        r1 = cb.time_forward_scipy_1thread(Lt, x2, w, K, budget_s=budget_s * 0.15)
        out["other_port"] = {"value": round(M2 * Fout / r1["seconds"] / 1e6, 3), "cores": 1,
                             "what": f"scipy CSR @ dense + numpy GEMM on one thread, nside=256, batch=1, {r1['seconds'] * 1e3:.0f} ms"}
    except Exception as exc:  # noqa: BLE001
        out["other_port"] = {"error": repr(exc)}
    return out


def measured_error(cols, vals, x, y, w_np, K, nside, seed=3, n_random=24):
    """max |y - y_ref| / max |y| of the TIMED output against the float64 oracle (used as the checker, after the timed
    region): the oracle is run on the (K-1)-hop patches around rows in every map -- base-pixel corners and borders,
    tile corners, first / last rows, random rows."""
    from scipy import sparse

    from oracle import cheb_oracle as orc

    M = cols.shape[0]
    ns2 = nside * nside
    rng = np.random.default_rng(seed)
    centres = [0, 1, M - 1, ns2 - 1, ns2, min(M - 1, 5 * ns2 + 77)] + [int(v) for v in rng.integers(0, M, size=n_random)]
    centres += [min(M - 1, f * ns2 + d) for f in range(0, 12, 3) for d in (0, ns2 - 1, ns2 // 2, 255, 256)]
    centres = np.unique(np.array(centres, dtype=np.int64))
    region = centres.copy()
    for _ in range(K - 1):
        region = np.unique(np.concatenate([region, cols[region][vals[region] != 0]]))
    lut = -np.ones(M, dtype=np.int64)
    lut[region] = np.arange(region.size)
    rc, rv = cols[region], vals[region]
    keep = (rv != 0) & (lut[rc] >= 0)
    rows = np.repeat(np.arange(region.size), cols.shape[1]).reshape(rc.shape)
    sub = sparse.csr_matrix((rv[keep].astype(np.float64), (rows[keep], lut[rc][keep])), shape=(region.size, region.size))
    xs = x[:, torch.as_tensor(region, device=x.device)].cpu().numpy().astype(np.float64)
    ref = orc.chebyshev_forward(sub, xs, w_np.astype(np.float64), K)[:, lut[centres]]
    got = y[:, torch.as_tensor(centres, device=y.device)].cpu().numpy().astype(np.float64)
    s_max = float(y.abs().max())
    return float(np.abs(got - ref).max() / s_max), int(centres.size * x.shape[0])


STRIP_FORM = "quad"  # --strip-form: which strip kernel the 64 -> 64 shape runs (names the kernel in the roofline block)


def fused_kernel_name(plan, K, Fin, Fout, prec_code, N=1, split="auto"):
    """Which kernels one fused forward launches: the strip kernel on the rectangles of plain structured tiles it takes for this
    shape (dsph_plan_strip_tiles), the structured-tile kernel on the other structured tiles, the BFS-tile kernel on the rest
    (dsph_plan_tile_counts)."""
    if K > 5 and (split == "always" or (split == "auto" and plan.uses_chain(Fin, Fout, K))):
        terms, k = [], K
        while k > 5:
            k -= 4
            terms.append(5)
        terms.append(k)
        cz = (Fin + Fout + 3) // 4 * 4
        return (f"{len(terms)} passes of {' / '.join(str(t) for t in reversed(terms))} terms (T_(4+j) = 2 T_4 T_j - T_|4-j|) through [x | u] of "
                f"{cz} channels; last pass: " + fused_kernel_name(plan, 5, cz, Fout, prec_code, N))
    if K > 10:
        return f"cheb_step_kernel x{K - 1} + cheb_contract_f32_kernel"
    n_struct, n_bfs = plan.tile_counts(K)
    n_strip = plan.strip_tiles(Fin, Fout, K, prec_code, N=N)
    parts = []
    if n_strip and K == 8:  # (round 6: the three-role quad strips of the K = 8, 32 -> 32 shape; the other tiles stay on the tile kernel)
        return f"cheb_qstrip8_kernel ({n_strip} tiles) + cheb_fused_kernel ({n_struct + n_bfs - n_strip} tiles)"
    if n_strip:
        parts.append(f"{'cheb_istrip1_kernel' if Fin <= 2 else ('cheb_istrip_kernel' if Fin <= 16 else ('cheb_qstrip5_kernel' if STRIP_FORM == 'quad' else 'cheb_strip5_kernel'))} ({n_strip} tiles)")
    if n_struct - n_strip:
        parts.append(f"cheb_struct_kernel ({n_struct - n_strip} tiles)")
    if n_bfs:
        parts.append(f"cheb_fused_kernel ({n_bfs} tiles)")
    return " + ".join(parts)


def training_step_leg(cols, vals, K, lmax, Fout, device, args, plan_options, w_np, x, steps):
    """Forward + backward of the layer (no bias, no activation) on the bench's input: ms by HIP events, split at the backward."""
    from deepsphere import gnn_layers

    layer = gnn_layers.Chebyshev.from_prepared_ell(cols, vals, K, lmax=lmax, Fout=Fout, device=device, precision=args.precision,
                                                   algo=args.algo, initializer=lambda t: t.copy_(torch.from_numpy(w_np)),
                                                   plan_options=plan_options)
    xg = x.detach().requires_grad_(True)
    dy = torch.randn((x.shape[0], x.shape[1], Fout), device=device)

    def one():
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        y = layer(xg)
        e[1].record()
        y.backward(dy)
        e[2].record()
        xg.grad = None
        layer.kernel.grad = None
        return e

    for _ in range(2):
        one()
    evs = [one() for _ in range(steps)]
    torch.cuda.synchronize()
    fwd = float(np.mean([a.elapsed_time(b) for a, b, _ in evs]))
    bwd = float(np.mean([b.elapsed_time(c) for _, b, c in evs]))
    return {"forward_ms": round(fwd, 3), "backward_ms": round(bwd, 3), "ms_per_step": round(fwd + bwd, 3), "steps": steps,
            "note": "autograd of the layer: dx = the forward kernels on dy, dkernel = dsph_cheb_backward_weights at the layer's "
                    "weight-gradient precision (gnn_layers.resolve_wgrad_precision: three-term bf16 from 4,096 pixels up, exact fp32 below)"}


def timed_ms(run, steps, warm=3):
    """Mean HIP-event time of `steps` forwards on the current stream, after `warm` untimed ones."""
    for _ in range(warm):
        run()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in ev:
        a.record()
        run()
        b.record()
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in ev]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--precision", default=None, choices=["auto", "fp32", "bf16x3", "bf16x6", "f16x3"],
                    help="contraction arithmetic; default: the layers' own default (gnn_layers.DEFAULT_PRECISION = auto: bf16x3 with 16 "
                         "or more input channels, else bf16x6).  bf16x3 = 3-term split-bf16 MFMA with fp32 accumulate (error "
                         "measured in the line); bf16x6 = fp32-equivalent 6-term split; fp32 = exact fp32 MFMA")
    ap.add_argument("--algo", default="auto", choices=["auto", "unfused", "fused"])
    ap.add_argument("--split", default="auto", choices=["auto", "always", "never"],
                    help="K > 5: the product-identity chain of K <= 5 passes (csrc/cheb_split.hip) -- plan option DSPH_OPT_SPLIT")
    ap.add_argument("--strips", default="auto", choices=["auto", "always", "never"], help="plan option DSPH_OPT_STRIPS")
    ap.add_argument("--strip-form", default="quad", choices=["quad", "pairs"],
                    help="plan option DSPH_OPT_STRIP_FORM: quad strips (round 5) or the strip pairs of round 3")
    ap.add_argument("--tstep", default="on", choices=["on", "off"], help="plan option DSPH_OPT_TSTEP (wide graphs: tiled step)")
    ap.add_argument("--struct", default="on", choices=["on", "off"],
                    help="plan option DSPH_OPT_STRUCT (off: every tile on the breadth-first tile kernel)")
    ap.add_argument("--fork", default="on", choices=["on", "off"],
                    help="plan option DSPH_OPT_FORK (the BFS-tile launch beside the structured ones, on the plan's side stream)")
    ap.add_argument("--quick", action="store_true", help="the headline leg only: no side legs in the other arithmetics, no CPU baseline")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for --gpus > 1: nccl = RCCL over xGMI (one rank per GPU); gloo = halo rows "
                         "staged through the host, ranks dealt round-robin over the visible GPUs (debugging on a 1-GPU box)")
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of CPU baseline; 0 disables it")
    args = ap.parse_args()
    global STRIP_FORM
    STRIP_FORM = args.strip_form

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python3 bench.py --gpus N` as given: start the one-rank-per-GPU job as a CHILD process -- before anything in this
        # process has touched the GPU (no torch.cuda call above), never by exec -- relay its output (rank 0's JSON line) and
        # leave with its exit code
        import socket
        import subprocess

        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1")).returncode)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} under WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus} "
                         "(or plain `python3 bench.py --gpus N`, which does that itself)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    if args.backend == "gloo":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend="gloo")

    from deepsphere import _native, gnn_layers

    plan_options = {_native.OPT_SPLIT: {"auto": 0, "always": 1, "never": 2}[args.split],
                    _native.OPT_STRIPS: {"auto": 0, "always": 1, "never": 2}[args.strips],
                    _native.OPT_STRIP_FORM: _native.STRIP_FORM_QUAD if args.strip_form == "quad" else _native.STRIP_FORM_PAIRS,
                    _native.OPT_TSTEP: 1 if args.tstep == "on" else 0,
                    _native.OPT_FORK: 1 if args.fork == "on" else 0}
    if args.struct == "off":
        plan_options[_native.OPT_STRUCT] = 0
    nside, K, Fin, Fout, N = CONFIGS[args.config]
    # what is timed is what a user of the layer gets: the layer's default arithmetic unless --precision says otherwise
    layer_default = args.precision is None
    if layer_default:
        args.precision = gnn_layers.DEFAULT_PRECISION
    resolved = gnn_layers.resolve_precision(args.precision, Fin, K)
    prec_code = gnn_layers._PRECISIONS[resolved]
    t0 = time.time()
    if args.config in KNN:
        cols, vals, lmax = build_laplacian_knn(nside, device, KNN[args.config])
    else:
        cols, vals, lmax = build_laplacian_masked(nside, device, nside_super=MASKED[args.config]) if args.config in MASKED else build_laplacian(nside, device)
    M, W_ell = cols.shape
    w_np = (np.random.default_rng(13).standard_normal((Fin * K, Fout)) / np.sqrt(Fin * (K + 0.5) / 2)).astype(np.float32)

    if world == 1:
        layer = gnn_layers.Chebyshev.from_prepared_ell(
            cols, vals, K, lmax=lmax, Fout=Fout, device=device, precision=args.precision, algo=args.algo,
            initializer=lambda t: t.copy_(torch.from_numpy(w_np)), plan_options=plan_options,
        )
        gen = torch.Generator(device=device).manual_seed(11)
        x = torch.randn((N, M, Fin), device=device, generator=gen)
        # (precision "f16x3" only: the bound on |x| its power-of-two input scale is taken from -- a property of the synthetic
        # input the caller knows, handed over once; without it the layer would reduce max|x| inside every timed forward)
        layer.x_absmax = float(x.abs().amax())
        def run():
            with torch.no_grad():  # inference forward: bias/activation fused in the kernel epilogue
                return layer(x)
        fused = layer._get_plan().fused_ok(Fin, Fout, K) and args.algo != "unfused"
        if K > 9:  # (the layer's "auto" arithmetic follows the plan's route: one pass -- K = 10 on the grid -- or the chain of passes)
            resolved = gnn_layers.resolve_precision(args.precision, Fin, K, layer._get_plan().uses_chain(Fin, Fout, K))
            prec_code = gnn_layers._PRECISIONS[resolved]
        step_name = "cheb_tstep_kernel" if (12 < W_ell <= 32 and args.tstep == "on") else "cheb_step_kernel"
        kernel_name = fused_kernel_name(layer._get_plan(), K, Fin, Fout, prec_code, N, args.split) if fused else f"{step_name} x{K - 1} + cheb_contract_f32_kernel"
    else:
        from deepsphere import sharding

        # the ELL as a row producer: the rank reads its own rows and halo only, and learns its send lists from the
        # set-up gather of every rank's requests (sharding.ShardLayout)
        shard = sharding.ShardedChebyshev(lambda ids: (cols[ids], vals[ids]), None, K, Fout=Fout, rank=rank, world=world,
                                          device=device, precision=resolved, algo=args.algo, kernel=w_np, M=M,
                                          plan_options=plan_options)
        gen = torch.Generator(device=device).manual_seed(11 + rank)
        # this rank's rows live in the extended buffer the kernel reads (own rows, then halo rows): a producer
        # layer would write them there; no per-step copy
        x = shard.own_rows_view(N, Fin)
        x.normal_(generator=gen)
        run = lambda: shard(x)  # noqa: E731
        fused = shard.plan.fused_ok(Fin, Fout, K) and args.algo != "unfused"
        kernel_name = (fused_kernel_name(shard.plan, K, Fin, Fout, prec_code, N, "never") if fused else f"cheb_step_kernel x{K - 1} + cheb_contract_f32_kernel") + \
            " + rows_pack_kernel + " + \
            ("RCCL send/recv" if args.backend == "nccl" else "gloo send/recv (host-staged)") + " of the (K-1)-ring halo"
    setup_s = time.time() - t0

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # A sharded forward that cannot run (an exception on any rank during its first step, agreed on by all ranks) ends the
    # job with a non-zero exit code on every rank: the line below is the sharded metric or nothing (there is no replica leg).
    if world > 1:
        def agree(err):
            flag = torch.tensor([1.0 if err else 0.0], dtype=torch.float32, device=device if args.backend == "nccl" else "cpu")
            try:
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                return flag.item() > 0, err
            except Exception as exc:  # noqa: BLE001
                return True, err or repr(exc)

        # 1. everything local first (no point-to-point operation yet): a rank that cannot run its own plan says so
        #    before its peers post receives they would wait on forever
        err = None
        try:
            if os.environ.get("DSPH_BENCH_FAIL_SHARD"):  # exercises the exit below
                raise RuntimeError("forced by DSPH_BENCH_FAIL_SHARD")
            shard.dry_run(N, Fin)
        except Exception as exc:  # noqa: BLE001
            err = repr(exc)
        failed, err = agree(err)
        # 2. the first real step, with the exchange
        if not failed:
            try:
                run()
                torch.cuda.synchronize()
            except Exception as exc:  # noqa: BLE001
                err = repr(exc)
            failed, err = agree(err)
        if failed:
            if rank == 0:
                print(f"bench.py: the sharded forward failed ({err or 'on another rank'})", file=sys.stderr, flush=True)
            dist.destroy_process_group()
            raise SystemExit(3)
    # Warm up exactly as the timed loop runs: the previous output stays referenced while the next forward allocates its
    # own, so BOTH output blocks are in the caching allocator before the clock starts (a first-ever hipMalloc of a second
    # 12.9 GB block inside the timed region costs one forward 350 ms on a box whose memory has not been touched yet).
    y_timed = run()
    y_timed = run()  # (allocator priming, before the W warm-up steps; not counted anywhere)
    for _ in range(args.warmup):
        y_timed = run()
    # HIP events on the stream the kernels run on.  A pair per forward when a forward is long enough not to notice them; for
    # the small maps (BASELINE configs[0]: 25 us per forward, where two event records per step cost as much again and the
    # loop would time the events; under 2 ms per forward in general) ONE pair around the whole timed region: average = region / steps.
    torch.cuda.synchronize()
    t_probe = time.perf_counter()
    y_timed = run()
    torch.cuda.synchronize()
    per_step_events = (time.perf_counter() - t_probe) > 2e-3
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps if per_step_events else 1)]
    barrier()
    t_start = time.perf_counter()
    if per_step_events:
        for a, b in ev:
            a.record()
            y_timed = run()
            b.record()
    else:
        ev[0][0].record()
        for _ in range(args.steps):
            y_timed = run()
        ev[0][1].record()
    barrier()
    elapsed = time.perf_counter() - t_start
    per_fwd_ms = [a.elapsed_time(b) / (1 if per_step_events else args.steps) for a, b in ev]

    el = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = N * M * Fout / (elapsed / args.steps) / 1e6

    if rank == 0:
        # per rank: its share of the map (the halo rows it also reads are not algorithmic bytes)
        b_alg = algorithmic_bytes(N, M // world if world > 1 else M, Fin, Fout, K, W_ell)
        dev_ms = float(np.mean(per_fwd_ms))
        achieved = b_alg / (dev_ms * 1e-3) / 1e9
        traffic = None
        err_note = ""
        if world == 1 and not os.environ.get("DSPH_BENCH_NO_CHECK"):
            # the oracle as the checker of what was just timed (after the timed region): max error over the patches
            try:
                e_rel, n_pts = measured_error(cols, vals, x, y_timed, w_np, K, nside)
                err_note = f"measured max err {e_rel:.2e} of max|y| on {n_pts} (map, row) outputs of the timed run vs the float64 oracle"
            except Exception as exc:  # noqa: BLE001
                err_note = f"error measurement failed: {exc!r}"
        def measured_error_of(yq):
            if os.environ.get("DSPH_BENCH_NO_CHECK"):
                return None
            try:
                return float("%.3g" % measured_error(cols, vals, x, yq, w_np, K, nside)[0])
            except Exception as exc:  # noqa: BLE001
                return repr(exc)

        issue = {}
        key = f"{args.config}:{resolved}:{'fused' if fused else 'unfused'}:{world}"
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                rec = json.load(open(tpath))
                traffic = rec.get(key, {}).get("hbm_bytes_per_forward")
                issue = rec.get(key, {})
            except Exception:
                traffic = None
        out = {
            "metric": f"Mpix*channels/s Chebyshev-conv fwd, nside={nside} K={K} F={Fin if Fin == Fout else f'{Fin}->{Fout}'}; HBM roofline %",
            "value": round(value, 2),
            "unit": "Mpix*channels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": ({"fp32": "f32 (recurrence and contraction exact f32: v_mfma_f32_32x32x2_f32",
                       "bf16x6": "f32 (recurrence f32; contraction fp32-equivalent: 6-term exact split on the bf16 MFMA with f32 accumulate",
                       "f16x3": "f32 (recurrence f32; contraction fp32-equivalent: 3-term split on the f16 MFMA (11 + 11 mantissa bits) with f32 accumulate",
                       "bf16x3": "f32 (recurrence f32; contraction 3-pass split-bf16 MFMA with f32 accumulate"}[resolved]
                      + (f"; precision '{resolved}' = the layer default for {Fin} input channels" if layer_default else f"; --precision {args.precision}")
                      + (f"; {err_note})" if err_note else ")")),
            "data": "synthetic",
            "config": {
                "workload": f"nside={nside} {f'partial sky (cap of 1/3 of the sphere, nside-{MASKED[args.config]} superpixels)' if args.config in MASKED else 'full-sphere'}, K={K}, Fin={Fin}, Fout={Fout}, batch={N} ({args.config})",
                "pixels": M,
                "ell_width": W_ell,
                "graph": (f"symmetrised {KNN[args.config]}-nearest-neighbour Gaussian-kernel graph on the HEALPix pixel centres" if args.config in KNN
                          else "8-neighbour HEALPix grid stencil") + ", normalised Laplacian, lmax by 64-step Lanczos",
                "algo": "fused" if fused else "unfused",
                "sharding": "none" if world == 1 else f"{world} contiguous NEST ranges, (K-1)-ring halo of x per step, exchange hidden behind the interior tiles",
                "setup_s": round(setup_s, 1),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": kernel_name,
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "frac_of_measured_copy_6290": round(achieved / 6290.0, 4),
                "median_forward_ms_hip_events": round(float(np.median(per_fwd_ms)), 4),
                "traffic": traffic,
                "traffic_source": (f"profiles/hbm_traffic.json entry '{key}' (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE passes, 2 x FETCH + WRITE: "
                                   f"{issue.get('source', 'source not recorded')}); build: {issue.get('kernel', 'not recorded')}; not re-measured in this run") if traffic else None,
                "traffic_kernels": issue.get("kernels"),
                "traffic_matches_kernel": (issue.get("kernels") == kernel_name) if (traffic and issue.get("kernels")) else None,
                "timing_note": "steady state: the packed weight images of the previous forward are kept (DSPH_FWD_KEEP_WEIGHTS), the weight-preparation launches (3 kernels, ~15 us) are outside the timed forwards",
                "algorithmic_bytes": b_alg,
                "avg_forward_ms_hip_events": round(dev_ms, 4),
                "hip_events": "one pair per forward" if per_step_events else "one pair around the timed region (forwards under 2 ms)",
                "min_max_forward_ms_hip_events": [round(float(np.min(per_fwd_ms)), 4), round(float(np.max(per_fwd_ms)), 4)],
            },
        }
        if world == 1 and not args.quick:
            # the same forward in the other contraction arithmetics, for the record (HIP events over `steps` forwards each).
            # Each leg's roofline is the largest of its lower bounds: algorithmic bytes at 8 TB/s, and the dense flops at the
            # peak of the pipe the arithmetic really runs on -- fp32 MFMA 157.3 TF/s; bf16 MFMA 2500 TF/s, counted three / six
            # times for the three- / six-term split.
            f_d = 2.0 * N * M * K * Fin * Fout  # flops of the dense contraction
            t_hbm = b_alg / (HBM_PEAK_GBS * 1e9) * 1e3
            legs = {"fp32": ("fp32_exact", "contraction on v_mfma_f32_32x32x2_f32 (bitwise an fp32 fma chain)", "mfma_f32", f_d / 157.3e12 * 1e3),
                    "bf16x6": ("fp32_split", "fp32-equivalent six-term split on v_mfma_f32_32x32x16_bf16 (operands split exactly into "
                                             "8 + 8 + 8 mantissa bits, products down to 2^-16 kept)", "mfma_bf16 x6", 6 * f_d / 2500e12 * 1e3),
                    "f16x3": ("fp32_f16x3", "fp32-equivalent three-term split on v_mfma_f32_16x16x32_f16 (operands split into 11 + 11 mantissa "
                                            "bits, x times the power of two that puts max|x| in [2^13, 2^14) -- x_absmax handed to the layer "
                                            "once; the quad strips' arithmetic, the other tiles run the six-term bf16 split)", "mfma_f16 x3",
                              3 * f_d / 2500e12 * 1e3),
                    "bf16x3": ("bf16_split3", "three-term split on v_mfma_f32_16x16x32_bf16 (quad strips) / 32x32x16 (tiles)", "mfma_bf16 x3", 3 * f_d / 2500e12 * 1e3)}
            out["roofline"]["bounds_ms"] = {"hbm": round(t_hbm, 3), legs[resolved][2]: round(legs[resolved][3], 3)}
            if issue.get("valu_insts_per_forward"):
                # What the kernels of this forward ISSUE (SQ_INSTS_VALU / SQ_INSTS_MFMA of the same build, profiles/hbm_traffic.json),
                # priced with the machine model of MI355X_MICROARCH.md and tools/ubench/issue_share (profiles/r5_ubench_issue_share.txt):
                # a SIMD-32 runs a wave64 vector instruction in 2 cycles (4 with a DPP operand: measured, whatever the waves per SIMD);
                # ONE wave issues one every 4 -- so a kernel with w waves per SIMD cannot go below 4 / w per instruction;
                # a matrix instruction holds the SIMD's vector issue for 8 cycles and the matrix pipe for 16 (16x16x32) or 32
                # (32x32x16).  1,024 SIMDs at the nominal 2.4 GHz (the chip grants 1.85-1.95 GHz under this load: DESIGN 4.0).
                simd_hz = 1024 * 2.4e9
                nv, nm = issue["valu_insts_per_forward"], issue.get("mfma_insts_per_forward", 0)
                dpp = issue.get("dpp_share_of_valu", 0.0)
                pipe_each = issue.get("mfma_pipe_cycles_each", 32)
                waves = issue.get("waves_per_simd", 2)
                b = out["roofline"]["bounds_ms"]
                b["valu_pipe"] = round(nv * (2 + 2 * dpp) / simd_hz * 1e3, 3)
                b["valu_single_wave_issue"] = round(nv * 4 / waves / simd_hz * 1e3, 3)
                b["mfma_pipe"] = round(nm * pipe_each / simd_hz * 1e3, 3)
                b["mfma_issue_hold"] = round(nm * 8 / simd_hz * 1e3, 3)
                b["issue_sum"] = round((nv * max(2 + 2 * dpp, 4 / waves) + nm * 8) / simd_hz * 1e3, 3)
                b["issue_note"] = ("issue_sum = vector instructions at max(pipe, single-wave issue / waves per SIMD) + 8 cycles per MFMA: what "
                                   "tools/ubench/issue_share measures for waves that carry both (the matrix pipe time is hidden, its issue is not); "
                                   f"waves per SIMD {waves}, DPP share {dpp}")
                b["issue_source"] = "SQ_INSTS_VALU / SQ_INSTS_MFMA per forward, replayed from profiles/hbm_traffic.json"
            for prec_name, (key, note, pipe, t_pipe) in legs.items():
                if resolved == prec_name:
                    continue
                layer.precision = prec_name
                msq = timed_ms(run, max(args.steps, 20))
                yq = run()  # for the error measurement below
                bound = max(t_hbm, t_pipe)
                out[key] = {"ms_per_step": round(msq, 4), "value": round(N * M * Fout / msq / 1e3, 2), "note": note,
                            "measured_error": measured_error_of(yq),
                            "roofline": {"bound": "hbm" if t_hbm >= t_pipe else pipe, "bound_ms": round(bound, 3),
                                         "frac": round(bound / msq, 4), "hbm_ms": round(t_hbm, 3), "pipe_ms": round(t_pipe, 3),
                                         "flops": f_d, "hbm_frac": round(b_alg / (msq * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
                del yq
            layer.precision = args.precision
        if world == 1 and not args.quick:
            # one run with bias + ReLU fused into the kernel epilogue
            layer.precision = args.precision
            layer.use_bias = True
            layer.bias = torch.nn.Parameter(torch.randn(1, 1, Fout, device=device))
            layer.activation, layer._act_code = gnn_layers._resolve_activation("relu")
            out["bias_relu"] = {"ms_per_step": round(timed_ms(run, max(args.steps, 20)), 4)}
        if world == 1 and not args.quick and K <= 5:  # (K > 5: the weight gradient goes through planes in memory: tens of GB at c4)
            # SURVEY 8 (f1), for the record: the layer's training step through autograd -- dx on the forward kernels, dkernel by
            # dsph_cheb_backward_weights (K = 5, 64 -> 64 j: the quad-strip weight-gradient kernel) -- HIP events, not the metric
            try:
                out["training_step"] = training_step_leg(cols, vals, K, lmax, Fout, device, args, plan_options, w_np, x, max(3, min(args.steps, 10)))
            except Exception as exc:  # noqa: BLE001  (a side leg must not take the headline down: out of memory on a small box)
                out["training_step"] = {"error": repr(exc)[:200]}
        if world == 1 and args.cpu_budget > 0 and not args.quick:
            out["cpu_baseline"] = cpu_baseline(K, Fin, Fout, device, args.cpu_budget)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
