"""T4: pixel-range sharding.  Layout consistency on one process, and the halo exchange end to end
under gloo with world_size 2, 4 and 8 (CPU): the exchange moves real bytes between the two processes, the
per-shard compute is the float64 oracle injected through the test seam, and the stitched result
must equal the unsharded oracle."""

import os
import socket

import numpy as np
import pytest
import torch
from scipy import sparse

from deepsphere import healpix, sharding, utils
from oracle import cheb_oracle as orc


def _prepared_ell(nside=8, mode="knn", indices=None):
    L = healpix.healpix_laplacian(nside, indices=indices, mode=mode)
    Lt, _ = orc.prepare_L(L)
    cols, vals = utils.csr_to_ell(Lt)
    return Lt, cols, vals


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("K", [1, 2, 5])
def test_layout_is_consistent(world, K):
    Lt, cols, vals = _prepared_ell(8, "knn")
    M = cols.shape[0]
    lays = [sharding.ShardLayout(cols, vals, K, r, world) for r in range(world)]
    assert sum(l.n_own for l in lays) == M
    A = (Lt != 0).astype(np.int8).tocsr()
    for r, lay in enumerate(lays):
        a, b = lay.own
        # hop levels by brute force
        reach = np.zeros(M, bool)
        reach[a:b] = True
        for _ in range(K - 1):
            reach[np.unique(A[np.nonzero(reach)[0]].indices)] = True
        assert set(lay.local_ids.tolist()) == set(np.nonzero(reach)[0].tolist())
        assert lay.n_cols == reach.sum() and lay.levels[0] == b - a
        # the local ELL reproduces the global rows it holds
        v = np.random.default_rng(r).standard_normal(M)
        loc = (lay.local_vals * v[lay.local_ids][lay.local_cols]).sum(1)
        ref = (Lt.astype(np.float64) @ v)[lay.local_ids[: lay.n_rows]]
        assert K == 1 or np.allclose(loc, ref, atol=1e-6)  # K = 1 never multiplies by L~
        # what r receives from p is exactly what p sends to r, in the same order
        for p in range(world):
            if p == r:
                continue
            got = lay.local_ids[lay.recv_local[p]] if p in lay.recv_local else np.zeros(0, np.int64)
            sent = lays[p].send_local.get(r, np.zeros(0, np.int64)) + lays[p].own[0]
            assert np.array_equal(got, sent)


def _oracle_compute(layout, x_ext, kernel):
    """float64 oracle on the shard's extended graph; rows beyond K-2 hops have no matrix row, which
    cannot reach the owned rows within K-1 steps."""
    n = layout.n_cols
    rows = np.repeat(np.arange(layout.n_rows), layout.local_cols.shape[1])
    A = sparse.csr_matrix((layout.local_vals.reshape(-1).astype(np.float64), (rows, layout.local_cols.reshape(-1))),
                          shape=(n, n))
    y = orc.chebyshev_forward(A, x_ext.numpy().astype(np.float64), kernel.numpy().astype(np.float64), layout.K)
    return torch.from_numpy(y[:, : layout.n_own])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, K, partial, out):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        idx = None
        if partial:
            idx = healpix.extend_indices(healpix.cap_indices(8, fraction=0.4), 8, 2)
        Lt, cols, vals = _prepared_ell(8, "knn", idx)
        M = cols.shape[0]
        rng = np.random.default_rng(7)
        Fin, Fout, N = 3, 4, 2
        x = rng.standard_normal((N, M, Fin)).astype(np.float32)
        W = rng.standard_normal((Fin * K, Fout)).astype(np.float32)
        fetched = []
        if partial:
            sh = sharding.ShardedChebyshev(cols, vals, K, rank=rank, world=world, kernel=W, _compute=_oracle_compute)
        else:  # the ELL as a row producer: this rank must read its own rows and halo only, and no other rank's range
            def producer(ids):
                fetched.append(np.asarray(ids))
                return cols[ids], vals[ids]

            sh = sharding.ShardedChebyshev(producer, None, K, rank=rank, world=world, kernel=W, _compute=_oracle_compute, M=M)
            got = np.concatenate(fetched) if fetched else np.zeros(0, np.int64)
            assert got.size == np.unique(got).size <= sh.layout.n_cols  # every row once, nothing outside own + halo
            assert set(got.tolist()) <= set(sh.layout.local_ids.tolist())
        a, b = sh.layout.own
        if partial:  # second form of the input: written straight into the extended buffer (no copy of own rows)
            view = sh.own_rows_view(N, Fin)
            view.copy_(torch.from_numpy(x[:, a:b].copy()))
            y_local = sh(view)
        else:
            y_local = sh(torch.from_numpy(x[:, a:b].copy()))
        ref = orc.chebyshev_forward(Lt, x, W, K)[:, a:b]
        err = float(np.abs(y_local.numpy() - ref).max() / np.abs(ref).max())
        # the halo rows really came over the wire: compare the extended input with the global one
        halo_ok = bool(np.array_equal(sh._x_ext.numpy(), x[:, sh.layout.local_ids]))
        # the packed send / receive rows are allocated once per (N, F): two more steps, still one allocation
        for _ in range(2):
            y_again = sh(torch.from_numpy(x[:, a:b].copy()))
            halo_ok = halo_ok and bool(np.array_equal(y_again.numpy(), y_local.numpy()))
        if sh._send_idx or sh._recv_idx:
            halo_ok = halo_ok and sh.exchange_allocations == 1
        res = torch.tensor([err, 1.0 if halo_ok else 0.0], dtype=torch.float64)
        gathered = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(gathered, res)
        if rank == 0:
            out.put([g.tolist() for g in gathered])
    finally:
        dist.destroy_process_group()


# world 2: the basic exchange; world 4, K 8: BASELINE configs[3]'s split (3 base pixels per rank, 7-ring halo);
# world 8, partial sky: configs[4]'s (the sorted index list cut into 8 chunks); world 8 full sphere: quarter base pixels
@pytest.mark.parametrize("world,K,partial", [(2, 5, False), (2, 3, True), (4, 8, False), (8, 5, True), (8, 5, False)])
def test_halo_exchange_gloo(world, K, partial):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, K, partial, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = out.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert len(res) == world
    for err, halo_ok in res:
        assert halo_ok == 1.0
        assert err < 1e-12


def test_sharded_layer_refuses_cpu_compute():
    _, cols, vals = _prepared_ell(4, "knn")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sharding.ShardedChebyshev(cols, vals, 3, rank=0, world=1, kernel=np.zeros((9, 2), np.float32))


@pytest.mark.parametrize("world,K", [(4, 5), (8, 8), (3, 1)])
def test_layout_from_row_producer_and_gathered_requests(world, K):
    """A rank that sees only a row producer and the other ranks' requests builds the same layout as one that walks
    every range on the whole ELL."""
    _, cols, vals = _prepared_ell(8, "knn")
    M = cols.shape[0]
    ref = [sharding.ShardLayout(cols, vals, K, r, world) for r in range(world)]
    own = [sharding.ShardLayout(lambda ids: (cols[ids], vals[ids]), None, K, r, world, M=M, peer_requests=None)
           for r in range(world)]
    for r in range(world):
        assert own[r].send_local is None
        own[r].set_peer_requests({p: own[p].requests.get(r) for p in range(world) if p != r})
        for name in ("local_ids", "local_cols", "local_vals", "levels"):
            assert np.array_equal(getattr(own[r], name), getattr(ref[r], name)), name
        assert own[r].n_rows == ref[r].n_rows and own[r].n_cols == ref[r].n_cols
        assert set(own[r].send_local) == set(ref[r].send_local) and set(own[r].recv_local) == set(ref[r].recv_local)
        for p in ref[r].send_local:
            assert np.array_equal(own[r].send_local[p], ref[r].send_local[p])
        for p in ref[r].recv_local:
            assert np.array_equal(own[r].recv_local[p], ref[r].recv_local[p])
    with pytest.raises(ValueError, match="does not own"):
        own[0].set_peer_requests({1: np.array([M - 1])})


def _planes_local(layout, x_ext):
    """T_k x on the shard's extended graph (float64), (K, N, n_cols, F): exact on the rank's own rows."""
    n = layout.n_cols
    rows = np.repeat(np.arange(layout.n_rows), layout.local_cols.shape[1])
    A = sparse.csr_matrix((layout.local_vals.reshape(-1).astype(np.float64), (rows, layout.local_cols.reshape(-1))),
                          shape=(n, n))
    return orc.chebyshev_planes(A, x_ext.numpy().astype(np.float64), layout.K)


def _oracle_compute_f32(layout, x_ext, kernel):
    return _oracle_compute(layout, x_ext, kernel).float()


def _oracle_wgrad(layout, x_ext, dy):
    planes = _planes_local(layout, x_ext)[:, :, : layout.n_own]
    dW = np.einsum("knmf,nmo->fko", planes, dy.numpy().astype(np.float64))
    return torch.from_numpy(dW.reshape(-1, dy.shape[2])).float()


def _train_worker(rank, world, port, K, out):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        Lt, cols, vals = _prepared_ell(8, "knn")
        M = cols.shape[0]
        rng = np.random.default_rng(3)
        Fin, Fout, N = 3, 4, 2
        x = rng.standard_normal((N, M, Fin)).astype(np.float32)
        W = rng.standard_normal((Fin * K, Fout)).astype(np.float32)
        b = rng.standard_normal(Fout).astype(np.float32)
        g = rng.standard_normal((N, M, Fout)).astype(np.float32)
        kernel = torch.nn.Parameter(torch.from_numpy(W.copy()))
        bias = torch.nn.Parameter(torch.from_numpy(b.copy()))
        sh = sharding.ShardedChebyshev(cols, vals, K, rank=rank, world=world, kernel=kernel, bias=bias,
                                       _compute=_oracle_compute_f32, _compute_wgrad=_oracle_wgrad)
        a, e = sh.layout.own
        xl = torch.from_numpy(x[:, a:e].copy()).requires_grad_(True)
        y = sh(xl)
        (y * torch.from_numpy(g[:, a:e].copy())).sum().backward()
        dx_ref, dW_ref = orc.chebyshev_backward(Lt, x, W, K, g)
        errs = [float(np.abs(xl.grad.numpy() - dx_ref[:, a:e]).max() / np.abs(dx_ref).max()),
                float(np.abs(kernel.grad.numpy() - dW_ref).max() / np.abs(dW_ref).max()),
                float(np.abs(bias.grad.numpy() - g.sum((0, 1))).max() / np.abs(g.sum((0, 1))).max()),
                float(np.abs(y.detach().numpy() - (orc.chebyshev_forward(Lt, x, W, K) + b)[:, a:e]).max())]
        res = torch.tensor(errs, dtype=torch.float64)
        gathered = [torch.zeros(4, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(gathered, res)
        if rank == 0:
            out.put([t.tolist() for t in gathered])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,K", [(2, 5), (4, 3)])
def test_sharded_training_gloo(world, K):
    """Autograd through the sharded layer: dx from one more halo exchange (of dy), dkernel from the ranks' partial sums
    and ONE all-reduce; both against the unsharded float64 oracle gradients."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, K, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = out.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for e_dx, e_dw, e_db, e_y in res:
        assert e_dx < 1e-5 and e_dw < 1e-5 and e_db < 1e-5 and e_y < 1e-4


def _scope_worker(rank, world, port, out):
    """Two data-parallel ranks, each with its OWN one-rank layer (world = 1, group = None): the gradient must not be summed
    over the job's default group.  A layer on a one-rank sub-group still goes through the collective."""
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _, cols, vals = _prepared_ell(4, "knn")
        W = np.ones((2 * 3, 2), dtype=np.float32)
        sh = sharding.ShardedChebyshev(cols, vals, 3, rank=0, world=1, kernel=W, _compute=_oracle_compute)
        t = torch.full((4,), float(rank + 1))
        sh._all_reduce(t)
        own = bool(torch.equal(t, torch.full((4,), float(rank + 1))))
        groups = [dist.new_group([r]) for r in range(world)]  # (every rank creates every group, in the same order)
        sh1 = sharding.ShardedChebyshev(cols, vals, 3, rank=0, world=1, kernel=W, group=groups[rank], _compute=_oracle_compute)
        t1 = torch.full((4,), float(rank + 1))
        sh1._all_reduce(t1)
        own = own and bool(torch.equal(t1, torch.full((4,), float(rank + 1))))
        refused = False
        try:  # a two-rank layer on a one-rank group is a configuration error, at construction
            sharding.ShardedChebyshev(cols, vals, 3, rank=rank, world=2, kernel=W, group=groups[rank], _compute=_oracle_compute)
        except ValueError:
            refused = True
        res = torch.tensor([1.0 if own else 0.0, 1.0 if refused else 0.0], dtype=torch.float64)
        gathered = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(gathered, res)
        if rank == 0:
            out.put([g.tolist() for g in gathered])
    finally:
        dist.destroy_process_group()


def test_one_rank_layer_inside_a_larger_job_keeps_its_gradient_gloo():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_scope_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = out.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res == [[1.0, 1.0], [1.0, 1.0]]


def test_sharded_layer_precision_names():
    """"f16x3" is a name every layer takes (KeyError here); its dx runs the six-term bf16 split, its dW exact fp32."""
    _, cols, vals = _prepared_ell(4, "knn")
    W = np.ones((2 * 3, 2), dtype=np.float32)
    sh = sharding.ShardedChebyshev(cols, vals, 3, kernel=W, precision="f16x3", _compute=_oracle_compute)
    assert (sh.precision, sh.precision_dx, sh.precision_dw) == (3, 2, 0)
    sh = sharding.ShardedChebyshev(cols, vals, 3, kernel=W, precision="bf16x3", _compute=_oracle_compute)
    assert (sh.precision, sh.precision_dx, sh.precision_dw) == (1, 1, 1)
    with pytest.raises(ValueError):
        sharding.ShardedChebyshev(cols, vals, 3, kernel=W, precision="fp8", _compute=_oracle_compute)
