"""Pixel-range sharding of one Chebyshev layer over the GPUs of a node (one process per GPU).

The reference has no distributed code at all; this is the multi-GPU form of the
same forward (``gnn_layers.py:106-161``).  Rows (pixels) are split into ``world`` contiguous
ranges -- in NEST order that is whole HEALPix base pixels (or quarters of them at 8 GPUs).  A
rank needs T_0 = x on every row within K-1 hops of its range to evaluate all K planes on its
own rows, so the exchange step is: every rank sends the x rows the others need (packed by a HIP
gather kernel), receives its own halo rows (point-to-point over RCCL/xGMI; no all-reduce is
involved anywhere), and then runs the ordinary single-GPU forward on a local plan whose rows are
ordered by hop distance (``dsph_plan_set_levels``): step k is evaluated only on the rows still
needed.  One exchange of the (K-1)-ring halo per forward instead of K-1 one-ring exchanges: the
same bytes in one message round, and the fused single-launch kernel keeps working.

A rank walks the hop levels of its own range only and reads only the ELL rows of its own rows and halo (the ELL may be
given as a row producer); which of its rows the others need it learns from one ``all_gather_object`` at set-up.
Without a process group (single-process tests) the send lists are derived by walking the other ranges on the whole ELL.
"""

import numpy as np
import torch

from . import _native

__all__ = ["row_ranges", "ShardLayout", "ShardedChebyshev"]


def row_ranges(M, world):
    """Contiguous, balanced row ranges [(begin, end)] * world."""
    return [(M * r // world, M * (r + 1) // world) for r in range(world)]


class ShardLayout:
    """What one rank needs to know: its rows, its halo by hop distance, who sends what to whom.

    ``ell_cols`` / ``ell_vals``: the prepared Laplacian in padded ELL form -- either the whole ``[M, W]`` arrays or,
    with ``M`` given, a callable ``ell_cols(ids) -> (cols[len(ids), W], vals[len(ids), W])`` that produces rows on
    demand (then ``ell_vals`` is ignored and the rank never holds more than its own rows and halo).  The rank walks
    the K-1 hop levels of ITS OWN range only; which of its rows the other ranks need is either told to it
    (``set_peer_requests``: every rank's ``requests`` gathered once at set-up, see ``ShardedChebyshev``) or, in
    single-process use with the whole ELL at hand, derived by walking the other ranges too (``peer_requests_replicated``).

    All index arrays are numpy int64 of *global* row ids unless called ``*_local``.
    """

    def __init__(self, ell_cols, ell_vals, K, rank, world, M=None, peer_requests="replicated"):
        if callable(ell_cols):
            if M is None:
                raise ValueError("a row producer needs the number of rows M")
            fetch = ell_cols
            self._whole = None
        else:
            cols = np.asarray(ell_cols)
            vals = np.asarray(ell_vals)
            M = cols.shape[0]
            self._whole = (cols, vals)

            def fetch(ids):
                return cols[ids], vals[ids]
        M = int(M)
        self.M, self.K, self.rank, self.world = M, int(K), int(rank), int(world)
        self.ranges = row_ranges(M, world)
        depth = max(self.K - 1, 0)
        a, b = self.ranges[rank]
        self.own = (a, b)
        self.n_own = b - a
        # hop levels of this rank's range (after the first hop only the boundary is touched); the ELL rows fetched on
        # the way are kept: they are exactly the rows of the local plan
        my, row_ids, row_cols, row_vals = self._levels(fetch, M, a, b, depth)
        self.halo_ids = np.concatenate(my) if my else np.zeros(0, np.int64)
        # local order: own rows, then level 1, level 2, ... (each ascending)
        self.local_ids = np.concatenate([np.arange(a, b, dtype=np.int64), self.halo_ids])
        self.n_cols = int(self.local_ids.shape[0])
        counts = np.cumsum([self.n_own] + [len(l) for l in my])  # rows within j hops, j = 0..depth
        self.rows_within = counts
        # rows that carry an ELL row: everything within K-2 hops (the outermost ring is input only)
        self.n_rows = int(counts[depth - 1]) if depth >= 1 else self.n_own
        # plan levels: rows_at_level[j] for j = 0..K-2 (step k is evaluated on level K-1-k)
        self.levels = counts[:depth].astype(np.int64) if depth >= 1 else np.array([self.n_own], np.int64)
        # local ELL with remapped columns (a sparse global -> local map: nothing of size M is allocated per rank
        # beyond the BFS's bitmap)
        order = np.argsort(self.local_ids, kind="stable")
        sorted_ids = self.local_ids[order]

        def to_local(g):
            pos = np.searchsorted(sorted_ids, g)
            pos = np.minimum(pos, sorted_ids.size - 1)
            hit = sorted_ids[pos] == g
            return np.where(hit, order[pos], -1)

        self._to_local = to_local
        if depth >= 1:
            rc = np.concatenate(row_cols[:depth]) if self.n_rows else np.zeros((0, 1), np.int64)
            rv = np.concatenate(row_vals[:depth]) if self.n_rows else np.zeros((0, 1), np.float32)
            assert np.array_equal(np.concatenate(row_ids[:depth]), self.local_ids[: self.n_rows])
        else:  # K = 1: no product with L~ is ever taken, the local matrix is a placeholder
            rc, rv = fetch(np.arange(a, b, dtype=np.int64))
            rv = np.zeros_like(rv)
        lv = np.asarray(rv, dtype=np.float32)
        lc = to_local(np.asarray(rc, dtype=np.int64))
        dead = lv == 0
        lc = np.where(dead, np.arange(self.n_rows, dtype=np.int64)[:, None], lc)
        if (lc < 0).any():
            raise RuntimeError("a row within K-2 hops has a neighbour outside the K-1 hop halo")
        self.local_cols = lc.astype(np.int32)
        self.local_vals = lv
        # what I receive from p = halo(me) ∩ own(p): also the request p has to serve
        self.requests = {}    # p -> global ids (ascending) of p's rows this rank needs
        self.recv_local = {}  # p -> local positions (>= n_own) the rows from p land in
        for p in range(world):
            if p == rank:
                continue
            pa, pb = self.ranges[p]
            theirs = np.sort(self.halo_ids[(self.halo_ids >= pa) & (self.halo_ids < pb)])
            if theirs.size:
                self.requests[p] = theirs
                self.recv_local[p] = to_local(theirs)
        # what I send to p = halo(p) ∩ own(me) = p's request to me
        self.send_local = {} if world == 1 else None  # p -> local row indices (into my own rows) to pack, ascending global id
        if world == 1:
            pass
        elif isinstance(peer_requests, str) and peer_requests == "replicated":
            if self._whole is not None:
                self.set_peer_requests(self.peer_requests_replicated())
        elif peer_requests is not None:
            self.set_peer_requests(peer_requests)

    def set_peer_requests(self, peer_requests):
        """``peer_requests[p]`` = the global ids (of this rank's rows) that rank p asked for."""
        a, b = self.own
        self.send_local = {}
        for p, ids in peer_requests.items():
            if p == self.rank or ids is None:
                continue
            ids = np.sort(np.asarray(ids, dtype=np.int64))
            if ids.size == 0:
                continue
            if ids[0] < a or ids[-1] >= b:
                raise ValueError(f"rank {p} asked rank {self.rank} for rows it does not own")
            self.send_local[p] = ids - a

    def peer_requests_replicated(self):
        """Single-process stand-in for the set-up gather: walk every other rank's range on the whole ELL (world x the
        work; what ``ShardedChebyshev`` avoids when a process group exists)."""
        if self._whole is None:
            raise RuntimeError("the other ranks' requests need the whole ELL or a set-up gather")
        cols, vals = self._whole
        a, b = self.own
        depth = max(self.K - 1, 0)
        out = {}
        for p, (pa, pb) in enumerate(self.ranges):
            if p == self.rank:
                continue
            lv, _, _, _ = self._levels(lambda ids: (cols[ids], vals[ids]), self.M, pa, pb, depth)
            hp = np.concatenate(lv) if lv else np.zeros(0, np.int64)
            out[p] = np.sort(hp[(hp >= a) & (hp < b)])
        return out

    @staticmethod
    def _levels(fetch, M, a, b, depth):
        """Hop levels 1..depth of the row range [a, b) and the ELL rows read on the way: ids / cols / vals of level
        0 (the range itself), 1, ..., depth-1."""
        seen = np.zeros(M, dtype=bool)
        seen[a:b] = True
        out, row_ids, row_cols, row_vals = [], [], [], []
        frontier = np.arange(a, b, dtype=np.int64)
        for _ in range(depth):
            if frontier.size == 0:
                out.append(np.zeros(0, np.int64))
                row_ids.append(frontier)
                row_cols.append(np.zeros((0, row_cols[0].shape[1] if row_cols else 1), np.int64))
                row_vals.append(np.zeros((0, row_vals[0].shape[1] if row_vals else 1), np.float32))
                continue
            c, v = fetch(frontier)
            c = np.asarray(c)
            v = np.asarray(v)
            row_ids.append(frontier)
            row_cols.append(c.astype(np.int64))
            row_vals.append(v.astype(np.float32))
            nb = np.unique(c[v != 0])
            new = nb[~seen[nb]].astype(np.int64)
            seen[new] = True
            out.append(new)
            frontier = new
        return out, row_ids, row_cols, row_vals


def _pack(src, idx, out=None):
    """buf[n, i, :] = src[n, idx[i], :] (into ``out`` when given).  HIP gather kernel on the GPU; plain indexing on CPU
    tensors (reached only by the gloo tests: it moves bytes, it computes nothing)."""
    if src.is_cuda:
        return _native.rows_pack(src, idx, out=out)
    if out is None:
        return src[:, idx.long()].contiguous()
    return torch.index_select(src, 1, idx.long(), out=out)


def _unpack(dst, idx, buf):
    if dst.is_cuda:
        return _native.rows_unpack(dst, idx, buf)
    dst[:, idx.long()] = buf
    return dst


class _ShardedConvFunction(torch.autograd.Function):
    """The sharded forward as a differentiable op (no bias, no activation), so that ``ShardedChebyshev`` can train.

    For an upstream gradient dy on this rank's rows (L~ symmetric, as every graph Laplacian is):
    * dx on the rank's rows = the same forward applied to dy with the weights re-indexed ``[o*K + k, f]`` -- it needs dy
      on the (K-1)-ring halo, i.e. ONE more halo exchange with the very send / receive lists of the forward;
    * dkernel = the rank's partial sum ``sum_{n, m in own rows} (T_k x)[n,m,f] dy[n,m,o]`` (``dsph_cheb_backward_weights`` on the
      local plan, whose planes are exact on the own rows) followed by ONE all-reduce of the ``[Fin*K, Fout]`` array (80 KiB
      at K 5, 64 -> 64) -- the only collective of the whole path.  The weights are replicated, so every rank ends up with
      the same gradient.
    The reference trains these layers on one device (``examples/advanced_tutorial.ipynb``) and has no distributed code."""

    @staticmethod
    def forward(ctx, x_local, kernel, shard):
        y = shard._forward(x_local, kernel.detach(), None, _native.ACT_NONE)
        # the extended input is needed again for dkernel: keep this buffer, the next forward gets a fresh one
        ctx.x_ext = shard._x_ext
        shard._x_ext = None
        ctx.shard = shard
        ctx.save_for_backward(kernel)
        return y

    @staticmethod
    def backward(ctx, dy):
        shard = ctx.shard
        (kernel,) = ctx.saved_tensors
        K = shard.K
        x_ext = ctx.x_ext
        Fin = x_ext.shape[2]
        Fout = kernel.shape[1]
        dy = dy.contiguous()
        dx = dk = None
        if ctx.needs_input_grad[0]:
            kernel_t = kernel.detach().reshape(Fin, K, Fout).permute(2, 1, 0).reshape(Fout * K, Fin).contiguous()
            dx = shard._forward(dy, kernel_t, None, _native.ACT_NONE, buf="_dy_ext", precision=shard.precision_dx)
        if ctx.needs_input_grad[1]:
            dk = shard._wgrad_local(x_ext, dy)
            shard._all_reduce(dk)
        ctx.x_ext = None
        return dx, dk, None


class _ReplicatedGrad(torch.autograd.Function):
    """Identity on a replicated parameter (the bias) whose gradient is the sum of the ranks' partial gradients."""

    @staticmethod
    def forward(ctx, p, shard):
        ctx.shard = shard
        return p.view_as(p)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().clone()
        ctx.shard._all_reduce(g)
        return g, None


class ShardedChebyshev:
    """One rank's share of a Chebyshev layer whose map is split over ``world`` processes.

    ``ell_cols`` / ``ell_vals``: the prepared Laplacian in padded ELL form -- the whole arrays (every rank passes
    the same ones) or, with ``M``, a callable producing rows on demand (``ShardLayout``).  ``kernel``: the layer's [Fin*K, Fout] weights (same on every rank).
    Call with this rank's rows ``x_local`` of shape (N, own_rows, Fin); returns (N, own_rows, Fout).
    ``group``: a torch.distributed process group (default: the world group).
    ``_compute``: test seam -- a callable ``(layout, x_ext, kernel) -> y`` replacing the HIP forward so
    that the exchange logic can be exercised under gloo on CPU; never set by product code (``_compute_wgrad``:
    the same for the rank's partial weight gradient, ``(layout, x_ext, dy) -> dkernel``).
    ``plan_options``: ``_native.OPT_*`` -> value for the rank's plan (``dsph_plan_set_option``), e.g. ``{OPT_STRIPS: STRIPS_NEVER}``
    when the shards must reproduce the unsharded rows bit for bit whatever their size.
    Pass ``kernel`` (and ``bias``) as torch tensors that require grad and the call is differentiable: see
    ``_ShardedConvFunction`` (one more halo exchange for dx, one all-reduce of dkernel).
    """

    def __init__(self, ell_cols, ell_vals, K, Fout=None, rank=0, world=1, device=None, precision="fp32",
                 algo="auto", kernel=None, bias=None, act=_native.ACT_NONE, group=None, _compute=None, M=None,
                 _compute_wgrad=None, plan_options=None):
        import torch.distributed as dist

        gather = int(world) > 1 and dist.is_available() and dist.is_initialized()
        if gather and dist.get_world_size(group) != int(world):
            raise ValueError(f"ShardedChebyshev(world={int(world)}) on a process group of {dist.get_world_size(group)} ranks")
        self.layout = ShardLayout(ell_cols, ell_vals, K, rank, world, M=M, peer_requests=None if gather else "replicated")
        if gather:
            # set-up only: every rank publishes which rows it needs from whom; a rank learns its send lists from that
            # instead of walking the other ranks' ranges (world x the work, and the whole ELL on every rank)
            everyone = [None] * int(world)
            dist.all_gather_object(everyone, {int(p): v for p, v in self.layout.requests.items()}, group=group)
            self.layout.set_peer_requests({p: everyone[p].get(int(rank)) for p in range(int(world)) if p != int(rank)})
        if self.layout.send_local is None:
            raise RuntimeError("ShardedChebyshev with a row producer needs an initialised process group (or world = 1)")
        self.K, self.rank, self.world = int(K), int(rank), int(world)
        self.own_rows = self.layout.n_own
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        self.group = group
        self._compute = _compute
        precisions = {"fp32": _native.PREC_FP32, "bf16x3": _native.PREC_BF16X3, "bf16x6": _native.PREC_BF16X6,
                      "f16x3": _native.PREC_F16X3}
        if precision not in precisions:
            raise ValueError(f"precision must be one of {sorted(precisions)}")
        self.precision = precisions[precision]
        # the input gradient runs the forward kernels on dy, whose scale nobody vouches for: "f16x3" (x split into f16 pairs as it
        # is) takes the six-term bf16 split there, like gnn_layers.resolve_dx_precision; the weight gradient has kernels for exact
        # fp32 and the three-term bf16 split only (include/dsphere.h)
        self.precision_dx = _native.PREC_BF16X6 if precision == "f16x3" else self.precision
        self.precision_dw = _native.PREC_BF16X3 if precision == "bf16x3" else _native.PREC_FP32
        self.algo = {"auto": _native.ALGO_AUTO, "unfused": _native.ALGO_UNFUSED, "fused": _native.ALGO_FUSED}[algo]
        self.act = act
        # a torch tensor (e.g. a Parameter that requires grad) is kept as it is: calling the layer is then differentiable
        self.kernel = (kernel if isinstance(kernel, torch.Tensor) else
                       None if kernel is None else torch.as_tensor(np.asarray(kernel), dtype=torch.float32).to(self.device))
        self.bias = (bias if isinstance(bias, torch.Tensor) else
                     None if bias is None else torch.as_tensor(np.asarray(bias), dtype=torch.float32).reshape(-1).to(self.device))
        self.Fout = Fout if Fout is not None else (None if self.kernel is None else int(self.kernel.shape[1]))
        lay = self.layout
        self._send_idx = {p: torch.as_tensor(v.astype(np.int32)).to(self.device) for p, v in lay.send_local.items()}
        self._recv_idx = {p: torch.as_tensor(v.astype(np.int32)).to(self.device) for p, v in lay.recv_local.items()}
        self.plan = None
        self.fused = False
        if _compute is None:
            if self.device.type != "cuda":
                raise RuntimeError("ShardedChebyshev computes on a HIP device only; there is no CPU fallback")
            self.plan = _native.LaplacianPlan(lay.local_cols, lay.local_vals, n_cols=lay.n_cols,
                                              device=self.device.index, levels=lay.levels, options=plan_options)
        self._x_ext = None
        self._dy_ext = None
        self._xbufs = {}                # packed send / receive rows per extended buffer, see _exchange_buffers
        self.exchange_allocations = 0   # how often those were (re)allocated: 1 per buffer and (N, F) in steady state
        self._workspace = None
        self._workspace_w = None
        self._compute_wgrad = _compute_wgrad

    def own_rows_view(self, N, F):
        """The (N, own, F) window of the extended input buffer that holds this rank's rows.  A producer that
        writes its output straight into this view (and then passes the view to ``__call__``) saves the copy of
        the rank's rows into the buffer -- 2 x |x_local| of HBM traffic per forward."""
        lay = self.layout
        if self._x_ext is None or tuple(self._x_ext.shape) != (N, lay.n_cols, F):
            self._x_ext = torch.empty((N, lay.n_cols, F), dtype=torch.float32, device=self.device)
        return self._x_ext[:, : lay.n_own]

    def dry_run(self, N, Fin):
        """One forward of the local plan on the extended buffer as it is, WITHOUT the exchange: whatever fails locally
        (allocation, unsupported shape, kernel launch) fails here, before this rank posts or skips a point-to-point
        operation that its peers would wait on forever.  Callers agree on the outcome (an all-reduce of a flag) and
        only then run the first real step; ``bench.py`` does."""
        if self._compute is not None:
            return
        self.own_rows_view(N, Fin)  # creates the extended buffer if it does not exist yet
        y, self._workspace = _native.cheb_forward(self.plan, self._x_ext, self.kernel.detach(), None, self.K,
                                                  act=_native.ACT_NONE, precision=self.precision, algo=self.algo,
                                                  workspace=self._workspace)
        torch.cuda.synchronize(self.device)
        return tuple(y.shape)

    def exchange(self, x_local, buf="_x_ext"):
        """(N, own, F) -> (N, n_cols, F): own rows followed by the halo rows fetched from their owners."""
        x_ext, finish = self.exchange_start(x_local, buf)
        finish()
        return x_ext

    def exchange_start(self, x_local, buf="_x_ext"):
        """Starts the halo exchange and returns ``(x_ext, finish)``: the transfers are in flight until ``finish()``
        (wait + unpack into the halo rows of ``x_ext``) -- work that touches no halo row can be issued in between."""
        import torch.distributed as dist

        lay = self.layout
        N, own, F = x_local.shape
        if own != lay.n_own:
            raise ValueError(f"this rank owns {lay.n_own} rows, got {own}")
        buf_name = buf
        cur = getattr(self, buf)  # "_x_ext" for the input, "_dy_ext" for the upstream gradient of the backward pass
        in_place = (cur is not None and tuple(cur.shape) == (N, lay.n_cols, F)
                    and x_local.data_ptr() == cur.data_ptr() and x_local.stride() == cur.stride())
        if cur is None or tuple(cur.shape) != (N, lay.n_cols, F):
            cur = torch.empty((N, lay.n_cols, F), dtype=torch.float32, device=x_local.device)
            setattr(self, buf, cur)
        x_ext = cur
        if not in_place:
            x_ext[:, :own].copy_(x_local)
        if self.world == 1 or not (self._send_idx or self._recv_idx):
            return x_ext, (lambda: None)
        # RCCL moves device buffers directly (xGMI peer-to-peer).  Under a gloo group (no RCCL: several
        # ranks sharing one GPU, debugging) the packed rows are staged through host memory instead;
        # only the transport differs, the pack / unpack kernels and the forward are the same.
        via_host = x_local.is_cuda and dist.get_backend(self.group) == "gloo"
        xb = self._exchange_buffers(buf_name, N, F, x_local.device, via_host)
        ops = []
        for p, idx in self._send_idx.items():
            _pack(x_ext, idx, out=xb["send"][p])
            if via_host:
                xb["send_host"][p].copy_(xb["send"][p])
            ops.append(dist.P2POp(dist.isend, xb["send_host" if via_host else "send"][p], self._peer(p), group=self.group))
        for p in self._recv_idx:
            ops.append(dist.P2POp(dist.irecv, xb["recv_host" if via_host else "recv"][p], self._peer(p), group=self.group))
        reqs = dist.batch_isend_irecv(ops)

        def finish():
            for req in reqs:
                req.wait()
            for p, idx in self._recv_idx.items():
                if via_host:
                    xb["recv"][p].copy_(xb["recv_host"][p])
                _unpack(x_ext, idx, xb["recv"][p])

        return x_ext, finish

    def _exchange_buffers(self, name, N, F, device, via_host):
        """The packed send and receive rows of one extended buffer (``_x_ext`` | ``_dy_ext``), allocated once per
        (N, F) and kept: a step of the exchange allocates nothing.  Under a host-staged (gloo) group of HIP ranks every
        buffer has a pinned host twin."""
        key = (name, int(N), int(F), str(device), bool(via_host))
        xb = self._xbufs.get(name)
        if xb is not None and xb["key"] == key:
            return xb
        mk = lambda idx, dev, pin=False: torch.empty((N, idx.numel(), F), dtype=torch.float32, device=dev, pin_memory=pin)
        xb = {"key": key,
              "send": {p: mk(i, device) for p, i in self._send_idx.items()},
              "recv": {p: mk(i, device) for p, i in self._recv_idx.items()}}
        if via_host:
            xb["send_host"] = {p: mk(i, "cpu", True) for p, i in self._send_idx.items()}
            xb["recv_host"] = {p: mk(i, "cpu", True) for p, i in self._recv_idx.items()}
        self._xbufs[name] = xb
        self.exchange_allocations += 1
        return xb

    def _peer(self, p):
        import torch.distributed as dist

        return p if self.group is None else dist.get_global_rank(self.group, p)

    def __call__(self, x_local):
        train = torch.is_grad_enabled() and (x_local.requires_grad or (self.kernel is not None and self.kernel.requires_grad)
                                             or (self.bias is not None and self.bias.requires_grad))
        if not train:
            return self._forward(x_local, self.kernel, self.bias, self.act)
        # differentiable: the convolution through _ShardedConvFunction, bias and activation by the host framework
        y = _ShardedConvFunction.apply(x_local, self.kernel, self)
        if self.bias is not None:
            y = y + _ReplicatedGrad.apply(self.bias, self).reshape(1, 1, -1)
        if self.act != _native.ACT_NONE:
            y = {_native.ACT_RELU: torch.relu, _native.ACT_ELU: torch.nn.functional.elu, _native.ACT_SIGMOID: torch.sigmoid,
                 _native.ACT_TANH: torch.tanh}[self.act](y)
        return y

    def _forward(self, x_local, kernel, bias, act, buf="_x_ext", precision=None):
        if self._compute is not None:
            return self._compute(self.layout, self.exchange(x_local, buf), kernel)
        Fin = x_local.shape[2]
        self.fused = self.plan.fused_ok(Fin, int(kernel.shape[1]), self.K) and self.algo != _native.ALGO_UNFUSED
        kw = dict(act=act, precision=self.precision if precision is None else precision, algo=self.algo)
        if not self.fused or self.world == 1:
            x_ext = self.exchange(x_local, buf)
            y, self._workspace = _native.cheb_forward(self.plan, x_ext, kernel, bias, self.K,
                                                      workspace=self._workspace, **kw)
            return y
        # the tiles that read no halo row run while the halo rows are on the wire (RCCL works on its own stream),
        # the boundary tiles after they have landed: the exchange hides behind the interior
        x_ext, finish = self.exchange_start(x_local, buf)
        y, self._workspace = _native.cheb_forward(self.plan, x_ext, kernel, bias, self.K,
                                                  workspace=self._workspace, part=_native.PART_INTERIOR, **kw)
        finish()
        y, self._workspace = _native.cheb_forward(self.plan, x_ext, kernel, bias, self.K,
                                                  workspace=self._workspace, part=_native.PART_BOUNDARY, out=y, **kw)
        return y

    def _wgrad_local(self, x_ext, dy):
        """This rank's partial weight gradient: sum over its own rows of (T_k x) dy^T."""
        if self._compute_wgrad is not None:
            return self._compute_wgrad(self.layout, x_ext, dy)
        dk, self._workspace_w = _native.cheb_backward_weights(self.plan, x_ext, dy, self.K, algo=self.algo,
                                                              workspace=self._workspace_w, precision=self.precision_dw)
        return dk

    def _all_reduce(self, t):
        """Sum over the ranks, in place (RCCL; host-staged under a gloo group like the halo exchange)."""
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized()):
            if self.world != 1:
                raise RuntimeError("ShardedChebyshev: the gradient of a layer sharded over several ranks needs a process group")
            return t
        # The sum runs over THIS layer's ranks only: a one-rank layer (world = 1, group = None) inside a job whose default
        # group is larger -- data parallel over 8 ranks, say -- must not sum its gradient over those unrelated ranks (or hang
        # when they call unevenly).  A one-rank group still goes through the collective: the sum over one rank.
        if dist.get_world_size(self.group) != self.world:
            if self.world == 1:
                return t
            raise RuntimeError(f"ShardedChebyshev(world={self.world}) on a process group of {dist.get_world_size(self.group)} ranks")
        if t.is_cuda and dist.get_backend(self.group) == "gloo":
            h = t.cpu()
            dist.all_reduce(h, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, group=self.group)
        return t
