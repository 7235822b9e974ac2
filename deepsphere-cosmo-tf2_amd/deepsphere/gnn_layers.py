"""Graph layers: the Chebyshev graph convolution on MI355X.

Host-side mirror of the reference's ``deepsphere.gnn_layers.Chebyshev`` (a Keras ``Model``,
``/root/reference/src/deepsphere/gnn_layers.py:12-161``) as a ``torch.nn.Module`` with the same
constructor arguments, lazy weight creation, attribute names, weight layout and error
behaviour.  The arithmetic is not here: ``forward`` hands device pointers to the C ABI of
``include/dsphere.h`` (hand-written HIP for gfx950).  Without that library and a GPU the
forward raises; there is no CPU path in this package.
"""

import numpy as np
import torch
from scipy import sparse

from . import _native
from . import utils

# Activations addressable by name, like ``getattr(tf.keras.activations, name)``
# (gnn_layers.py:55-60).  Those with a code are fused into the kernel epilogue; the others run
# as one extra elementwise pass after it.
_ACT_BY_NAME = {
    "linear": (None, _native.ACT_NONE),
    "relu": (torch.relu, _native.ACT_RELU),
    "elu": (torch.nn.functional.elu, _native.ACT_ELU),
    "sigmoid": (torch.sigmoid, _native.ACT_SIGMOID),
    "tanh": (torch.tanh, _native.ACT_TANH),
    "softplus": (torch.nn.functional.softplus, None),
    "softsign": (torch.nn.functional.softsign, None),
    "selu": (torch.selu, None),
    "gelu": (torch.nn.functional.gelu, None),
    "swish": (torch.nn.functional.silu, None),
    "silu": (torch.nn.functional.silu, None),
    "leaky_relu": (lambda t: torch.nn.functional.leaky_relu(t, 0.2), None),
    "exponential": (torch.exp, None),
    "hard_sigmoid": (lambda t: torch.clamp(t / 6.0 + 0.5, 0.0, 1.0), None),
    "mish": (torch.nn.functional.mish, None),
    "softmax": (lambda t: torch.softmax(t, dim=-1), None),
}

_PRECISIONS = {"fp32": _native.PREC_FP32, "bf16x3": _native.PREC_BF16X3, "bf16x6": _native.PREC_BF16X6, "f16x3": _native.PREC_F16X3,
               "auto": None}
_ALGOS = {"auto": _native.ALGO_AUTO, "unfused": _native.ALGO_UNFUSED, "fused": _native.ALGO_FUSED}


def _resolve_activation(activation):
    """-> (callable or None, fused code or None).  Unknown names raise ValueError exactly where
    the reference does (gnn_layers.py:60)."""
    if activation is None:
        return None, _native.ACT_NONE
    if callable(activation):
        for fn, code in ((torch.relu, _native.ACT_RELU), (torch.nn.functional.relu, _native.ACT_RELU),
                         (torch.nn.functional.elu, _native.ACT_ELU), (torch.sigmoid, _native.ACT_SIGMOID),
                         (torch.tanh, _native.ACT_TANH)):
            if activation is fn:
                return activation, code
        return activation, None
    if isinstance(activation, str) and activation in _ACT_BY_NAME:
        return _ACT_BY_NAME[activation]
    raise ValueError(f"Could not find activation <{activation}> in tf.keras.activations...")


# The contraction arithmetic a layer uses when the caller does not say: "auto" =
#   * "bf16x3" for layers with 16 or more input channels: both operands of the dense product split hi + lo into bf16, the
#     three products hi.hi + hi.lo + lo.hi on the bf16 matrix pipe, fp32 accumulation.  A product is off by at most
#     3 * 2^-18 = 1.15e-5 of its size (so |error| <= 1.15e-5 * sum |x||w| in the worst case -- not a bound on max|y|);
#     measured 2-6e-6 of max|y| against the float64 oracle on every test shape with 16 or more input channels, and held to
#     2e-5 there (TOL_BF16X3 of the tests; SURVEY 8c allows the split-bf16 contraction 1e-4).  It is the arithmetic of the
#     strip kernel (csrc/cheb_strip_kernel.h) and what bench.py times: product and benchmark are the same code path.
#     (For scale: the reference's own matmul on the GPUs it targets is TF32 by default, 2^-11 per product.)
#   * "bf16x6" for layers with fewer input channels (the first layer of a network: Fin = 1): the fp32-EQUIVALENT six-term
#     split (both operands split exactly into 8 + 8 + 8 mantissa bits, the six products down to 2^-16 kept): 2-7e-7 of
#     max|y|.  With five products per output nothing averages out -- the three-term split measured 1.02e-5 on BASELINE
#     configs[0] -- and such layers cost next to nothing anyway.
# "fp32" (the bitwise fp32-fma-chain MFMA), "bf16x6" and "bf16x3" can be asked for by name; so can "f16x3" (round 5): the
# fp32-equivalent three-term split on f16 pairs (11 + 11 mantissa bits) where the quad-strip kernel runs (K = 5, 64 -> 64 per
# column block, HEALPix rectangles) and "bf16x6" everywhere else; x enters the f16 split times a power of two chosen from
# max|x| (the layer's ``x_absmax`` or a reduction over x: Chebyshev._set_f16_scale, DSPH_OPT_F16_XEXP), so the arithmetic is
# fp32-equivalent at every input scale.  Its backward: dx on "bf16x6" (dy has no known scale), dW exact fp32.
DEFAULT_PRECISION = "auto"


def resolve_precision(precision, Fin, K=None, chain=None):
    """The arithmetic a contraction over ``Fin`` input channels runs for ``precision`` ("auto" | "fp32" | "bf16x3" |
    "bf16x6").  "auto" is resolved PER CONTRACTION: the forward with the layer's Fin, the input gradient -- the forward of
    the transposed layer on dy -- with the layer's Fout (a 64 -> 1 layer runs its dx through the six-term split).  A layer
    with more than nine terms runs as a chain of passes (csrc/cheb_split.hip), each of which rounds its input to bf16
    hi + lo again under the three-term split (1.0e-5 measured at K = 10, three passes): "auto" gives those the
    fp32-equivalent six-term split.  ``chain``: whether the plan runs this shape as that chain (``LaplacianPlan.uses_chain``;
    None = not known: every K > 9 is taken for one) -- K = 10 on the 8-neighbour grid is ONE pass over 9-ring regions since
    round 6 and keeps the three-term split (3 - 5e-6 measured)."""
    if precision == "auto":
        chained = (K is not None and K > 9) if chain is None else bool(chain)
        return "bf16x3" if (Fin >= 16 and not chained) else "bf16x6"
    return precision


def resolve_dx_precision(precision, Fout, K=None, chain=None):
    """The arithmetic of the input gradient -- the forward of the transposed layer on dy, a contraction over ``Fout`` channels.
    As ``resolve_precision`` except for "f16x3": the quad strips split their x operand (here dy) into f16 pairs as it is, and
    upstream gradients are routinely far below the f16 range (a mean loss over 12.6 M pixels gives dy ~ 1e-8, the smallest f16
    subnormal is 6e-8) -- dx would come out zero or a few bits wide on the strips' pixels.  dy has no caller-known scale, so
    "f16x3" runs its dx on the six-term bf16 split (same accuracy, the exponent range of fp32)."""
    if precision == "f16x3":
        return "bf16x6"
    return resolve_precision(precision, Fout, K, chain)


# the weight gradient takes the three-term split from this many pixels (N * M) on; the one statement of the rule -- DESIGN.md 4.1,
# INTEGRATION.md 3, bench.py and tests/test_gpu_round5.py quote this constant
WGRAD_SPLIT_MIN_PIXELS = 4096


def resolve_wgrad_precision(precision, n_terms):
    """The arithmetic of the weight gradient: its contraction runs over the ``n_terms`` = N * M pixels of the batch, so the
    three-term split's per-product error (<= 1.15e-5) averages out over thousands of terms -- "auto" takes it from
    ``WGRAD_SPLIT_MIN_PIXELS`` = 4,096 pixels on (measured 4-7e-6 of max|dW| at nside 16 and above, held to 2e-5) and exact fp32
    below; "bf16x6" and "f16x3" have no weight-gradient kernel and run exact fp32 (include/dsphere.h)."""
    if precision == "auto":
        return "bf16x3" if n_terms >= WGRAD_SPLIT_MIN_PIXELS else "fp32"
    return "fp32" if precision in ("bf16x6", "f16x3") else precision


class _ChebConvFunction(torch.autograd.Function):
    """y = sum_k T_k(L~) x W_k as a differentiable op (no bias, no activation).

    The reference defines no custom gradient -- TensorFlow differentiates the op sequence of
    ``gnn_layers.py:131-150``.  Here the backward is assembled from the same HIP kernels:

    * dx = sum_k T_k(L~)^T (dy W_k^T) is the *forward* of the layer on dy with the plan of L~^T (the
      same plan when L~ is symmetric, as every graph Laplacian is) and the weights re-indexed as
      kernel_T[o*K + k, f] = kernel[f*K + k, o];
    * dkernel[f*K + k, o] = sum_{n,m} (T_k x)[n,m,f] dy[n,m,o] by ``dsph_cheb_backward_weights``: the fused
      tile kernel in weight-gradient mode (the planes never leave the LDS; each tile's T_k is contracted
      over its pixels against dy by exact-fp32 MFMAs), or, where that kernel does not apply, the planes
      rebuilt by ``dsph_cheb_planes`` and reduced against dy by ``dsph_cheb_wgrad`` (a split-over-pixels MFMA
      kernel with a fixed-order second stage: a library GEMM has no split-K for a 64 x 64 result
      reduced over 5e7 rows and took 150 ms here).
    """

    @staticmethod
    def forward(ctx, x, kernel, layer):
        plan = layer._get_plan()
        layer._set_f16_scale(plan, x)
        y, layer._workspace = _native.cheb_forward(
            plan, x, kernel.detach(), None, layer.K, act=_native.ACT_NONE,
            precision=layer._prec_code(), algo=_ALGOS[layer.algo], workspace=layer._workspace,
            basis=layer._basis)
        layer._wkey = None  # (training: the weights change between steps; the inference path re-packs once afterwards)
        ctx.layer = layer
        ctx.save_for_backward(x, kernel)
        return y

    @staticmethod
    def backward(ctx, dy):
        layer = ctx.layer
        x, kernel = ctx.saved_tensors
        K = layer.K
        N, M, Fin = x.shape
        Fout = kernel.shape[1]
        dy = dy.contiguous()
        dx = dk = None
        if ctx.needs_input_grad[0]:
            plan_t = layer._get_plan(transposed=True)
            kernel_t = kernel.detach().reshape(Fin, K, Fout).permute(2, 1, 0).reshape(Fout * K, Fin).contiguous()
            dx, layer._workspace_t = _native.cheb_forward(
                plan_t, dy, kernel_t, None, K, act=_native.ACT_NONE,
                precision=_PRECISIONS[resolve_dx_precision(layer.precision, Fout, K, plan_t.uses_chain(Fout, Fin, K) if K > 9 else None)],
                algo=_ALGOS[layer.algo], workspace=layer._workspace_t, basis=layer._basis)
        if ctx.needs_input_grad[1]:
            plan = layer._get_plan()
            dk, layer._workspace_w = _native.cheb_backward_weights(
                plan, x, dy, K, basis=layer._basis, algo=_ALGOS[layer.algo], workspace=getattr(layer, "_workspace_w", None),
                precision=_PRECISIONS[resolve_wgrad_precision(layer.precision, N * M)])
        return dx, dk, None


class Chebyshev(torch.nn.Module):
    """A graph convolutional layer using the Chebyshev approximation.

    y[n, m, o] = act( BN( sum_f sum_k (T_k(L~) x[n, :, f])[m] * kernel[f*K + k, o] ) + bias[o] )
    with T_0 = I, T_1 = L~, T_k = 2 L~ T_{k-1} - T_{k-2} and L~ = 1.5/lmax * L - I,
    lmax = 1.02 * lambda_max(L).
    """

    _basis = _native.BASIS_CHEBYSHEV
    _scale = 0.75  # rescale_L(L, lmax, scale=0.75), gnn_layers.py:67

    def _default_stddev(self, Fin):
        return 1.0 / np.sqrt(Fin * (self.K + 0.5) / 2.0)  # gnn_layers.py:92

    def __init__(
        self,
        L,
        K,
        Fout=None,
        initializer=None,
        activation=None,
        use_bias=False,
        use_bn=False,
        n_matmul_splits=1,
        **kwargs,
    ):
        """
        :param L: graph Laplacian (M x M): scipy sparse matrix or dense array
        :param K: number of polynomial terms T_0 .. T_{K-1}
        :param Fout: output channels, defaults to the number of input channels
        :param initializer: ``None`` (truncated normal, stddev 1/sqrt(Fin*(K+0.5)/2)), or a callable
            applied in place to the new ``[K*Fin, Fout]`` kernel tensor (``torch.nn.init`` style), or
            a callable ``shape -> array``
        :param activation: ``None``, a callable, or a Keras activation name ("linear", "relu", "elu", ...)
        :param use_bias: add a learnable bias of shape [1, 1, Fout]
        :param use_bn: batch normalisation (no scale/shift, momentum 0.9, eps 1e-5) before the bias
        :param n_matmul_splits: accepted for compatibility; the reference needs it only to stay under
            TensorFlow-GPU's sparse-matmul size limit, the HIP kernel has none
        :param kwargs: the reference forwards these to ``add_weight`` (regularizer, ...); stored in
            ``self.kwargs``.  Three keys are consumed here: ``device`` (torch device of the layer,
            default: current CUDA device), ``precision`` ("auto", the default: "bf16x3" -- the three-term bf16 split, 2-6e-6 of
            max|y| from the float64 oracle -- for 16 or more input channels, else "bf16x6" | "bf16x6": fp32-equivalent six-term
            split, 2-7e-7 | "fp32": exact-fp32 MFMA, bitwise an fp32 fma chain; the recurrence is fp32 in all of them; see
            DEFAULT_PRECISION; "f16x3": the fp32-equivalent three-term split on f16 pairs where the quad strips run -- takes
            ``x_absmax``, a bound on |x| known to the caller, for its power-of-two input scale; without one the layer reduces
            max|x| on the device and synchronises once per forward), ``algo`` ("auto" | "unfused" | "fused") and ``plan_options`` (a dict of
            ``_native.OPT_*`` -> value handed to ``dsph_plan_set_option``, e.g. ``{OPT_STRIPS: STRIPS_NEVER}`` for results
            that do not depend on the batch size).  ``graph=True`` (inference only): the prepared forward is captured into
            a HIP graph on first use and replayed while the input buffer, the weights and the shapes stay the same -- one
            graph launch instead of three to six kernel launches, the BFS-tile launch beside the structured ones; the
            returned tensor is then the SAME buffer on every call (copy it if it must outlive the next forward).
        """
        super().__init__()
        self.L = L
        self.K = int(K)
        if self.K < 1:
            raise ValueError("K must be at least 1")
        self.Fout = Fout
        self.use_bias = use_bias
        self.use_bn = use_bn
        self.bn = None  # created in build() once Fout is known
        self.initializer = initializer
        self.activation, self._act_code = _resolve_activation(activation)
        self.n_matmul_splits = n_matmul_splits
        device = kwargs.pop("device", None)
        precision = kwargs.pop("precision", DEFAULT_PRECISION)
        algo = kwargs.pop("algo", "auto")
        self._plan_options = dict(kwargs.pop("plan_options", None) or {})
        self._use_graph = bool(kwargs.pop("graph", False))
        self.x_absmax = kwargs.pop("x_absmax", None)
        if precision not in _PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_PRECISIONS)}")
        if algo not in _ALGOS:
            raise ValueError(f"algo must be one of {sorted(_ALGOS)}")
        self.precision = precision
        self.algo = algo
        self.kwargs = kwargs
        self._device = torch.device(device) if device is not None else None

        # Rescaled Laplacian (host, float64 -> float32), then padded ELL for the kernels.
        Lt, self.lmax = utils.prepare_L(L, scale=self._scale)
        self._M = Lt.shape[0]
        self._ell_cols, self._ell_vals = utils.csr_to_ell(Lt)
        self._nnz = int(Lt.nnz)
        self._plan = None
        self._plan_t = None
        self._workspace = None
        self._workspace_t = None
        self.kernel = None
        self.bias = None
        self._built = False

    # -- Keras-style lazy build ---------------------------------------------------------------
    def build(self, input_shape):
        """Create the weights for inputs of shape (batch, M, Fin)."""
        Fin = int(input_shape[-1])
        Fout = Fin if self.Fout is None else int(self.Fout)
        dev = self._resolve_device(None)
        kernel = torch.empty((self.K * Fin, Fout), dtype=torch.float32)
        if self.initializer is None:
            stddev = self._default_stddev(Fin)
            torch.nn.init.trunc_normal_(kernel, mean=0.0, std=stddev, a=-2.0 * stddev, b=2.0 * stddev)
        else:
            res = self.initializer(kernel)
            if res is not None and res is not kernel:
                kernel = torch.as_tensor(np.asarray(res), dtype=torch.float32).reshape(self.K * Fin, Fout).clone()
        self.kernel = torch.nn.Parameter(kernel.to(dev))
        if self.use_bias:
            # the reference creates the bias without an initializer (gnn_layers.py:104), i.e. the
            # framework default for add_weight: glorot-uniform on shape [1, 1, Fout]
            lim = float(np.sqrt(6.0 / (1.0 + Fout)))
            self.bias = torch.nn.Parameter((torch.rand((1, 1, Fout)) * 2.0 - 1.0).mul_(lim).to(dev))
        if self.use_bn:
            # Keras momentum 0.9 == torch momentum 0.1; center=False, scale=False
            self.bn = torch.nn.BatchNorm1d(Fout, eps=1e-5, momentum=0.1, affine=False).to(dev)
        self._Fin = Fin
        self._built = True

    def _resolve_device(self, x):
        if self._device is None:
            if x is not None and isinstance(x, torch.Tensor) and x.is_cuda:
                self._device = x.device
            elif torch.cuda.is_available():
                self._device = torch.device("cuda", torch.cuda.current_device())
            else:
                self._device = torch.device("cpu")  # weights can be created; forward will refuse
        return self._device

    def _get_plan(self, transposed=False):
        if self._plan is None:
            _native.require_gpu()
            dev = self._resolve_device(None)
            if dev.type != "cuda":
                raise RuntimeError("the Chebyshev forward needs a HIP device; there is no CPU fallback")
            index = dev.index if dev.index is not None else torch.cuda.current_device()
            self._device = torch.device("cuda", index)
            self._plan = _native.LaplacianPlan(self._ell_cols, self._ell_vals, device=index,
                                               options=getattr(self, "_plan_options", None))
        if not transposed:
            return self._plan
        if getattr(self, "_plan_t", None) is None:
            # the backward applies T_k(L~)^T; a symmetric L~ (every graph Laplacian) reuses the plan
            M, W = self._ell_cols.shape
            rows = np.repeat(np.arange(M, dtype=np.int64), W)
            A = sparse.csr_matrix((self._ell_vals.reshape(-1), (rows, self._ell_cols.reshape(-1).astype(np.int64))),
                                  shape=(M, M))
            A.sum_duplicates()
            if abs(A - A.T).max() == 0:
                self._plan_t = self._plan
            else:
                At = A.T.tocsr()
                At.sort_indices()
                tc, tv = utils.csr_to_ell(At)
                self._plan_t = _native.LaplacianPlan(tc, tv, device=self._device.index,
                                                     options=getattr(self, "_plan_options", None))
        return self._plan_t

    def _prec_code(self):
        """C-ABI code of the contraction arithmetic of this (built) layer: ``precision="auto"`` resolved with its Fin."""
        return _PRECISIONS[resolve_precision(self.precision, self._Fin, self.K, self._chained())]

    def _chained(self, transposed=False):
        """Does the plan run this layer's forward (``transposed``: its input gradient, Fout -> Fin on the transposed plan) as the
        chain of <= 5-term passes?  None below K = 10 (no plan is asked: resolve_precision's rule needs nothing there)."""
        if self.K <= 9 or getattr(self, "_plan", None) is None:
            return None
        if transposed:
            return self._get_plan(transposed=True).uses_chain(self.Fout, self._Fin, self.K)
        return self._plan.uses_chain(self._Fin, self.Fout, self.K)

    # -- forward --------------------------------------------------------------------------------
    def forward(self, input_tensor, training=False):
        """
        :param input_tensor: (batch, M, Fin) tensor or array; cast to float32 like Keras does
        :param training: batch-norm mode.  Default False like the reference's ``call(input_tensor, training=False)``
            (gnn_layers.py:106): a direct ``layer(x)`` normalises with the moving statistics and leaves them
            untouched, whatever ``module.training`` says; pass ``training=True`` (``HealpyGCNN`` forwards its own
            argument) to use and update batch statistics.  ``None`` means "follow ``self.training``" (torch style).
        :return: (batch, M, Fout) float32 tensor on the layer's device
        """
        if not isinstance(input_tensor, torch.Tensor):
            input_tensor = torch.as_tensor(np.asarray(input_tensor))
        if input_tensor.dim() != 3:
            raise ValueError("input must have shape (batch, M, Fin)")
        N, M, Fin = (int(s) for s in input_tensor.shape)
        if M != self._M:
            raise ValueError(f"input has {M} nodes, the Laplacian has {self._M}")
        self._resolve_device(input_tensor)
        if not self._built:
            self.build(input_tensor.shape)
        if Fin != self._Fin:
            raise ValueError(f"layer was built for Fin = {self._Fin}, got {Fin}")
        plan = self._get_plan()
        wants_grad = torch.is_grad_enabled() and (
            self.kernel.requires_grad or input_tensor.requires_grad or (self.use_bias and self.bias.requires_grad))
        prepared = getattr(self, "_prepared", None)
        if prepared is None or prepared[:2] != (self.K, Fin) or (wants_grad and not prepared[2]):
            # tile tables now, not inside the first kernel launch (allocation + synchronisation: dsph_plan_prepare_layer); with
            # autograd on also what the backward needs (DSPH_PREPARE_BACKWARD: the weight-gradient tables and the one-second host
            # pass that decides whether L~ is symmetric) -- otherwise the first training step would stall inside
            # dsph_cheb_backward_weights, mid-capture under a HIP graph
            plan.prepare(self.K, Fin, Fout=self.Fout, backward=wants_grad)
            self._prepared = (self.K, Fin, bool(wants_grad) or bool(prepared and prepared[:2] == (self.K, Fin) and prepared[2]))
        if wants_grad:
            # differentiable path: the linear part through the autograd function above, the epilogue
            # (BN -> bias -> activation, gnn_layers.py:152-159) as ordinary torch ops
            x = input_tensor.to(device=self._device, dtype=torch.float32).contiguous()
            y = _ChebConvFunction.apply(x, self.kernel, self)
            if self.use_bn:
                was_training = self.bn.training
                self.bn.train(self.training if training is None else bool(training))
                y = self.bn(y.transpose(1, 2)).transpose(1, 2)
                self.bn.train(was_training)
            if self.use_bias:
                y = y + self.bias
            if self.activation is not None:
                y = self.activation(y)
            return y
        x = input_tensor.detach().to(device=self._device, dtype=torch.float32).contiguous()
        bias = self.bias.detach().reshape(-1).contiguous() if self.use_bias else None
        kernel = self.kernel.detach()
        kver = self.kernel._version

        # Batch norm with the moving statistics (the reference's default call, training=False) is a per-channel scale and
        # shift between the contraction and the bias (gnn_layers.py:152-159: BN -> bias -> activation; center=False,
        # scale=False): folded into the weights and the bias, the whole epilogue runs inside the kernel -- no transposed
        # copies, no elementwise pass.  With batch statistics (training=True) the host framework normalises, as before.
        bn_training = self.use_bn and (self.training if training is None else bool(training))
        if self.use_bn and not bn_training:
            kernel, bias, kver = self._folded_bn()
        fuse_epilogue = not bn_training
        act_code = self._act_code if (fuse_epilogue and self._act_code is not None) else _native.ACT_NONE
        self._set_f16_scale(plan, x)
        # Inference steady state: the packed weight images of the previous forward are still in the workspace -- same kernel
        # tensor at the same version (torch bumps it on every in-place write: optimiser step, copy_, load_state_dict), same
        # arithmetic, same workspace -- so the call launches no weight-preparation kernel (DSPH_FWD_KEEP_WEIGHTS).
        # (and same batch class: the tile kernels pack four maps of a narrow layer into one item when there is more than one map)
        wkey = (kernel.data_ptr(), kver, self._prec_code(), (self.algo, N > 1),
                None if self._workspace is None else self._workspace.data_ptr())
        if getattr(self, "_use_graph", False):
            y = self._graph_forward(plan, x, bias if fuse_epilogue else None, act_code, wkey[:4], kernel)
        else:
            y, self._workspace = _native.cheb_forward(
                plan, x, kernel, bias if fuse_epilogue else None, self.K, act=act_code,
                precision=self._prec_code(), algo=_ALGOS[self.algo], workspace=self._workspace,
                basis=self._basis, keep_weights=getattr(self, "_wkey", None) == wkey,
            )
            self._wkey = wkey[:4] + (self._workspace.data_ptr() if self._workspace is not None else None,)
        if bn_training:  # BN -> bias -> activation, the reference's order (gnn_layers.py:152-159)
            was_training = self.bn.training
            self.bn.train(True)
            y = self.bn(y.transpose(1, 2)).transpose(1, 2).contiguous()
            self.bn.train(was_training)
            if bias is not None:
                y = y + self.bias.detach()
            if self.activation is not None:
                y = self.activation(y)
        elif self.activation is not None and self._act_code is None:
            y = self.activation(y)
        return y

    def _folded_bn(self):
        """Inference batch norm folded into the layer's parameters: ``(kernel * s, bias - mean * s, version)`` with
        ``s = 1 / sqrt(running_var + eps)`` per output channel, so that ``act(BN(conv(x)) + bias)`` is one kernel call with the
        epilogue fused (reference order BN -> bias -> activation, gnn_layers.py:152-159).  Cached: rebuilt (three tiny torch ops
        on [K*Fin, Fout]) only when the kernel, the bias or the moving statistics change -- their version counters move on every
        in-place write (optimiser step, ``load_state_dict``; a training-mode BN call moves ``num_batches_tracked``) -- and written into the SAME tensors, so the
        kept weight images of the steady state are keyed on a stable pointer and a fold counter."""
        bn = self.bn
        # (a training-mode call updates the moving statistics inside the framework's native batch norm WITHOUT moving their
        # version counters; it does bump num_batches_tracked, in Python)
        key = (self.kernel.data_ptr(), self.kernel._version, bn.running_mean._version, bn.running_var._version,
               bn.num_batches_tracked._version if bn.num_batches_tracked is not None else None,
               None if not self.use_bias else (self.bias.data_ptr(), self.bias._version))
        fold = getattr(self, "_bn_fold", None)
        if fold is None or fold["key"] != key:
            with torch.no_grad():
                s = torch.rsqrt(bn.running_var.to(torch.float32) + bn.eps)
                kf = self.kernel.detach() * s
                bf = -bn.running_mean.to(torch.float32) * s
                if self.use_bias:
                    bf = bf + self.bias.detach().reshape(-1)
                if fold is None:
                    fold = {"kernel": kf.contiguous(), "bias": bf.contiguous(), "count": 0}
                else:
                    fold["kernel"].copy_(kf)
                    fold["bias"].copy_(bf)
                fold["key"] = key
                fold["count"] += 1
            self._bn_fold = fold
        return fold["kernel"], fold["bias"], ("bn-fold", fold["count"])

    def _set_f16_scale(self, plan, x):
        """``precision="f16x3"``: the power of two x is multiplied by on its way into the f16 split (DSPH_OPT_F16_XEXP) --
        max|x| 2^e in [2^13, 2^14), from the caller's ``x_absmax`` (a known bound on |x|: no extra work) or, when the caller
        named none, from a reduction over x on the device and ONE host synchronisation per forward."""
        if self.precision != "f16x3":
            return
        amax = getattr(self, "x_absmax", None)
        if amax is None:
            amax = float(x.abs().amax())
        e = 0
        if np.isfinite(amax) and amax > 0.0:
            e = int(np.clip(14 - np.frexp(amax)[1], -100, 100))
        if getattr(self, "_f16_xexp", None) != e:
            plan.set_option(_native.OPT_F16_XEXP, e)
            self._f16_xexp = e

    call = forward

    def forward_pool(self, input_tensor, pool_type="MAX"):
        """``HealpyPool(p=1, pool_type)(self(input_tensor))`` in ONE pass where the kernels can (``dsph_poly_forward_pool``: a
        layer with activation none or ReLU (batch norm, if any, with its moving statistics: folded), a width that is a multiple of four, inference, whole unsharded maps --
        every shape but 64 -> 64 on maps large enough for the Clenshaw strips): the strip kernel reduces the four NEST children in its epilogue and the full-resolution output, 16 times the
        input of a 1 -> 16 layer, is never written.  Returns ``None`` when it cannot -- the caller then runs the two layers
        (``HealpyGCNN.forward`` does).  Same values as the two layers (bit for bit for "MAX")."""
        if pool_type not in ("MAX", "AVG") or not isinstance(input_tensor, torch.Tensor) or input_tensor.dim() != 3:
            return None
        if self._act_code not in (_native.ACT_NONE, _native.ACT_RELU) or getattr(self, "_use_graph", False):
            return None
        if torch.is_grad_enabled() and (input_tensor.requires_grad or (self._built and (self.kernel.requires_grad or (
                self.use_bias and self.bias.requires_grad)))):
            return None
        N, M, Fin = (int(v) for v in input_tensor.shape)
        if M != self._M or not input_tensor.is_cuda:
            return None
        self._resolve_device(input_tensor)
        if not self._built:
            if torch.is_grad_enabled():
                return None  # (new parameters require grad)
            self.build(input_tensor.shape)
        if Fin != self._Fin:
            return None
        plan = self._get_plan()
        Fout = int(self.kernel.shape[1])
        if not _native.pool_fusable(plan, N, Fin, Fout, self.K, self._act_code):
            return None
        x = input_tensor.detach().to(device=self._device, dtype=torch.float32).contiguous()
        bias = self.bias.detach().reshape(-1).contiguous() if self.use_bias else None
        kernel, kver = self.kernel.detach(), self.kernel._version
        if self.use_bn:  # (inference: the moving statistics folded into weights and bias, as in forward)
            kernel, bias, kver = self._folded_bn()
        wkey = (kernel.data_ptr(), kver, self._prec_code(), (self.algo, N > 1),
                None if self._workspace is None else self._workspace.data_ptr())
        y, self._workspace = _native.cheb_forward_pool(
            plan, x, kernel, bias, self.K, pool_type=_native.POOL_MAX if pool_type == "MAX" else _native.POOL_AVG,
            act=self._act_code, precision=self._prec_code(), workspace=self._workspace,
            basis=self._basis, keep_weights=getattr(self, "_wkey", None) == wkey)
        self._wkey = wkey[:4] + (self._workspace.data_ptr() if self._workspace is not None else None,)
        return y

    def invalidate_weights(self):
        """Forget the packed weight images (and a captured graph): the next forward re-packs.  The layer notices weight updates
        by the version counter of ``self.kernel`` -- optimiser steps, ``copy_``, ``load_state_dict`` all move it -- but an
        in-place write through ``kernel.data`` does not; call this after one."""
        self._wkey = None
        self._graph = None

    def _graph_forward(self, plan, x, bias, act_code, wkey, kernel):
        """The prepared forward as one HIP-graph launch (``graph=True``).  Captured after an ordinary forward has packed the
        weight images and sized the workspace; the captured call keeps them (DSPH_FWD_KEEP_WEIGHTS), so the graph holds the
        compute kernels only -- and under capture the BFS-tile launch always runs beside the structured ones
        (csrc/cheb_fused.hip).  Re-captured when the input buffer, the shapes, the weights' version or the epilogue change."""
        key = (x.data_ptr(), tuple(x.shape), wkey, None if bias is None else bias.data_ptr(), act_code)
        g = getattr(self, "_graph", None)
        if g is None or g["key"] != key:
            kw = dict(act=act_code, precision=self._prec_code(), algo=_ALGOS[self.algo], basis=self._basis)
            y, self._workspace = _native.cheb_forward(plan, x, kernel, bias, self.K, workspace=self._workspace, **kw)
            self._wkey = None
            graph = torch.cuda.CUDAGraph()
            out = torch.empty_like(y)
            cur = torch.cuda.current_stream(x.device)
            side = torch.cuda.Stream(device=x.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    _native.cheb_forward(plan, x, kernel, bias, self.K, workspace=self._workspace, out=out, keep_weights=True, **kw)
            cur.wait_stream(side)
            g = self._graph = {"key": key, "graph": graph, "out": out, "hold": (x, kernel, bias, self._workspace)}
        g["graph"].replay()
        return g["out"]

    @classmethod
    def from_prepared_ell(cls, ell_cols, ell_vals, K, lmax=None, **kwargs):
        """Layer over an already rescaled Laplacian given as padded ELL arrays (cols int32,
        vals float32, shape [M, W]), skipping the ARPACK eigen-solve of the constructor -- for
        graphs too large for it (nside >= 512) and for tests that share one prepared L~ between
        the oracle and the kernels.  Not part of the reference API."""
        self = cls.__new__(cls)
        torch.nn.Module.__init__(self)
        self.L = None
        self.K = int(K)
        if self.K < 1:
            raise ValueError("K must be at least 1")
        self.Fout = kwargs.pop("Fout", None)
        self.use_bias = kwargs.pop("use_bias", False)
        self.use_bn = kwargs.pop("use_bn", False)
        self.bn = None
        self.initializer = kwargs.pop("initializer", None)
        self.activation, self._act_code = _resolve_activation(kwargs.pop("activation", None))
        self.n_matmul_splits = kwargs.pop("n_matmul_splits", 1)
        device = kwargs.pop("device", None)
        self.precision = kwargs.pop("precision", DEFAULT_PRECISION)
        self.algo = kwargs.pop("algo", "auto")
        self._plan_options = dict(kwargs.pop("plan_options", None) or {})
        self._use_graph = bool(kwargs.pop("graph", False))
        self.x_absmax = kwargs.pop("x_absmax", None)
        if self.precision not in _PRECISIONS or self.algo not in _ALGOS:
            raise ValueError("unknown precision or algo")
        self.kwargs = kwargs
        self._device = torch.device(device) if device is not None else None
        self.lmax = lmax
        cols = np.ascontiguousarray(ell_cols, dtype=np.int32)
        vals = np.ascontiguousarray(ell_vals, dtype=np.float32)
        if cols.ndim != 2 or cols.shape != vals.shape:
            raise ValueError("ELL arrays must have shape [M, W]")
        self._M = cols.shape[0]
        self._ell_cols, self._ell_vals = cols, vals
        self._nnz = int(np.count_nonzero(vals))
        self._plan = None
        self._plan_t = None
        self._workspace = None
        self._workspace_t = None
        self.kernel = None
        self.bias = None
        self._built = False
        return self


class Monomial(Chebyshev):
    """A graph convolutional layer using monomials: T_k = L~^k with L~ = 2/lmax * L - I.

    Mirror of the reference's ``gnn_layers.Monomial`` (``gnn_layers.py:164-309``): same constructor, same
    forward structure as ``Chebyshev`` with the recurrence ``x_k = L~ x_{k-1}`` (``:283-286``),
    ``rescale_L(L, lmax)`` at its default scale 1 (``:219``) and a truncated-normal default
    initialiser of stddev 0.1 (``:243``).  Runs the same HIP kernels with ``DSPH_BASIS_MONOMIAL``.
    """

    _basis = _native.BASIS_MONOMIAL
    _scale = 1.0

    def _default_stddev(self, Fin):
        return 0.1


class GCNN_ResidualLayer(torch.nn.Module):
    """A generic residual layer:  in -> layer -> [norm] -> layer -> [norm] -> act(out + alpha * in).

    Mirror of the reference's ``gnn_layers.GCNN_ResidualLayer`` (``gnn_layers.py:312-413``), including its
    quirks: both sub-layers get the same ``layer_kwargs``; with ``activation=None`` the skip is
    ``x + input`` and ``alpha`` is ignored (``:407-408``); unknown ``layer_type`` raises ``IOError``
    (``:370``), unknown ``norm_type`` ``ValueError`` (``:379``).
    """

    def __init__(self, layer_type, layer_kwargs, activation=None, act_before=False, use_bn=False,
                 norm_type="batch_norm", bn_kwargs=None, alpha=1.0):
        super().__init__()
        self.layer_type = layer_type
        self.layer_kwargs = layer_kwargs
        self.activation, self._act_code = _resolve_activation(activation)
        self.act_before = act_before
        self.use_bn = use_bn
        self.norm_type = norm_type
        if bn_kwargs is None:
            self.bn_kwargs = {"axis": -1}
        else:
            self.bn_kwargs = bn_kwargs
            if "axis" not in self.bn_kwargs and norm_type != "moving_norm":
                self.bn_kwargs.update({"axis": -1})
        if self.layer_type == "CHEBY":
            self.layer1 = Chebyshev(**self.layer_kwargs)
            self.layer2 = Chebyshev(**self.layer_kwargs)
        elif self.layer_type == "MONO":
            self.layer1 = Monomial(**self.layer_kwargs)
            self.layer2 = Monomial(**self.layer_kwargs)
        else:
            raise IOError(f"Layertype not understood: {self.layer_type}")
        self.bn1 = self.bn2 = None
        if use_bn and norm_type not in ("layer_norm", "batch_norm"):
            raise ValueError(f"norm_type <{norm_type}> not understood!")
        self.alpha = alpha

    def _norm(self, which, x, training):
        """Keras BatchNormalization (momentum 0.99, eps 1e-3, affine) / LayerNormalization (eps 1e-3) over
        ``bn_kwargs['axis']`` of a (batch, nodes, channels) tensor, created on first use."""
        axis = self.bn_kwargs.get("axis", -1)
        mod = getattr(self, which)
        if mod is None:
            if self.norm_type == "batch_norm":
                mod = torch.nn.BatchNorm1d(x.shape[-1], eps=1e-3, momentum=0.01, affine=True).to(x.device)
            else:
                axes = (axis,) if isinstance(axis, int) else tuple(axis)
                shape = [x.shape[a] for a in axes]
                mod = torch.nn.LayerNorm(shape, eps=1e-3).to(x.device)
                mod._axes = axes
            setattr(self, which, mod)
        if self.norm_type == "batch_norm":
            was = mod.training
            mod.train(self.training if training is None else bool(training))
            y = mod(x.transpose(1, 2)).transpose(1, 2)
            mod.train(was)
            return y
        axes = [a % x.dim() for a in mod._axes]
        if axes == list(range(x.dim() - len(axes), x.dim())):
            return mod(x)
        raise NotImplementedError("layer_norm over non-trailing axes")

    def forward(self, input_tensor, training=False):
        if not isinstance(input_tensor, torch.Tensor):
            input_tensor = torch.as_tensor(np.asarray(input_tensor))
        x = self.layer1(input_tensor, training=training)
        inp = input_tensor.to(device=x.device, dtype=torch.float32)
        if self.use_bn:
            x = self._norm("bn1", x, training)
        x = self.layer2(x, training=training)
        if self.use_bn:
            x = self._norm("bn2", x, training)
        # inference on the GPU: skip connection and activation in ONE elementwise pass over the map (dsph_residual_epilogue)
        # instead of the host framework's three or four; with autograd on, or an activation the kernels do not know, the
        # host framework composes them as the reference does (gnn_layers.py:407-413)
        native = (x.is_cuda and inp.is_cuda and inp.shape == x.shape and x.dtype == torch.float32
                  and not (torch.is_grad_enabled() and (x.requires_grad or inp.requires_grad))
                  and (self.activation is None or self._act_code is not None))
        if native:
            x = x.contiguous()
            if self.activation is None:
                return _native.residual_epilogue(x, inp.contiguous(), 1.0, _native.ACT_NONE, False)
            return _native.residual_epilogue(x, inp.contiguous(), float(self.alpha), self._act_code, bool(self.act_before))
        if self.activation is None:
            return x + inp
        if self.act_before:
            return self.activation(x) + self.alpha * inp
        return self.activation(x + self.alpha * inp)

    call = forward


__all__ = ["Chebyshev", "Monomial", "GCNN_ResidualLayer"]
