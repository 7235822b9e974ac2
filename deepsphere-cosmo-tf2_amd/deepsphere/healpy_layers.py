"""HEALPix-aware layer specs for the Chebyshev path.

Mirror of the reference's ``deepsphere.healpy_layers.HealpyChebyshev``
(``/root/reference/src/deepsphere/healpy_layers.py:219-264``): a deferred description of a
Chebyshev layer that a model builder turns into a real layer once it has computed the graph
Laplacian of the current resolution (``healpy_networks.py:110-137``).
"""

import numpy as np
import torch

from . import _native
from .gnn_layers import Chebyshev, GCNN_ResidualLayer, Monomial


def _as_tensor(x):
    return x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x), dtype=torch.float32)


class _NestPoolFunction(torch.autograd.Function):
    """HealpyPool on the GPU: ``dsph_healpix_pool`` forward, ``dsph_healpix_pool_backward`` for the input gradient (the
    reference gets the latter from TensorFlow's autodiff of the Keras pooling layer)."""

    @staticmethod
    def forward(ctx, x, group, pool_type):
        ctx.group, ctx.pool_type = group, pool_type
        ctx.save_for_backward(x if pool_type == _native.POOL_MAX else x.new_empty(0))
        return _native.healpix_pool(x, group, pool_type)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return _native.healpix_pool_backward(x if ctx.pool_type == _native.POOL_MAX else None, dy, ctx.group, ctx.pool_type), None, None


class HealpyPool(torch.nn.Module):
    """Pooling over the 4^p NEST children of a HEALPix pixel (reference ``healpy_layers.py:20-78``: a Keras
    MaxPool1D / AveragePooling1D with size = stride = 4^p on (batch, pixels, channels)).  On a HIP device: the
    ``dsph_healpix_pool`` kernels (one contiguous run of 4^p rows per output row); on the CPU (shape checks, tests without a
    GPU) the same reduction as a strided op of the host framework.  It only works for NEST ordering."""

    def __init__(self, p, pool_type="MAX", **kwargs):
        super().__init__()
        if not p >= 1:
            raise IOError("The reduction factors has to be at least 2!")
        self.p = p
        self.filter_size = int(4**p)
        self.pool_type = pool_type
        self.kwargs = kwargs
        if pool_type not in ("MAX", "AVG"):
            raise IOError(f"Pooling type not understood: {self.pool_type}")

    def forward(self, input_tensor):
        x = _as_tensor(input_tensor)
        N, M, F = x.shape
        if M % self.filter_size != 0:
            raise IOError(f"Input shape {tuple(x.shape)} not compatible with the filter size {self.filter_size}")
        if x.is_cuda and x.dtype == torch.float32:
            return _NestPoolFunction.apply(x.contiguous(), self.filter_size,
                                           _native.POOL_MAX if self.pool_type == "MAX" else _native.POOL_AVG)
        # (not a HIP float32 tensor -- a CPU tensor, another dtype: the same reduction as a strided op of the host framework.
        # The one host-framework branch of the package, kept for shape checks and tests without a GPU; it is not on the
        # convolution's path and DESIGN.md 1 says so.)
        x = x.reshape(N, M // self.filter_size, self.filter_size, F)
        return x.amax(dim=2) if self.pool_type == "MAX" else x.mean(dim=2)

    call = forward


class HealpyPseudoConv(torch.nn.Module):
    """Learnable 4^p -> 1 reduction of NEST children (reference ``healpy_layers.py:81-146``: Conv1D with
    kernel = stride = 4^p, channels last).  Weights: ``filter.weight`` (Fout, Fin, 4^p), ``filter.bias``."""

    def __init__(self, p, Fout, kernel_initializer=None, **kwargs):
        super().__init__()
        if not p >= 1:
            raise IOError("The reduction factors has to be at least 1!")
        self.p = p
        self.filter_size = int(4**p)
        self.Fout = Fout
        self.kernel_initializer = kernel_initializer
        self.kwargs = kwargs
        self.filter = None

    def build(self, input_shape):
        if int(input_shape[1]) % self.filter_size != 0:
            raise IOError(f"Input shape {tuple(input_shape)} not compatible with the filter size {self.filter_size}")
        self.filter = torch.nn.Conv1d(int(input_shape[-1]), self.Fout, self.filter_size, stride=self.filter_size)
        torch.nn.init.xavier_uniform_(self.filter.weight)  # Keras default glorot_uniform, zero bias
        torch.nn.init.zeros_(self.filter.bias)
        if self.kernel_initializer is not None:
            self.kernel_initializer(self.filter.weight)

    def forward(self, input_tensor):
        x = _as_tensor(input_tensor)
        if self.filter is None:
            self.build(x.shape)
            self.filter.to(x.device)
        # kernel = stride = 4^p on NEST-ordered rows: the 4^p children of an output pixel are consecutive rows, so the input
        # is, without any copy, a (N * M / 4^p) x (4^p * Fin) matrix and the layer one plain library GEMM against the
        # [4^p * Fin, Fout] view of the Conv1D weights (no transposes of the map, which Conv1d on channels-last data needs)
        N, M, Fin = x.shape
        g = self.filter_size
        w2 = self.filter.weight.permute(2, 1, 0).reshape(g * Fin, self.Fout)  # row (i, f) <- weight[o, f, i]
        y = torch.addmm(self.filter.bias, x.reshape(N * (M // g), g * Fin), w2.to(x.dtype))
        return y.reshape(N, M // g, self.Fout)

    call = forward


class HealpyPseudoConv_Transpose(torch.nn.Module):
    """Learnable 1 -> 4^p expansion into NEST children (reference ``healpy_layers.py:149-216``: a
    Conv2DTranspose with kernel = stride = (1, 4^p))."""

    def __init__(self, p, Fout, kernel_initializer=None, **kwargs):
        super().__init__()
        if not p >= 1:
            raise IOError("The boost factors has to be at least 1!")
        self.p = p
        self.filter_size = int(4**p)
        self.Fout = Fout
        self.kernel_initializer = kernel_initializer
        self.kwargs = kwargs
        self.filter = None

    def build(self, input_shape):
        if int(input_shape[1]) % self.filter_size != 0:  # the reference checks the same thing (:203-204)
            raise IOError(f"Input shape {tuple(input_shape)} not compatible with the filter size {self.filter_size}")
        self.filter = torch.nn.ConvTranspose1d(int(input_shape[-1]), self.Fout, self.filter_size,
                                               stride=self.filter_size)
        torch.nn.init.xavier_uniform_(self.filter.weight)
        torch.nn.init.zeros_(self.filter.bias)
        if self.kernel_initializer is not None:
            self.kernel_initializer(self.filter.weight)

    def forward(self, input_tensor):
        x = _as_tensor(input_tensor)
        if self.filter is None:
            self.build(x.shape)
            self.filter.to(x.device)
        # the mirror image: every input pixel writes its 4^p NEST children, consecutive rows of the output -- one GEMM of the
        # (N * M) x Fin map against the [Fin, 4^p * Fout] view of the transposed-convolution weights, reshaped for free
        N, M, Fin = x.shape
        g = self.filter_size
        w2 = self.filter.weight.permute(0, 2, 1).reshape(Fin, g * self.Fout)  # column (i, o) <- weight[f, o, i]
        y = x.reshape(N * M, Fin) @ w2.to(x.dtype)
        return y.reshape(N, M * g, self.Fout) + self.filter.bias

    call = forward


class HealpyChebyshev:
    """Deferred spec of a Chebyshev graph convolution on a HEALPix map."""

    def __init__(self, K, Fout=None, initializer=None, activation=None, use_bias=False, use_bn=False, **kwargs):
        """
        :param K: number of polynomial terms
        :param Fout: output channels, defaults to the number of input channels
        :param initializer: weight initialiser, see ``Chebyshev``
        :param activation: activation by name or callable, see ``Chebyshev``
        :param use_bias: learnable bias
        :param use_bn: batch norm before the bias
        :param kwargs: forwarded to the layer
        """
        self.K = K
        self.Fout = Fout  # read by the model builder to track the channel count (healpy_networks.py:160-164)
        self.initializer = initializer
        self.activation = activation
        self.use_bias = use_bias
        self.use_bn = use_bn
        self.kwargs = kwargs

    def _get_layer(self, L, n_matmul_splits=1):
        """Instantiate the layer for graph Laplacian ``L``.

        :param L: the graph Laplacian of the map's pixels (NEST order)
        :param n_matmul_splits: the builder's split count for TensorFlow's sparse matmul; accepted,
            not needed by the HIP kernels
        :return: a callable ``Chebyshev`` layer
        """
        return Chebyshev(
            L=L,
            K=self.K,
            Fout=self.Fout,
            initializer=self.initializer,
            activation=self.activation,
            use_bias=self.use_bias,
            use_bn=self.use_bn,
            n_matmul_splits=n_matmul_splits,
            **self.kwargs,
        )


class HealpyMonomial(HealpyChebyshev):
    """Deferred spec of a monomial graph convolution (reference ``healpy_layers.py:267-313``)."""

    def _get_layer(self, L, n_matmul_splits=1):
        return Monomial(L=L, K=self.K, Fout=self.Fout, initializer=self.initializer, activation=self.activation,
                        use_bias=self.use_bias, use_bn=self.use_bn, n_matmul_splits=n_matmul_splits, **self.kwargs)


class Healpy_ResidualLayer:
    """Deferred spec of a residual block of two graph convolutions (reference ``healpy_layers.py:316-378``).

    ``layer_kwargs`` lacks ``L``; ``_get_layer`` adds it (and ``n_matmul_splits``) to a copy -- the reference
    writes them into the caller's dict."""

    def __init__(self, layer_type, layer_kwargs, activation=None, act_before=False, use_bn=False,
                 norm_type="batch_norm", bn_kwargs=None, alpha=1.0):
        self.layer_type = layer_type
        self.layer_kwargs = layer_kwargs
        self.activation = activation
        self.act_before = act_before
        self.use_bn = use_bn
        self.norm_type = norm_type
        self.bn_kwargs = bn_kwargs
        self.alpha = alpha

    def _get_layer(self, L, n_matmul_splits=1):
        kwargs = dict(self.layer_kwargs)
        kwargs.update({"L": L, "n_matmul_splits": n_matmul_splits})
        return GCNN_ResidualLayer(layer_type=self.layer_type, layer_kwargs=kwargs, activation=self.activation,
                                  act_before=self.act_before, use_bn=self.use_bn, norm_type=self.norm_type,
                                  bn_kwargs=self.bn_kwargs, alpha=self.alpha)


__all__ = ["HealpyPool", "HealpyPseudoConv", "HealpyPseudoConv_Transpose", "HealpyChebyshev", "HealpyMonomial",
           "Healpy_ResidualLayer"]
