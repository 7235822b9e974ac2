"""HEALPix-aware layer specs for the Chebyshev path.

Mirror of the reference's ``deepsphere.healpy_layers.HealpyChebyshev``
(``/root/reference/src/deepsphere/healpy_layers.py:219-264``): a deferred description of a
Chebyshev layer that a model builder turns into a real layer once it has computed the graph
Laplacian of the current resolution (``healpy_networks.py:110-137``).
"""

from .gnn_layers import Chebyshev, GCNN_ResidualLayer, Monomial


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


__all__ = ["HealpyChebyshev", "HealpyMonomial", "Healpy_ResidualLayer"]
