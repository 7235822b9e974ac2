"""CPU restatement of the reference's Chebyshev graph-convolution forward.  TEST INFRASTRUCTURE.

PARITY UNPINNED.  The reference (``/root/reference/src/deepsphere``) is Python on
TensorFlow; tensorflow, keras, healpy and pygsp are installed on neither box, so the
reference cannot be imported to generate vectors, and its own tests hold no golden
vector or numeric assertion for this path (``tests/test_gnn_layers.py:9-33`` and
``tests/test_healpy_layers.py:66-85`` only call the layer).  This file therefore
*defines* the expected result by following the reference op for op; it is anchored
by (a) the literal transpose/reshape/stack chain below being checked against an
independent closed-form derivation, and (b) known-answer tests that need no oracle
(``tests/test_oracle.py``: L = I, eigenvector inputs, K = 1, one-hot weights).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package.  Nothing under ``deepsphere-cosmo-tf2_amd/``
does; the product path fails loudly when its HIP library is missing.

Third-party arithmetic the reference delegates to (not vendored under /root/reference):
``tensorflow>=2.14.0`` (``setup.cfg:17``): ``tf.sparse.sparse_dense_matmul`` at
``utils.py:73,76`` and ``tf.matmul`` at ``gnn_layers.py:149`` -- restated here as
scipy CSR @ dense and numpy matmul, i.e. the textbook definitions;
``scipy.sparse.linalg.eigsh`` at ``gnn_layers.py:66`` -- scipy is present and is called
the same way.
"""

import numpy as np
from scipy import sparse
from scipy.sparse.linalg import eigsh

# ----------------------------------------------------------------------------------------
# Laplacian preparation
# ----------------------------------------------------------------------------------------


def rescale_L(L, lmax=2, scale=1):
    """``utils.rescale_L`` (utils.py:40-46): L <- L*(2*scale/lmax) - I.  Works on a copy
    (the reference scales the caller's data in place; results do not
    depend on that because the preparation is scale invariant)."""
    L = sparse.csr_matrix(L, dtype=np.float64, copy=True)
    M = L.shape[0]
    L = L * (2.0 * scale / lmax)
    L = L - sparse.identity(M, format="csr", dtype=np.float64)
    return L.tocsr()


def prepare_L(L, scale=0.75):
    """``Chebyshev.__init__`` L-prep (gnn_layers.py:64-72).

    Returns (Lt, lmax): Lt = rescaled Laplacian as CSR with float32 values (the reference
    stores ``tf.constant(L_coo.data, dtype=floatx())``, gnn_layers.py:71), lmax = 1.02 *
    largest-magnitude eigenvalue from ARPACK (gnn_layers.py:66).
    """
    Lc = sparse.csr_matrix(L, dtype=np.float64)
    if Lc.shape[0] <= 2:
        # ARPACK needs k < M - 1; tiny matrices are solved densely
        ev = np.linalg.eigvalsh(Lc.toarray())
        lam = ev[np.argmax(np.abs(ev))]
    else:
        lam = eigsh(Lc, k=1, which="LM", return_eigenvectors=False)[0]
    lmax = 1.02 * lam
    Lt = rescale_L(Lc, lmax=lmax, scale=scale)
    Lt.sort_indices()  # tf.sparse.reorder, gnn_layers.py:115
    return Lt.astype(np.float32), float(lmax)


# ----------------------------------------------------------------------------------------
# activations the reference can look up by name in tf.keras.activations (gnn_layers.py:55-60)
# ----------------------------------------------------------------------------------------


def _elu(v):
    return np.where(v > 0, v, np.expm1(np.minimum(v, 0)))


ACTIVATIONS = {
    "linear": lambda v: v,
    "relu": lambda v: np.maximum(v, 0),
    "elu": _elu,
    "sigmoid": lambda v: 1.0 / (1.0 + np.exp(-v)),
    "tanh": np.tanh,
    "softplus": lambda v: np.logaddexp(v, 0.0),
    "softsign": lambda v: v / (1.0 + np.abs(v)),
    "selu": lambda v: 1.0507009873554805 * np.where(v > 0, v, 1.6732632423543772 * np.expm1(np.minimum(v, 0))),
    "swish": lambda v: v / (1.0 + np.exp(-v)),
    "silu": lambda v: v / (1.0 + np.exp(-v)),
    "leaky_relu": lambda v: np.where(v > 0, v, 0.2 * v),
    "exponential": np.exp,
}


def _resolve_activation(activation):
    if activation is None or callable(activation):
        return activation
    if activation in ACTIVATIONS:
        return ACTIVATIONS[activation]
    raise ValueError(f"Could not find activation <{activation}> in tf.keras.activations...")


# ----------------------------------------------------------------------------------------
# forward
# ----------------------------------------------------------------------------------------


def split_sparse_dense_matmul(Lt, dense, n_splits=1):
    """``utils.split_sparse_dense_matmul`` (utils.py:49-78): optional even column split of the
    dense operand, one sparse @ dense per chunk, concatenated -- numerically the unsplit
    product."""
    if n_splits > 1:
        if dense.shape[1] % n_splits != 0:
            raise ValueError("n_splits must divide the number of dense columns (tf.split)")
        parts = np.split(dense, n_splits, axis=1)
        return np.concatenate([Lt @ p for p in parts], axis=1)
    return Lt @ dense


def chebyshev_forward(
    Lt,
    x,
    kernel,
    K,
    bias=None,
    activation=None,
    bn=None,
    n_matmul_splits=1,
    dtype=np.float64,
):
    """Literal restatement of ``Chebyshev.call`` (gnn_layers.py:106-161).

    Lt      rescaled Laplacian (CSR, as returned by ``prepare_L``)
    x       (N, M, Fin)
    kernel  (K*Fin, Fout); row index = f*K + k (gnn_layers.py:144-149)
    bias    (1, 1, Fout) / (Fout,) or None             (gnn_layers.py:155-156)
    bn      None, or (mean, var) per output channel applied as (y-mean)/sqrt(var+1e-5)
            -- BatchNormalization(center=False, scale=False, epsilon=1e-5), gnn_layers.py:53
    dtype   arithmetic type: float64 = specification, float32 = what TF computes in
    Every intermediate keeps the reference's shape and layout on purpose.
    """
    x = np.asarray(x, dtype=dtype)
    kernel = np.asarray(kernel, dtype=dtype)
    Lt = sparse.csr_matrix(Lt).astype(dtype)
    N, M, Fin = x.shape
    assert Lt.shape == (M, M)
    assert kernel.shape[0] == K * Fin
    Fout = kernel.shape[1]

    x0 = np.transpose(x, (1, 2, 0))  # M x Fin x N          (:131)
    x0 = np.reshape(x0, (M, -1))  # M x Fin*N               (:132)
    stack = [x0]
    if K > 1:
        x1 = split_sparse_dense_matmul(Lt, x0, n_matmul_splits)  # (:138)
        stack.append(x1)
    for _k in range(2, K):
        x2 = 2 * split_sparse_dense_matmul(Lt, x1, n_matmul_splits) - x0  # (:141)
        stack.append(x2)
        x0, x1 = x1, x2
    xs = np.stack(stack, axis=0)  # K x M x Fin*N           (:144)
    xs = np.reshape(xs, (K, M, Fin, -1))  # K x M x Fin x N (:145)
    xs = np.transpose(xs, (3, 1, 2, 0))  # N x M x Fin x K  (:146)
    xs = np.reshape(xs, (-1, Fin * K))  # N*M x Fin*K       (:147)
    y = xs @ kernel  # N*M x Fout                           (:149)
    y = np.reshape(y, (-1, M, Fout))  # N x M x Fout        (:150)

    if bn is not None:  # (:152-153)
        mean, var = bn
        y = (y - np.asarray(mean, dtype=dtype)) / np.sqrt(np.asarray(var, dtype=dtype) + dtype(1e-5))
    if bias is not None:  # (:155-156)
        y = y + np.reshape(np.asarray(bias, dtype=dtype), (1, 1, Fout))
    act = _resolve_activation(activation)
    if act is not None:  # (:158-159)
        y = act(y)
    return y.astype(dtype, copy=False)


def monomial_forward(Lt, x, kernel, K, bias=None, activation=None, bn=None, n_matmul_splits=1, dtype=np.float64):
    """Literal restatement of ``Monomial.call`` (gnn_layers.py:262-309): as ``chebyshev_forward`` with the
    recurrence x_k = Lt x_{k-1} (:283-286).  Lt comes from ``prepare_L(L, scale=1)`` (:219)."""
    x = np.asarray(x, dtype=dtype)
    kernel = np.asarray(kernel, dtype=dtype)
    Lt = sparse.csr_matrix(Lt).astype(dtype)
    N, M, Fin = x.shape
    Fout = kernel.shape[1]
    x0 = np.reshape(np.transpose(x, (1, 2, 0)), (M, -1))  # (:276-277)
    stack = [x0]
    for _k in range(1, K):  # (:283-286)
        x1 = split_sparse_dense_matmul(Lt, x0, n_matmul_splits)
        stack.append(x1)
        x0 = x1
    xs = np.stack(stack, axis=0)  # (:288)
    xs = np.reshape(xs, (K, M, Fin, -1))
    xs = np.transpose(xs, (3, 1, 2, 0))
    xs = np.reshape(xs, (-1, Fin * K))
    y = np.reshape(xs @ kernel, (-1, M, Fout))  # (:293-294)
    if bn is not None:
        mean, var = bn
        y = (y - np.asarray(mean, dtype=dtype)) / np.sqrt(np.asarray(var, dtype=dtype) + dtype(1e-5))
    if bias is not None:
        y = y + np.reshape(np.asarray(bias, dtype=dtype), (1, 1, Fout))
    act = _resolve_activation(activation)
    if act is not None:
        y = act(y)
    return y.astype(dtype, copy=False)


def chebyshev_planes(Lt, x, K, dtype=np.float64):
    """T_k(Lt) x for k = 0..K-1 in the caller's (N, M, Fin) layout, shape (K, N, M, Fin)."""
    x = np.asarray(x, dtype=dtype)
    Lt = sparse.csr_matrix(Lt).astype(dtype)
    N, M, Fin = x.shape
    flat = np.transpose(x, (1, 0, 2)).reshape(M, N * Fin)
    planes = [flat]
    if K > 1:
        planes.append(Lt @ flat)
    for _k in range(2, K):
        planes.append(2 * (Lt @ planes[-1]) - planes[-2])
    out = np.stack(planes, axis=0).reshape(K, M, N, Fin)
    return np.transpose(out, (0, 2, 1, 3))


def monomial_planes(Lt, x, K, dtype=np.float64):
    """Lt^k x for k = 0..K-1 (the planes of Monomial.call, gnn_layers.py:262-309), shape (K, N, M, Fin)."""
    x = np.asarray(x, dtype=dtype)
    Lt = sparse.csr_matrix(Lt).astype(dtype)
    N, M, Fin = x.shape
    planes = [np.transpose(x, (1, 0, 2)).reshape(M, N * Fin)]
    for _k in range(1, K):
        planes.append(Lt @ planes[-1])
    out = np.stack(planes, axis=0).reshape(K, M, N, Fin)
    return np.transpose(out, (0, 2, 1, 3))


def chebyshev_forward_closed_form(Lt, x, kernel, K, dtype=np.float64):
    """Independent second derivation: dense Chebyshev matrices T_k and
    y = einsum('kmp,npf,fko->nmo', T, x, kernel.reshape(Fin, K, Fout)).  Small M only."""
    x = np.asarray(x, dtype=dtype)
    kernel = np.asarray(kernel, dtype=dtype)
    Ld = np.asarray(sparse.csr_matrix(Lt).astype(dtype).todense())
    N, M, Fin = x.shape
    Fout = kernel.shape[1]
    T = [np.eye(M, dtype=dtype)]
    if K > 1:
        T.append(Ld.copy())
    for _k in range(2, K):
        T.append(2 * Ld @ T[-1] - T[-2])
    T = np.stack(T, axis=0)
    Wr = kernel.reshape(Fin, K, Fout)
    return np.einsum("kmp,npf,fko->nmo", T, x, Wr)


def default_kernel_stddev(Fin, K):
    """Scale of the reference's default TruncatedNormal initialiser (gnn_layers.py:92)."""
    return 1.0 / np.sqrt(Fin * (K + 0.5) / 2.0)


# ----------------------------------------------------------------------------------------
# backward.  The reference defines no custom gradient: TensorFlow differentiates the
# op sequence of gnn_layers.py:131-150.  These are the closed forms of that derivative.
# ----------------------------------------------------------------------------------------


def chebyshev_backward(Lt, x, kernel, K, dy, dtype=np.float64):
    """Gradients of  y = chebyshev_forward(Lt, x, kernel, K)  (no bias / activation) w.r.t. x and
    kernel for an upstream gradient dy (N, M, Fout).

        dkernel[f*K + k, o] = sum_{n,m} (T_k(Lt) x)[n,m,f] * dy[n,m,o]
        dx                  = sum_k T_k(Lt)^T (dy @ kernel[:, k, :]^T)

    Lt need not be symmetric (the layer accepts any L, gnn_layers.py:17)."""
    x = np.asarray(x, dtype=dtype)
    dy = np.asarray(dy, dtype=dtype)
    kernel = np.asarray(kernel, dtype=dtype)
    N, M, Fin = x.shape
    Fout = kernel.shape[1]
    planes = chebyshev_planes(Lt, x, K, dtype=dtype)  # (K, N, M, Fin)
    dW = np.einsum("knmf,nmo->fko", planes, dy).reshape(Fin * K, Fout)
    Wr = kernel.reshape(Fin, K, Fout)
    LtT = sparse.csr_matrix(Lt).T.tocsr()
    dx = np.zeros_like(x)
    for k in range(K):
        g = dy @ Wr[:, k, :].T  # (N, M, Fin)
        dx += chebyshev_planes(LtT, g, k + 1, dtype=dtype)[k]
    return dx, dW


# ----------------------------------------------------------------------------------------
# residual block: gnn_layers.GCNN_ResidualLayer, gnn_layers.py:312-413
# ----------------------------------------------------------------------------------------


def keras_batch_norm(v, axis=-1, training=False, moving_mean=None, moving_var=None, gamma=None, beta=None, eps=1e-3):
    """``tf.keras.layers.BatchNormalization(axis=axis)`` at its defaults (epsilon 1e-3, center and scale on, gamma 1,
    beta 0, moving mean 0 / variance 1 at creation; what GCNN_ResidualLayer builds at gnn_layers.py:376-377 from
    ``bn_kwargs = {"axis": -1}``, :357-361).  training: statistics of the batch over every axis but ``axis`` (biased
    variance); inference: the moving statistics.  Third-party arithmetic (Keras, not vendored): the published
    definition (v - mean) / sqrt(var + eps) * gamma + beta."""
    v = np.asarray(v)
    ax = axis % v.ndim
    red = tuple(a for a in range(v.ndim) if a != ax)
    shape = [1] * v.ndim
    shape[ax] = v.shape[ax]
    if training:
        mean, var = v.mean(axis=red, keepdims=True), v.var(axis=red, keepdims=True)
    else:
        mean = np.zeros(shape, v.dtype) if moving_mean is None else np.reshape(np.asarray(moving_mean, v.dtype), shape)
        var = np.ones(shape, v.dtype) if moving_var is None else np.reshape(np.asarray(moving_var, v.dtype), shape)
    out = (v - mean) / np.sqrt(var + v.dtype.type(eps))
    if gamma is not None:
        out = out * np.reshape(np.asarray(gamma, v.dtype), shape)
    if beta is not None:
        out = out + np.reshape(np.asarray(beta, v.dtype), shape)
    return out


def keras_layer_norm(v, axis=-1, gamma=None, beta=None, eps=1e-3):
    """``tf.keras.layers.LayerNormalization(axis=axis)`` at its defaults (epsilon 1e-3; gnn_layers.py:373-374): every
    sample normalised over ``axis`` with the biased variance, then gamma / beta over the same axes."""
    v = np.asarray(v)
    axes = (axis,) if isinstance(axis, int) else tuple(axis)
    axes = tuple(a % v.ndim for a in axes)
    mean, var = v.mean(axis=axes, keepdims=True), v.var(axis=axes, keepdims=True)
    out = (v - mean) / np.sqrt(var + v.dtype.type(eps))
    shape = [v.shape[a] if a in axes else 1 for a in range(v.ndim)]
    if gamma is not None:
        out = out * np.reshape(np.asarray(gamma, v.dtype), shape)
    if beta is not None:
        out = out + np.reshape(np.asarray(beta, v.dtype), shape)
    return out


def residual_forward(Lt, x, kernels, K, layer_type="CHEBY", layer_biases=(None, None), layer_activation=None,
                     activation=None, act_before=False, use_bn=False, norm_type="batch_norm", bn_axis=-1,
                     bn_params=(None, None), training=False, alpha=1.0, dtype=np.float64):
    """Literal restatement of ``GCNN_ResidualLayer.call`` (gnn_layers.py:385-413).

    Lt                the prepared Laplacian both sub-layers hold (they are built from the same ``layer_kwargs``, :364-369;
                      CHEBY: ``prepare_L(L, 0.75)``, MONO: ``prepare_L(L, 1)``)
    kernels           (kernel of layer1, kernel of layer2), each (K*F, F)
    layer_biases      their biases or None; layer_activation: the ``activation`` entry of ``layer_kwargs`` (both layers)
    bn_params         per norm layer None or a dict of ``keras_batch_norm`` / ``keras_layer_norm`` keyword arguments
    The quirks kept: ``activation is None`` returns x + input, ``alpha`` ignored (:407-408); an unknown ``layer_type``
    raises IOError (:370), an unknown ``norm_type`` ValueError (:379)."""
    if layer_type == "CHEBY":
        fwd = chebyshev_forward
    elif layer_type == "MONO":
        fwd = monomial_forward
    else:
        raise IOError(f"Layertype not understood: {layer_type}")
    if use_bn and norm_type not in ("layer_norm", "batch_norm"):
        raise ValueError(f"norm_type <{norm_type}> not understood!")

    def norm(v, p):
        p = dict(p or {})
        if norm_type == "layer_norm":
            return keras_layer_norm(v, axis=bn_axis, **p)
        return keras_batch_norm(v, axis=bn_axis, training=training, **p)

    inp = np.asarray(x, dtype=dtype)
    v = fwd(Lt, inp, kernels[0], K, bias=layer_biases[0], activation=layer_activation, dtype=dtype)  # (:393)
    if use_bn:  # (:396-397)
        v = norm(v, bn_params[0])
    v = fwd(Lt, v, kernels[1], K, bias=layer_biases[1], activation=layer_activation, dtype=dtype)  # (:400)
    if use_bn:  # (:403-404)
        v = norm(v, bn_params[1])
    act = _resolve_activation(activation)
    if act is None:  # (:407-408)
        return v + inp
    if act_before:  # (:410-411)
        return act(v) + alpha * inp
    return act(v + alpha * inp)  # (:413)


# ----------------------------------------------------------------------------------------
# the callers either side of the convolution: healpy_layers.HealpyPool / HealpyPseudoConv(_Transpose)
# ----------------------------------------------------------------------------------------


def healpy_pool(x, p, pool_type="MAX"):
    """``HealpyPool.call`` (healpy_layers.py:20-85): Keras MaxPool1D / AveragePooling1D with pool_size = strides = 4^p,
    padding "valid", channels last, on (batch, nodes, channels) -- i.e. the maximum / mean over each run of 4^p consecutive
    nodes (the NEST children of one coarse pixel).  Raises IOError like the reference for p < 1, an unknown type
    (:39-40, :64-65) or a node count that is not a multiple of 4^p (:73-75)."""
    if not p >= 1:
        raise IOError("The reduction factors has to be at least 2!")
    if pool_type not in ("MAX", "AVG"):
        raise IOError(f"Pooling type not understood: {pool_type}")
    x = np.asarray(x)
    g = int(4 ** p)
    N, M, F = x.shape
    if M % g != 0:
        raise IOError(f"Input shape {x.shape} not compatible with the filter size {g}")
    blocks = x.reshape(N, M // g, g, F)
    return blocks.max(axis=2) if pool_type == "MAX" else blocks.mean(axis=2)


def healpy_pseudo_conv(x, kernel, bias, p):
    """``HealpyPseudoConv.call`` (healpy_layers.py:88-146): Keras Conv1D(Fout, 4^p, strides=4^p, padding "valid",
    channels last); ``kernel`` in the Keras layout (4^p, Fin, Fout):  y[n, m, o] = sum_{i, f} x[n, 4^p m + i, f] kernel[i, f, o] + bias[o]."""
    x = np.asarray(x)
    g = int(4 ** p)
    N, M, Fin = x.shape
    kernel = np.asarray(kernel)
    assert kernel.shape[:2] == (g, Fin)
    y = np.einsum("nmif,ifo->nmo", x.reshape(N, M // g, g, Fin), kernel)
    return y if bias is None else y + np.asarray(bias)


def healpy_pseudo_conv_transpose(x, kernel, bias, p):
    """``HealpyPseudoConv_Transpose.call`` (healpy_layers.py:156-216): Conv2DTranspose with kernel = strides = (1, 4^p) on the
    map viewed as (batch, 1, nodes, channels); ``kernel`` (4^p, Fout, Fin) like Keras stores it (the leading 1 dropped):
    y[n, 4^p m + i, o] = sum_f x[n, m, f] kernel[i, o, f] + bias[o]."""
    x = np.asarray(x)
    g = int(4 ** p)
    N, M, Fin = x.shape
    kernel = np.asarray(kernel)
    assert kernel.shape[0] == g and kernel.shape[2] == Fin
    y = np.einsum("nmf,iof->nmio", x, kernel).reshape(N, M * g, kernel.shape[1])
    return y if bias is None else y + np.asarray(bias)
