"""T0: the CPU oracle against itself, against known answers and against the committed vectors."""

import numpy as np
import pytest
from scipy import sparse

from helpers import CASES, load_case, rel_err
from oracle import cheb_oracle as orc


def _rand_graph_L(M, seed, density=0.2):
    rng = np.random.default_rng(seed)
    A = sparse.random(M, M, density=density, random_state=rng, format="csr")
    A = A + A.T
    A.setdiag(0)
    A.eliminate_zeros()
    d = np.asarray(A.sum(1)).ravel() + 1e-3
    D = sparse.diags(1 / np.sqrt(d))
    return (sparse.identity(M) - D @ A @ D).tocsr()


@pytest.mark.parametrize("K", [1, 2, 3, 5, 8])
def test_literal_equals_closed_form(K):
    L = _rand_graph_L(23, seed=K)
    Lt, _ = orc.prepare_L(L)
    rng = np.random.default_rng(100 + K)
    x = rng.standard_normal((3, 23, 5))
    W = rng.standard_normal((5 * K, 4))
    a = orc.chebyshev_forward(Lt, x, W, K)
    b = orc.chebyshev_forward_closed_form(Lt, x, W, K)
    assert rel_err(a, b) < 1e-12


def test_identity_laplacian_known_answer():
    # L = I => lmax = 1.02, L~ = (1.5/1.02 - 1) I, T_k(L~) = T_k(t) I with t = 0.470588...
    # (the L of the reference's residual-layer test, tests/test_gnn_layers.py:105)
    M, N, Fin, Fout, K = 192, 3, 7, 7, 5
    Lt, lmax = orc.prepare_L(np.eye(M))
    assert abs(lmax - 1.02) < 1e-12
    t = 1.5 / 1.02 - 1.0
    assert np.allclose(Lt.toarray(), np.float32(t) * np.eye(M))
    tf32 = float(np.float32(t))
    T = [1.0, tf32]
    for _ in range(2, K):
        T.append(2 * tf32 * T[-1] - T[-2])
    assert np.allclose(T, np.cos(np.arange(K) * np.arccos(tf32)))
    rng = np.random.default_rng(5)
    x = rng.standard_normal((N, M, Fin))
    W = rng.standard_normal((Fin * K, Fout))
    Weff = np.einsum("k,fko->fo", np.array(T), W.reshape(Fin, K, Fout))
    y = orc.chebyshev_forward(Lt, x, W, K)
    assert rel_err(y, x @ Weff) < 1e-13


def test_eigenvector_input():
    # T_k(L~) v = cos(k arccos(lam)) v for an eigenpair (lam, v) of L~
    L = _rand_graph_L(31, seed=3)
    Lt, _ = orc.prepare_L(L)
    lam, V = np.linalg.eigh(Lt.toarray().astype(np.float64))
    K = 6
    for j in (0, 7, 30):
        x = V[:, j][None, :, None]
        planes = orc.chebyshev_planes(Lt, x, K)
        for k in range(K):
            assert np.allclose(planes[k, 0, :, 0], np.cos(k * np.arccos(np.clip(lam[j], -1, 1))) * V[:, j], atol=1e-10)


def test_K1_is_pointwise_matmul():
    L = _rand_graph_L(17, seed=9)
    Lt, _ = orc.prepare_L(L)
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 17, 3))
    W = rng.standard_normal((3, 5))
    assert rel_err(orc.chebyshev_forward(Lt, x, W, 1), x @ W) < 1e-14


def test_one_hot_weight_row_selects_plane():
    # kernel row f*K + k (channel-major, order-minor): a one-hot row extracts T_k x[:, :, f]
    L = _rand_graph_L(19, seed=4)
    Lt, _ = orc.prepare_L(L)
    rng = np.random.default_rng(2)
    Fin, K = 3, 4
    x = rng.standard_normal((2, 19, Fin))
    planes = orc.chebyshev_planes(Lt, x, K)
    for f in range(Fin):
        for k in range(K):
            W = np.zeros((Fin * K, 1))
            W[f * K + k, 0] = 1.0
            y = orc.chebyshev_forward(Lt, x, W, K)
            assert np.allclose(y[..., 0], planes[k, :, :, f], atol=1e-13)


def test_split_invariance_and_linearity():
    L = _rand_graph_L(21, seed=6)
    Lt, _ = orc.prepare_L(L)
    rng = np.random.default_rng(3)
    x1, x2 = rng.standard_normal((2, 4, 21, 6))
    W = rng.standard_normal((6 * 5, 3))
    y1 = orc.chebyshev_forward(Lt, x1, W, 5)
    assert rel_err(orc.chebyshev_forward(Lt, x1, W, 5, n_matmul_splits=4), y1) < 1e-14
    y2 = orc.chebyshev_forward(Lt, x2, W, 5)
    assert rel_err(orc.chebyshev_forward(Lt, 2 * x1 - 3 * x2, W, 5), 2 * y1 - 3 * y2) < 1e-12
    # batch independence
    assert np.allclose(orc.chebyshev_forward(Lt, x1[1:2], W, 5), y1[1:2])


def test_prepare_L_scale_invariance_and_spectrum():
    L = _rand_graph_L(40, seed=8)
    A, la = orc.prepare_L(L)
    B, lb = orc.prepare_L(L * 3.7)
    assert abs(lb / la - 3.7) < 1e-9
    assert np.allclose(A.toarray(), B.toarray(), atol=1e-6)
    ev = np.linalg.eigvalsh(A.toarray().astype(np.float64))
    assert ev.min() >= -1 - 1e-6 and ev.max() <= 1.5 / 1.02 - 1 + 1e-6
    # the caller's matrix is not modified
    L0 = L.copy()
    orc.prepare_L(L)
    assert (abs(L - L0)).max() == 0


def test_epilogue_order_bn_bias_activation():
    L = _rand_graph_L(11, seed=2)
    Lt, _ = orc.prepare_L(L)
    rng = np.random.default_rng(7)
    x = rng.standard_normal((2, 11, 3))
    W = rng.standard_normal((3 * 3, 2))
    b = np.array([0.5, -2.0])
    base = orc.chebyshev_forward(Lt, x, W, 3)
    mean, var = np.array([0.1, -0.2]), np.array([2.0, 0.5])
    y = orc.chebyshev_forward(Lt, x, W, 3, bias=b, activation="relu", bn=(mean, var))
    assert np.allclose(y, np.maximum((base - mean) / np.sqrt(var + 1e-5) + b, 0))
    with pytest.raises(ValueError):
        orc.chebyshev_forward(Lt, x, W, 3, activation="not_an_activation")


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden(name):
    c = load_case(name)
    y = orc.chebyshev_forward(c["Lt"], c["x"], c["kernel"], c["K"], bias=c["bias"], activation=c["activation"])
    assert y.shape == c["y"].shape
    assert rel_err(y, c["y"]) < 1e-13
    # the fp32 restatement (what TF computes in) stays within the stated fp32 tolerance of the spec
    y32 = orc.chebyshev_forward(c["Lt"], c["x"], c["kernel"], c["K"], bias=c["bias"], activation=c["activation"],
                                dtype=np.float32)
    assert rel_err(y32, c["y"]) < 1e-5


def test_cpu_baseline_port_matches_oracle():
    import torch

    from oracle import cheb_cpu_baseline as cb

    c = load_case("n8_k5")
    y = cb.forward_fp32(cb.to_torch_csr(c["Lt"]), torch.from_numpy(c["x"]), torch.from_numpy(c["kernel"]), c["K"])
    assert rel_err(y.numpy(), c["y"]) < 1e-5


@pytest.mark.parametrize("symmetric", [True, False])
def test_backward_closed_form_matches_finite_differences(symmetric):
    rng = np.random.default_rng(21)
    M, N, Fin, Fout, K = 13, 2, 3, 2, 4
    if symmetric:
        Lt, _ = orc.prepare_L(_rand_graph_L(M, seed=5))
    else:
        Lt = sparse.random(M, M, density=0.3, random_state=rng, format="csr") * 0.3
    Lt = sparse.csr_matrix(Lt).astype(np.float64)
    x = rng.standard_normal((N, M, Fin))
    W = rng.standard_normal((Fin * K, Fout))
    dy = rng.standard_normal((N, M, Fout))
    dx, dW = orc.chebyshev_backward(Lt, x, W, K, dy)
    loss = lambda xx, ww: float((orc.chebyshev_forward(Lt, xx, ww, K) * dy).sum())  # noqa: E731
    eps = 1e-6
    for idx in [(0, 0, 0), (1, 7, 2), (0, 12, 1)]:
        e = np.zeros_like(x)
        e[idx] = eps
        assert abs((loss(x + e, W) - loss(x - e, W)) / (2 * eps) - dx[idx]) < 1e-6
    for idx in [(0, 0), (5, 1), (Fin * K - 1, 0)]:
        e = np.zeros_like(W)
        e[idx] = eps
        assert abs((loss(x, W + e) - loss(x, W - e)) / (2 * eps) - dW[idx]) < 1e-6


def test_monomial_literal_against_powers():
    L = _rand_graph_L(15, seed=12)
    Lt, _ = orc.prepare_L(L, scale=1.0)
    ev = np.linalg.eigvalsh(Lt.toarray().astype(np.float64))
    assert ev.min() >= -1 - 1e-6 and ev.max() <= 2 / 1.02 - 1 + 1e-6  # spectrum in [-1, 0.96]
    rng = np.random.default_rng(4)
    x = rng.standard_normal((2, 15, 3))
    W = rng.standard_normal((3 * 4, 2))
    Ld = Lt.toarray().astype(np.float64)
    P = [np.linalg.matrix_power(Ld, k) for k in range(4)]
    ref = np.einsum("kmp,npf,fko->nmo", np.stack(P), x, W.reshape(3, 4, 2))
    assert rel_err(orc.monomial_forward(Lt, x, W, 4), ref) < 1e-12


# ---- residual block (reference gnn_layers.py:312-413) ---------------------------------------------------------


def test_residual_identity_laplacian_known_answer():
    """The reference's own residual-layer test graph (tests/test_gnn_layers.py:94-114: L = I_192, x (3, 192, 7), K = 5,
    Fout = None): with L~ = t I every convolution is a per-pixel matrix product, so the whole block has a closed form."""
    M, N, F, K = 192, 3, 7, 5
    Lt, _ = orc.prepare_L(np.eye(M))
    t = float(np.float32(1.5 / 1.02 - 1.0))
    T = np.cos(np.arange(K) * np.arccos(t))
    rng = np.random.default_rng(21)
    x = rng.standard_normal((N, M, F))
    k1, k2 = rng.standard_normal((F * K, F)) * 0.3, rng.standard_normal((F * K, F)) * 0.3
    E1 = np.einsum("k,fko->fo", T, k1.reshape(F, K, F))
    E2 = np.einsum("k,fko->fo", T, k2.reshape(F, K, F))
    core = x @ E1 @ E2
    relu = lambda v: np.maximum(v, 0)  # noqa: E731
    assert rel_err(orc.residual_forward(Lt, x, (k1, k2), K, activation="relu", alpha=0.5), relu(core + 0.5 * x)) < 1e-12
    assert rel_err(orc.residual_forward(Lt, x, (k1, k2), K, activation="relu", act_before=True, alpha=0.5),
                   relu(core) + 0.5 * x) < 1e-12
    # activation None: x + input, alpha ignored (gnn_layers.py:407-408)
    assert rel_err(orc.residual_forward(Lt, x, (k1, k2), K, activation=None, alpha=7.0), core + x) < 1e-12
    # the sub-layers' own activation (layer_kwargs) sits inside both convolutions
    assert rel_err(orc.residual_forward(Lt, x, (k1, k2), K, layer_activation="relu", activation="relu"),
                   relu(relu(relu(x @ E1) @ E2) + x)) < 1e-12
    # monomial sub-layers: L~ = (2/1.02 - 1) I, T_k = t^k
    Lm, _ = orc.prepare_L(np.eye(M), scale=1)
    tm = float(np.float32(2 / 1.02 - 1.0))
    P = tm ** np.arange(K)
    M1 = np.einsum("k,fko->fo", P, k1.reshape(F, K, F))
    M2 = np.einsum("k,fko->fo", P, k2.reshape(F, K, F))
    assert rel_err(orc.residual_forward(Lm, x, (k1, k2), K, layer_type="MONO", activation="relu"), relu(x @ M1 @ M2 + x)) < 1e-12
    with pytest.raises(IOError):
        orc.residual_forward(Lt, x, (k1, k2), K, layer_type="juhu")
    with pytest.raises(ValueError):
        orc.residual_forward(Lt, x, (k1, k2), K, use_bn=True, norm_type="moving_norm")


@pytest.mark.parametrize("training", [False, True])
def test_residual_norms_against_an_independent_implementation(training):
    """The two norm layers of the block (Keras defaults: epsilon 1e-3, affine) against torch's modules, which implement the
    same published definitions independently: batch norm in both modes, layer norm over the channel axis."""
    import torch

    rng = np.random.default_rng(8)
    v = rng.standard_normal((4, 30, 6)) * 2 + 0.7
    mean, var = rng.standard_normal(6), rng.random(6) + 0.5
    gamma, beta = rng.standard_normal(6), rng.standard_normal(6)
    bn = torch.nn.BatchNorm1d(6, eps=1e-3, momentum=0.01).double()
    with torch.no_grad():
        bn.running_mean.copy_(torch.from_numpy(mean))
        bn.running_var.copy_(torch.from_numpy(var))
        bn.weight.copy_(torch.from_numpy(gamma))
        bn.bias.copy_(torch.from_numpy(beta))
    bn.train(training)
    with torch.no_grad():
        want = bn(torch.from_numpy(v).transpose(1, 2)).transpose(1, 2).numpy()
    got = orc.keras_batch_norm(v, axis=-1, training=training, moving_mean=mean, moving_var=var, gamma=gamma, beta=beta)
    assert rel_err(got, want) < 1e-12
    ln = torch.nn.LayerNorm(6, eps=1e-3).double()
    with torch.no_grad():
        ln.weight.copy_(torch.from_numpy(gamma))
        ln.bias.copy_(torch.from_numpy(beta))
        want = ln(torch.from_numpy(v)).numpy()
    assert rel_err(orc.keras_layer_norm(v, axis=-1, gamma=gamma, beta=beta), want) < 1e-12


def test_residual_block_with_norms_composes():
    """residual_forward with norms == the composition of its parts on a random graph (both sub-layers share Lt)."""
    L = _rand_graph_L(29, seed=4)
    Lt, _ = orc.prepare_L(L)
    rng = np.random.default_rng(14)
    N, F, K = 3, 5, 4
    x = rng.standard_normal((N, 29, F))
    k1, k2 = rng.standard_normal((F * K, F)) * 0.4, rng.standard_normal((F * K, F)) * 0.4
    b1, b2 = rng.standard_normal(F), rng.standard_normal(F)
    p1 = dict(moving_mean=rng.standard_normal(F), moving_var=rng.random(F) + 0.5, gamma=rng.standard_normal(F), beta=rng.standard_normal(F))
    p2 = dict(moving_mean=rng.standard_normal(F), moving_var=rng.random(F) + 0.5, gamma=rng.standard_normal(F), beta=rng.standard_normal(F))
    got = orc.residual_forward(Lt, x, (k1, k2), K, layer_biases=(b1, b2), layer_activation="elu", activation="tanh", use_bn=True,
                               bn_params=(p1, p2), alpha=0.3)
    v = orc.keras_batch_norm(orc.chebyshev_forward(Lt, x, k1, K, bias=b1, activation="elu"), **p1)
    v = orc.keras_batch_norm(orc.chebyshev_forward(Lt, v, k2, K, bias=b2, activation="elu"), **p2)
    assert rel_err(got, np.tanh(v + 0.3 * x)) < 1e-13


def test_pooling_restatements_known_answers():
    """HealpyPool / HealpyPseudoConv(_Transpose) (healpy_layers.py:20-216): known answers that need no oracle -- the children
    of coarse pixel m are the rows 4^p m .. 4^p (m+1) - 1; a pseudo-convolution with a one-hot kernel picks one child; the
    transposed one writes its input into one child."""
    rng = np.random.default_rng(2)
    x = rng.standard_normal((2, 48, 3))
    mx, av = orc.healpy_pool(x, 1, "MAX"), orc.healpy_pool(x, 1, "AVG")
    assert mx.shape == (2, 12, 3)
    assert np.array_equal(mx[1, 5], x[1, 20:24].max(axis=0)) and np.allclose(av[0, 11], x[0, 44:48].mean(axis=0))
    assert np.array_equal(orc.healpy_pool(x, 2, "MAX")[0, 2], x[0, 32:48].max(axis=0))
    for bad in (lambda: orc.healpy_pool(x, 0), lambda: orc.healpy_pool(x, 1, "MEAN"), lambda: orc.healpy_pool(x[:, :46], 1)):
        with pytest.raises(IOError):
            bad()
    k = np.zeros((4, 3, 2))
    k[2, 1, 0] = 1.0  # output channel 0 <- child 2, input channel 1
    y = orc.healpy_pseudo_conv(x, k, np.array([0.5, -1.0]), 1)
    assert np.allclose(y[..., 0], x[:, 2::4, 1] + 0.5) and np.allclose(y[..., 1], -1.0)
    kt = np.zeros((4, 2, 3))
    kt[3, 1, 2] = 2.0  # child 3, output channel 1 <- 2 * input channel 2
    z = orc.healpy_pseudo_conv_transpose(x, kt, None, 1)
    assert z.shape == (2, 192, 2) and np.allclose(z[:, 3::4, 1], 2.0 * x[:, :, 2]) and np.count_nonzero(z[:, 0::4]) == 0
