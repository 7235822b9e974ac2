"""HEALPix NEST geometry and graph-Laplacian producer (host side, numpy/scipy only).

The reference obtains its Laplacians from two third-party packages that exist on
neither box (healpy, and the ``SphereHealpix`` class of a pygsp fork):
``healpy_networks.py:110-118`` calls ``SphereHealpix(subdivisions=nside,
indexes=indices, nest=True, k=n_neighbors, lap_type="normalized").L``.  This module
produces Laplacians of the same family from the published HEALPix algorithm
(Gorski et al. 2005): pixel centres, the 8 grid neighbours of a pixel, and two
graph builders (symmetrised k-NN with a Gaussian kernel, and the fixed 8-neighbour
grid stencil of the north-star text).  The convolution never depends on *how* L was
produced (L is an input of the layer, ``gnn_layers.py:17``), so nothing here is on
the parity path; it only feeds tests and the benchmark with realistic matrices.
"""

import numpy as np
from scipy import sparse

__all__ = [
    "nside2npix",
    "npix2nside",
    "isnsideok",
    "nest2xyf",
    "xyf2nest",
    "pix2vec",
    "neighbours",
    "kernel_width",
    "healpix_graph",
    "healpix_laplacian",
    "grid_laplacian_ell",
    "grid_laplacian_ell_torch",
    "cap_indices",
    "extend_indices",
]

# base-pixel ("face") ring/phi offsets of the HEALPix projection
_JRLL = np.array([2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4], dtype=np.int64)
_JPLL = np.array([1, 3, 5, 7, 0, 2, 4, 6, 1, 3, 5, 7], dtype=np.int64)

# neighbour direction order: SW, W, NW, N, NE, E, SE, S
_NB_XOFF = np.array([-1, -1, 0, 1, 1, 1, 0, -1], dtype=np.int64)
_NB_YOFF = np.array([0, 1, 1, 1, 0, -1, -1, -1], dtype=np.int64)
# which face a step off the edge of a face lands in; row = 3*(dy+1) + (dx+1), -1: no face
_NB_FACE = np.array(
    [
        [8, 9, 10, 11, -1, -1, -1, -1, 10, 11, 8, 9],  # S
        [5, 6, 7, 4, 8, 9, 10, 11, 9, 10, 11, 8],  # SE
        [-1, -1, -1, -1, 5, 6, 7, 4, -1, -1, -1, -1],  # E
        [4, 5, 6, 7, 11, 8, 9, 10, 11, 8, 9, 10],  # SW
        [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11],  # centre
        [1, 2, 3, 0, 0, 1, 2, 3, 5, 6, 7, 4],  # NE
        [-1, -1, -1, -1, 7, 4, 5, 6, -1, -1, -1, -1],  # W
        [3, 0, 1, 2, 3, 0, 1, 2, 4, 5, 6, 7],  # NW
        [2, 3, 0, 1, -1, -1, -1, -1, 0, 1, 2, 3],  # N
    ],
    dtype=np.int64,
)
# coordinate fix-up when crossing into that face, indexed [row][face >> 2]:
# bit0 flip x, bit1 flip y, bit2 swap x and y
_NB_SWAP = np.array(
    [[0, 0, 3], [0, 0, 6], [0, 0, 0], [0, 0, 5], [0, 0, 0], [5, 0, 0], [0, 0, 0], [6, 0, 0], [3, 0, 0]],
    dtype=np.int64,
)


def nside2npix(nside):
    return 12 * int(nside) * int(nside)


def isnsideok(nside):
    nside = int(nside)
    return nside > 0 and (nside & (nside - 1)) == 0


def npix2nside(npix):
    nside = int(round(np.sqrt(npix / 12.0)))
    if 12 * nside * nside != npix:
        raise ValueError(f"{npix} is not a valid HEALPix pixel count")
    return nside


def _compress_bits(v):
    """Keep the even bits of v (uint64) and pack them into the low half."""
    v = v & np.uint64(0x5555555555555555)
    v = (v | (v >> np.uint64(1))) & np.uint64(0x3333333333333333)
    v = (v | (v >> np.uint64(2))) & np.uint64(0x0F0F0F0F0F0F0F0F)
    v = (v | (v >> np.uint64(4))) & np.uint64(0x00FF00FF00FF00FF)
    v = (v | (v >> np.uint64(8))) & np.uint64(0x0000FFFF0000FFFF)
    v = (v | (v >> np.uint64(16))) & np.uint64(0x00000000FFFFFFFF)
    return v


def _spread_bits(v):
    """Inverse of _compress_bits: put bit i of v at bit 2i."""
    v = v & np.uint64(0x00000000FFFFFFFF)
    v = (v | (v << np.uint64(16))) & np.uint64(0x0000FFFF0000FFFF)
    v = (v | (v << np.uint64(8))) & np.uint64(0x00FF00FF00FF00FF)
    v = (v | (v << np.uint64(4))) & np.uint64(0x0F0F0F0F0F0F0F0F)
    v = (v | (v << np.uint64(2))) & np.uint64(0x3333333333333333)
    v = (v | (v << np.uint64(1))) & np.uint64(0x5555555555555555)
    return v


def nest2xyf(nside, pix):
    """NEST index -> (ix, iy, face); NEST index = face*nside^2 + morton(ix, iy)."""
    pix = np.asarray(pix, dtype=np.int64)
    npface = int(nside) * int(nside)
    face = pix // npface
    p = (pix % npface).astype(np.uint64)
    ix = _compress_bits(p).astype(np.int64)
    iy = _compress_bits(p >> np.uint64(1)).astype(np.int64)
    return ix, iy, face


def xyf2nest(nside, ix, iy, face):
    ix = np.asarray(ix, dtype=np.int64).astype(np.uint64)
    iy = np.asarray(iy, dtype=np.int64).astype(np.uint64)
    m = _spread_bits(ix) | (_spread_bits(iy) << np.uint64(1))
    return np.asarray(face, dtype=np.int64) * (int(nside) * int(nside)) + m.astype(np.int64)


def pix2vec(nside, pix=None, dtype=np.float64):
    """Unit vectors of NEST pixel centres, shape (len(pix), 3)."""
    nside = int(nside)
    if pix is None:
        pix = np.arange(nside2npix(nside), dtype=np.int64)
    ix, iy, face = nest2xyf(nside, pix)
    nl4 = 4 * nside
    npix = 12 * nside * nside
    fact2 = 4.0 / npix
    fact1 = (2 * nside) * fact2
    jr = _JRLL[face] * nside - ix - iy - 1
    north = jr < nside
    south = jr > 3 * nside
    belt = ~(north | south)
    nr = np.where(north, jr, np.where(south, nl4 - jr, nside))
    z = np.empty(jr.shape, dtype=np.float64)
    z[north] = 1.0 - (nr[north].astype(np.float64) ** 2) * fact2
    z[south] = (nr[south].astype(np.float64) ** 2) * fact2 - 1.0
    z[belt] = (2 * nside - jr[belt]) * fact1
    kshift = np.where(belt, (jr - nside) & 1, 0)
    jp = (_JPLL[face] * nr + ix - iy + 1 + kshift) // 2
    jp = np.where(jp > nl4, jp - nl4, jp)
    jp = np.where(jp < 1, jp + nl4, jp)
    phi = (jp - (kshift + 1) * 0.5) * ((np.pi / 2) / nr)
    st = np.sqrt(np.maximum(0.0, (1.0 - z) * (1.0 + z)))
    out = np.empty(jr.shape + (3,), dtype=dtype)
    out[..., 0] = st * np.cos(phi)
    out[..., 1] = st * np.sin(phi)
    out[..., 2] = z
    return out


def neighbours(nside, pix=None):
    """The 8 grid neighbours (SW, W, NW, N, NE, E, SE, S) of NEST pixels; -1 where absent.

    Every pixel has 8 of them except the 24 pixels at the 8 corners where only three
    faces meet, which have 7.
    """
    nside = int(nside)
    if pix is None:
        pix = np.arange(nside2npix(nside), dtype=np.int64)
    pix = np.asarray(pix, dtype=np.int64)
    ix, iy, face = nest2xyf(nside, pix)
    out = np.empty(pix.shape + (8,), dtype=np.int64)
    for i in range(8):
        x = ix + _NB_XOFF[i]
        y = iy + _NB_YOFF[i]
        nb = np.full(pix.shape, 4, dtype=np.int64)
        lo = x < 0
        hi = x >= nside
        x = np.where(lo, x + nside, np.where(hi, x - nside, x))
        nb = nb - lo + hi
        lo = y < 0
        hi = y >= nside
        y = np.where(lo, y + nside, np.where(hi, y - nside, y))
        nb = nb - 3 * lo + 3 * hi
        f = _NB_FACE[nb, face]
        bits = _NB_SWAP[nb, face >> 2]
        x = np.where(bits & 1, nside - x - 1, x)
        y = np.where(bits & 2, nside - y - 1, y)
        sw = (bits & 4) != 0
        x, y = np.where(sw, y, x), np.where(sw, x, y)
        res = xyf2nest(nside, x, y, np.maximum(f, 0))
        out[..., i] = np.where(f >= 0, res, -1)
    return out


# Gaussian kernel widths of the reference's graph producer for k = 8 and 20 neighbours
# (recalled from the pygsp fork, unverifiable here; other values are
# extrapolated ~ c_k / nside).  Treated as a parameter: the convolution takes L as input.
_KW_TABLE = {
    8: {32: 0.02500, 64: 0.01228, 128: 0.00614, 256: 0.00307, 512: 0.00154, 1024: 0.00077},
    20: {32: 0.03185, 64: 0.01564, 128: 0.00782, 256: 0.00391, 512: 0.00196, 1024: 0.00098},
}
_KW_COEFF = {8: 0.786, 20: 1.001, 40: 1.3, 60: 1.55}


def kernel_width(nside, n_neighbors=8):
    tab = _KW_TABLE.get(int(n_neighbors), {})
    if int(nside) in tab:
        return tab[int(nside)]
    return _KW_COEFF.get(int(n_neighbors), 0.786 * np.sqrt(n_neighbors / 8.0)) / float(nside)


def _normalized_laplacian(W):
    d = np.asarray(W.sum(axis=1)).ravel()
    with np.errstate(divide="ignore"):
        dis = np.where(d > 0, 1.0 / np.sqrt(d), 0.0)
    D = sparse.diags(dis)
    M = W.shape[0]
    return (sparse.identity(M, format="csr", dtype=np.float64) - D @ W @ D).tocsr()


def healpix_graph(nside, indices=None, n_neighbors=8, mode="knn", kw=None):
    """Weighted adjacency W (CSR, float64, symmetric, zero diagonal) on NEST pixels.

    mode="knn":  symmetrised k-nearest-neighbour graph on the 3-D pixel centres with weights
                 exp(-(d/kw)^2) -- the family ``SphereHealpix(k=n_neighbors)`` produces
                 (``healpy_networks.py:110-118``); rows end up with k..k+2 neighbours.
    mode="grid": the fixed 8-neighbour HEALPix stencil (7 at the 24 corner pixels) with the
                 same kernel; cheap enough for nside >= 1024.
    ``indices``: sorted NEST pixel ids of a partial-sky map; the graph is built on that subset
                 only, like the reference passing ``indexes=current_indices``.
    """
    nside = int(nside)
    npix = nside2npix(nside)
    if indices is None:
        indices = np.arange(npix, dtype=np.int64)
    indices = np.asarray(indices, dtype=np.int64)
    M = indices.shape[0]
    if kw is None:
        kw = kernel_width(nside, n_neighbors)
    vec = pix2vec(nside, indices)
    if mode == "knn":
        from scipy.spatial import cKDTree

        k = int(n_neighbors)
        tree = cKDTree(vec)
        dist, nbr = tree.query(vec, k=k + 1)
        rows = np.repeat(np.arange(M, dtype=np.int64), k)
        cols = nbr[:, 1:].reshape(-1).astype(np.int64)
        w = np.exp(-((dist[:, 1:].reshape(-1) / kw) ** 2))
        A = sparse.csr_matrix((w, (rows, cols)), shape=(M, M))
        # symmetrise: an edge present in one direction is filled in the other
        W = A.maximum(A.T).tocsr()
    elif mode == "grid":
        if int(n_neighbors) != 8:
            raise NotImplementedError("the grid stencil has 8 neighbours")
        nb = neighbours(nside, indices)
        full = M == npix
        if not full:
            lut = np.full(npix, -1, dtype=np.int64)
            lut[indices] = np.arange(M, dtype=np.int64)
            nb = np.where(nb >= 0, lut[np.maximum(nb, 0)], -1)
        rows = np.repeat(np.arange(M, dtype=np.int64), 8)
        cols = nb.reshape(-1)
        ok = cols >= 0
        rows, cols = rows[ok], cols[ok]
        d = np.linalg.norm(vec[rows] - vec[cols], axis=1)
        w = np.exp(-((d / kw) ** 2))
        W = sparse.csr_matrix((w, (rows, cols)), shape=(M, M))
        W = W.maximum(W.T).tocsr()
    else:
        raise ValueError(f"unknown graph mode <{mode}>")
    W.setdiag(0.0)
    W.eliminate_zeros()
    W.sort_indices()
    return W


def healpix_laplacian(nside, indices=None, n_neighbors=8, mode="knn", kw=None):
    """Normalised Laplacian L = I - D^-1/2 W D^-1/2 (CSR float64) of ``healpix_graph``."""
    return _normalized_laplacian(healpix_graph(nside, indices, n_neighbors, mode, kw))


def grid_laplacian_ell(nside, kw=None, dtype=np.float64, chunk=1 << 22):
    """Full-sphere 8-neighbour normalised Laplacian directly in padded-ELL form.

    Returns (cols int32 [M,9], vals dtype [M,9]); slot 0 is the diagonal (1.0), slots 1..8
    the grid neighbours in ``neighbours`` order, absent ones padded with (col=row, val=0).
    Same matrix as ``healpix_laplacian(nside, mode="grid")`` without the scipy round trip,
    which at nside 1024 (12.6 M rows) costs minutes.
    """
    nside = int(nside)
    npix = nside2npix(nside)
    if kw is None:
        kw = kernel_width(nside, 8)
    cols = np.empty((npix, 9), dtype=np.int32)
    w = np.empty((npix, 9), dtype=np.float64)
    for s in range(0, npix, chunk):
        e = min(npix, s + chunk)
        p = np.arange(s, e, dtype=np.int64)
        nb = neighbours(nside, p)
        v0 = pix2vec(nside, p)
        cols[s:e, 0] = p
        w[s:e, 0] = 0.0
        for j in range(8):
            ok = nb[:, j] >= 0
            q = np.where(ok, nb[:, j], p)
            d = np.linalg.norm(pix2vec(nside, q) - v0, axis=1)
            cols[s:e, j + 1] = q
            w[s:e, j + 1] = np.where(ok, np.exp(-((d / kw) ** 2)), 0.0)
    deg = w.sum(axis=1)
    dis = 1.0 / np.sqrt(deg)
    vals = np.empty((npix, 9), dtype=dtype)
    for s in range(0, npix, chunk):
        e = min(npix, s + chunk)
        vals[s:e] = (-(w[s:e] * dis[s:e, None]) * dis[cols[s:e]]).astype(dtype)
        vals[s:e, 0] = 1.0
    return cols, vals


def cap_indices(nside, vec=(1.0, 0.0, 0.0), fraction=1.0 / 3.0):
    """Sorted NEST ids of the spherical cap around ``vec`` covering ``fraction`` of the sphere."""
    v = np.asarray(vec, dtype=np.float64)
    v = v / np.linalg.norm(v)
    cos_lim = 1.0 - 2.0 * float(fraction)
    c = pix2vec(int(nside)) @ v
    return np.nonzero(c >= cos_lim)[0].astype(np.int64)


def extend_indices(indices, nside_in, nside_out, nest=True):
    """Minimal superset of ``indices`` that coarsens cleanly to ``nside_out``.

    Same contract as the reference's ``utils.extend_indices`` (``utils.py:9-37``, which
    degrades and re-upgrades a 0/1 map with healpy): in NEST order a pixel's parent at
    nside_out is ``pix >> 2*log2(nside_in/nside_out)``, so the answer is every child of
    every touched parent.  RING order is not supported here.
    """
    if not nest:
        raise NotImplementedError("only NEST ordering is supported")
    indices = np.asarray(indices, dtype=np.int64)
    ratio = int(nside_in) // int(nside_out)
    if ratio < 1 or ratio * int(nside_out) != int(nside_in) or not isnsideok(ratio):
        raise ValueError("nside_in must be nside_out times a power of two")
    per = ratio * ratio
    parents = np.unique(indices // per)
    return (parents[:, None] * per + np.arange(per, dtype=np.int64)[None, :]).reshape(-1)


# ----------------------------------------------------------------------------------------------
# torch twin of the grid-stencil builder: same algorithm on torch tensors, so that a 12.6 M-pixel
# (nside 1024) Laplacian is produced in well under a second on the GPU instead of minutes in numpy.
# Checked against the numpy functions above in tests/test_host.py.
# ----------------------------------------------------------------------------------------------


def _t_compress(v):
    v = v & 0x5555555555555555
    v = (v | (v >> 1)) & 0x3333333333333333
    v = (v | (v >> 2)) & 0x0F0F0F0F0F0F0F0F
    v = (v | (v >> 4)) & 0x00FF00FF00FF00FF
    v = (v | (v >> 8)) & 0x0000FFFF0000FFFF
    v = (v | (v >> 16)) & 0x00000000FFFFFFFF
    return v


def _t_spread(v):
    v = v & 0x00000000FFFFFFFF
    v = (v | (v << 16)) & 0x0000FFFF0000FFFF
    v = (v | (v << 8)) & 0x00FF00FF00FF00FF
    v = (v | (v << 4)) & 0x0F0F0F0F0F0F0F0F
    v = (v | (v << 2)) & 0x3333333333333333
    v = (v | (v << 1)) & 0x5555555555555555
    return v


def _t_pix2vec(nside, ix, iy, face, jrll, jpll):
    import torch

    nl4 = 4 * nside
    fact2 = 4.0 / (12 * nside * nside)
    fact1 = (2 * nside) * fact2
    jr = jrll[face] * nside - ix - iy - 1
    north = jr < nside
    south = jr > 3 * nside
    belt = ~(north | south)
    nr = torch.where(north, jr, torch.where(south, nl4 - jr, torch.full_like(jr, nside)))
    nrf = nr.to(torch.float64)
    z = torch.where(north, 1.0 - nrf * nrf * fact2,
                    torch.where(south, nrf * nrf * fact2 - 1.0, (2 * nside - jr).to(torch.float64) * fact1))
    kshift = torch.where(belt, (jr - nside) & 1, torch.zeros_like(jr))
    jp = torch.div(jpll[face] * nr + ix - iy + 1 + kshift, 2, rounding_mode="floor")
    jp = torch.where(jp > nl4, jp - nl4, jp)
    jp = torch.where(jp < 1, jp + nl4, jp)
    phi = (jp.to(torch.float64) - (kshift + 1).to(torch.float64) * 0.5) * ((np.pi / 2) / nrf)
    st = torch.sqrt(torch.clamp((1.0 - z) * (1.0 + z), min=0.0))
    return torch.stack((st * torch.cos(phi), st * torch.sin(phi), z), dim=-1)


def grid_laplacian_ell_torch(nside, kw=None, device="cpu", chunk=1 << 22):
    """``grid_laplacian_ell`` on torch tensors: (cols int32 [M,9], vals float64 [M,9]) on ``device``."""
    import torch

    nside = int(nside)
    npface = nside * nside
    npix = 12 * npface
    if kw is None:
        kw = kernel_width(nside, 8)
    dev = torch.device(device)
    jrll = torch.as_tensor(_JRLL, device=dev)
    jpll = torch.as_tensor(_JPLL, device=dev)
    nbface = torch.as_tensor(_NB_FACE, device=dev)
    nbswap = torch.as_tensor(_NB_SWAP, device=dev)
    cols = torch.empty((npix, 9), dtype=torch.int32, device=dev)
    w = torch.zeros((npix, 9), dtype=torch.float64, device=dev)
    for s in range(0, npix, chunk):
        e = min(npix, s + chunk)
        p = torch.arange(s, e, dtype=torch.int64, device=dev)
        face = torch.div(p, npface, rounding_mode="floor")
        pm = p - face * npface
        ix = _t_compress(pm)
        iy = _t_compress(pm >> 1)
        v0 = _t_pix2vec(nside, ix, iy, face, jrll, jpll)
        cols[s:e, 0] = p.to(torch.int32)
        for i in range(8):
            x = ix + int(_NB_XOFF[i])
            y = iy + int(_NB_YOFF[i])
            lo, hi = x < 0, x >= nside
            x = torch.where(lo, x + nside, torch.where(hi, x - nside, x))
            nb = 4 - lo.to(torch.int64) + hi.to(torch.int64)
            lo, hi = y < 0, y >= nside
            y = torch.where(lo, y + nside, torch.where(hi, y - nside, y))
            nb = nb - 3 * lo.to(torch.int64) + 3 * hi.to(torch.int64)
            f = nbface[nb, face]
            bits = nbswap[nb, face >> 2]
            x = torch.where((bits & 1) != 0, nside - x - 1, x)
            y = torch.where((bits & 2) != 0, nside - y - 1, y)
            sw = (bits & 4) != 0
            x, y = torch.where(sw, y, x), torch.where(sw, x, y)
            ok = f >= 0
            fq = torch.clamp(f, min=0)
            q = torch.where(ok, fq * npface + (_t_spread(x) | (_t_spread(y) << 1)), p)
            d = torch.linalg.norm(_t_pix2vec(nside, x, y, fq, jrll, jpll) - v0, dim=1)
            cols[s:e, i + 1] = q.to(torch.int32)
            w[s:e, i + 1] = torch.where(ok, torch.exp(-((d / kw) ** 2)), torch.zeros_like(d))
    dis = 1.0 / torch.sqrt(w.sum(dim=1))
    vals = -(w * dis[:, None]) * dis[cols.to(torch.int64)]
    vals[:, 0] = 1.0
    return cols, vals
