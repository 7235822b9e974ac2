#!/usr/bin/env python3
"""How the reference's k-nearest-neighbour graphs (pygsp SphereHealpix k = 8, here healpix_laplacian(mode="knn")) split
between the fused kernels, next to the 8-neighbour grid stencil.  Usage (GPU box): tools/knn_tiles.py [nside ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "deepsphere-cosmo-tf2_amd"))
import numpy as np  # noqa: E402

from deepsphere import _native, healpix, utils  # noqa: E402

for nside in [int(v) for v in sys.argv[1:]] or [64, 128]:
    for mode in ("grid", "knn"):
        L = healpix.healpix_laplacian(nside, mode=mode)
        Lt, _ = utils.prepare_L(L)
        cols, vals = utils.csr_to_ell(Lt)
        plan = _native.LaplacianPlan(cols, vals, device=0)
        deg = np.diff(Lt.indptr) - 1
        try:
            ns, nb = plan.tile_counts(5)
        except RuntimeError as exc:
            ns, nb = -1, str(exc)
        print(f"nside {nside} {mode}: ELL width {cols.shape[1]}, rows with exactly 8 neighbours {float((deg == 8).mean()):.3f}, "
              f"tiles (structured, bfs) = ({ns}, {nb})", flush=True)
