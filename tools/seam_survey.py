#!/usr/bin/env python3
"""Which of the tiles that the strips leave could a strip with a table of tile bases take?  CPU only.
A 16 x 16 tile is ELIGIBLE when its eight neighbouring tiles exist and every seam it looks across is a pure translation of the
pixel grid (no flip, no swap of x and y: all seams inside a base pixel, and the seams between an equatorial and a polar base
pixel; the seams between two polar base pixels are rotated).  Counts for BASELINE configs[2] (full sphere) and configs[4] (cap).
    python3 tools/seam_survey.py [nside]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deepsphere-cosmo-tf2_amd"))
from deepsphere import healpix as hp  # noqa: E402


def tile_neighbourhood(nt, tiles):
    """For tile ids (NEST at tile resolution nt per face side): [n, 8] neighbour tile ids (-1: none) and [n, 8] seam bits."""
    ix, iy, face = hp.nest2xyf(nt, tiles)
    nbr = np.empty(tiles.shape + (8,), dtype=np.int64)
    bits_out = np.zeros(tiles.shape + (8,), dtype=np.int64)
    for i in range(8):
        x = ix + hp._NB_XOFF[i]
        y = iy + hp._NB_YOFF[i]
        nb = np.full(tiles.shape, 4, dtype=np.int64)
        lo, hi = x < 0, x >= nt
        x = np.where(lo, x + nt, np.where(hi, x - nt, x))
        nb = nb - lo + hi
        lo, hi = y < 0, y >= nt
        y = np.where(lo, y + nt, np.where(hi, y - nt, y))
        nb = nb - 3 * lo + 3 * hi
        f = hp._NB_FACE[nb, face]
        bits = hp._NB_SWAP[nb, face >> 2]
        x2 = np.where(bits & 1, nt - x - 1, x)
        y2 = np.where(bits & 2, nt - y - 1, y)
        sw = (bits & 4) != 0
        x2, y2 = np.where(sw, y2, x2), np.where(sw, x2, y2)
        nbr[:, i] = np.where(f >= 0, hp.xyf2nest(nt, x2, y2, np.maximum(f, 0)), -1)
        bits_out[:, i] = np.where(nb == 4, 0, bits)
    return nbr, bits_out


def survey(name, nt, tiles):
    present = np.zeros(12 * nt * nt, dtype=bool)
    present[tiles] = True
    nbr, bits = tile_neighbourhood(nt, tiles)
    have = (nbr >= 0) & present[np.maximum(nbr, 0)]
    ix, iy, face = hp.nest2xyf(nt, tiles)
    eligible = have.all(axis=1) & (bits == 0).all(axis=1)
    print(f"{name}: {len(tiles)} tiles; all eight neighbours present {int(have.all(axis=1).sum())}; of those behind translation-only seams "
          f"(strip-eligible with a table of tile bases) {int(eligible.sum())}; the rest {int(len(tiles) - eligible.sum())}")
    return eligible


nside = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nt = nside // 16
full = np.arange(12 * nt * nt, dtype=np.int64)
e = survey(f"full sphere, nside {nside}", nt, full)
ix, iy, _ = hp.nest2xyf(nt, full)
inner = (ix > 0) & (ix < nt - 1) & (iy > 0) & (iy < nt - 1)
print(f"   today's strips: the {int(inner.sum())} tiles off the base pixels' border rings; eligible among the {int((~inner).sum())} others: {int((e & ~inner).sum())}")
idx = hp.extend_indices(hp.cap_indices(nside, fraction=1.0 / 3.0), nside, 8)
tiles = np.unique(np.asarray(idx, dtype=np.int64) // 256)
e5 = survey(f"cap of a third of the sphere, nside {nside}, padded to nside-8 superpixels", nt, tiles)
six, siy, _ = hp.nest2xyf(nt, tiles)
inner5 = ((six % 8) > 0) & ((six % 8) < 7) & ((siy % 8) > 0) & ((siy % 8) < 7)
print(f"   today's strips: the {int(inner5.sum())} tiles off the superpixels' border rings; eligible among the {int((~inner5).sum())} others: {int((e5 & ~inner5).sum())}")
