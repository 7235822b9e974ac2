"""Shared helpers for the test-suite (not collected by pytest)."""

import os

import numpy as np
from scipy import sparse

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cheb_cases.npz")
CASES = ["dense3", "eye192", "n4_k1", "n4_k2", "n4_k5", "n8_k5", "n8_nb20_k8", "n8_cap_k5", "n8_grid_k5"]


def load_case(name):
    z = np.load(GOLDEN)
    data, indices, indptr = z[f"{name}/L_data"], z[f"{name}/L_indices"], z[f"{name}/L_indptr"]
    M = indptr.shape[0] - 1
    case = {
        "Lt": sparse.csr_matrix((data, indices, indptr), shape=(M, M)),
        "x": z[f"{name}/x"],
        "kernel": z[f"{name}/kernel"],
        "K": int(z[f"{name}/K"]),
        "y": z[f"{name}/y"],
        "bias": z[f"{name}/bias"] if f"{name}/bias" in z.files else None,
        "activation": str(z[f"{name}/activation"]) or None,
        "lmax": float(z[f"{name}/lmax"]),
    }
    return case


def rel_err(a, b):
    """max |a-b| / max |b|  -- the tolerance definition used throughout."""
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - b)) / max(np.max(np.abs(b)), 1e-300))
