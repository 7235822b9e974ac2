"""MI355X-native Chebyshev graph convolution behind the deepsphere layer API.

Only the hot path of deepsphere/deepsphere-cosmo-tf2 lives here: ``gnn_layers.Chebyshev``,
``healpy_layers.HealpyChebyshev``, the Laplacian preparation in ``utils`` and a HEALPix graph
producer in ``healpix``.  The arithmetic runs in ``_lib/libdsphere_hip.so`` (hand-written HIP
for gfx950, C ABI in ``include/dsphere.h``).
"""

__version__ = "0.1.0"

from . import healpix, utils  # noqa: F401
from .gnn_layers import Chebyshev, GCNN_ResidualLayer, Monomial  # noqa: F401
from .healpy_layers import (HealpyChebyshev, HealpyMonomial, HealpyPool, HealpyPseudoConv,  # noqa: F401
                            HealpyPseudoConv_Transpose, Healpy_ResidualLayer)
from .healpy_networks import HealpyGCNN  # noqa: F401
