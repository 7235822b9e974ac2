"""Model assembly on HEALPix maps: the caller of the Chebyshev path.

Mirror of the reference's ``deepsphere.healpy_networks.HealpyGCNN`` (``healpy_networks.py:14-167``) as a
``torch.nn.Sequential``: it walks a list of layers, builds the graph Laplacian of the current resolution for
every graph layer (the reference asks the pygsp fork for ``SphereHealpix(subdivisions=nside, indexes=indices,
nest=True, k=n_neighbors, lap_type="normalized").L`` at ``:110-118``; here ``healpix.healpix_laplacian`` produces
the same family of matrices), hands it to the layer spec's ``_get_layer`` and follows the pixel set through
pooling layers.  Filter plotting (``:190-385``) is not rebuilt.
"""

import numpy as np
import torch

from . import gnn_layers as gnn
from . import healpix
from . import healpy_layers as hp_nn


class HealpyGCNN(torch.nn.Sequential):
    """A graph convolutional network on a (partial) HEALPix map in NEST ordering."""

    def __init__(self, nside, indices, layers, n_neighbors=8, max_batch_size=None, initial_Fin=None,
                 graph_mode="knn"):
        """
        :param nside: nside of the input maps
        :param indices: sorted NEST pixel ids of the input maps
        :param layers: list of layers / layer specs (``HealpyChebyshev`` ...)
        :param n_neighbors: neighbours of the graph, 8 (default), 20, 40 or 60
        :param max_batch_size, initial_Fin: the reference derives ``n_matmul_splits`` for TensorFlow's sparse
            matmul from them (``:125-134``); the same number is computed and passed on, the HIP kernels ignore it
        :param graph_mode: "knn" (what the reference builds) or "grid" (fixed 8-neighbour stencil) -- not a
            reference argument
        """
        if n_neighbors not in [8, 20, 40, 60]:
            raise NotImplementedError(
                f"The requested number of neighbors {n_neighbors} is nor supported. Choose either 8, 20, 40 or 60.")
        indices = np.asarray(indices)
        nside_in = int(nside)
        reduction_fac = 1.0
        for layer in layers:
            if isinstance(layer, (hp_nn.HealpyPool, hp_nn.HealpyPseudoConv)):
                reduction_fac *= 2 ** layer.p
            if isinstance(layer, hp_nn.HealpyPseudoConv_Transpose):
                reduction_fac /= 2 ** layer.p
        nside_out = int(nside_in // reduction_fac)
        if nside_out < 1:
            raise ValueError("With the given input, the layers would reduce the nside below zero!"
                             "Use less layers that reduce the nside, e.g. HealpyPool or HealpyPseudoConv...")
        if not healpix.isnsideok(nside_out):
            raise ValueError(f"The ouput of the network does not have a valid nside {nside_out}...")
        # the pixel set must be closed under the coarsening the layers perform (:73-88)
        if nside_out < nside_in:
            closed = healpix.extend_indices(indices, nside_in, nside_out)
            if not np.array_equal(np.sort(closed.astype(int)), np.sort(indices.astype(int))):
                raise ValueError("With the given indices it would not be possible to properly reduce the input maps "
                                 "with the reduction factor determined by the layers. Use the function "
                                 "<extend_indices> from utils with the determined minimal nside to make your set of "
                                 "indices compatible...")
        layers_use = []
        current_nside, current_indices, current_Fin = nside_in, indices, initial_Fin
        for layer in layers:
            if isinstance(layer, (hp_nn.HealpyChebyshev, hp_nn.HealpyMonomial, hp_nn.Healpy_ResidualLayer)):
                L = healpix.healpix_laplacian(current_nside, indices=current_indices, n_neighbors=n_neighbors,
                                              mode=graph_mode)
                if max_batch_size is not None and current_Fin is not None:
                    splits = 1
                    while not (max_batch_size * current_Fin % splits == 0
                               and splits >= max_batch_size * current_Fin * L.nnz / 2**31):
                        splits += 1
                    layers_use.append(layer._get_layer(L, splits))
                else:
                    layers_use.append(layer._get_layer(L))
            elif isinstance(layer, (hp_nn.HealpyPool, hp_nn.HealpyPseudoConv)):
                new_nside = int(current_nside // 2 ** layer.p)
                current_indices = self._transform_indices(current_nside, new_nside, current_indices)
                current_nside = new_nside
                layers_use.append(layer)
            elif isinstance(layer, hp_nn.HealpyPseudoConv_Transpose):
                new_nside = int(current_nside * 2 ** layer.p)
                current_indices = self._transform_indices(current_nside, new_nside, current_indices)
                current_nside = new_nside
                layers_use.append(layer)
            else:
                layers_use.append(layer)
            fout = getattr(layer, "Fout", None)
            if fout is not None:
                current_Fin = fout
        super().__init__(*layers_use)
        self.nside_in, self.nside_out = nside_in, nside_out
        self.indices_in, self.indices_out = indices, current_indices
        self.layers_in, self.layers_use = layers, layers_use
        self.n_neighbors = n_neighbors
        self.reduction_fac = reduction_fac

    @staticmethod
    def _transform_indices(nside_in, nside_out, indices):
        """Pixel ids of the same sky area at another nside (NEST): parents when coarsening, all children when
        refining (the reference does this with ``hp.ud_grade`` on a mask, ``:169-188``)."""
        indices = np.asarray(indices, dtype=np.int64)
        if nside_in == nside_out:
            return indices
        if nside_out < nside_in:
            per = (nside_in // nside_out) ** 2
            return np.unique(indices // per)
        per = (nside_out // nside_in) ** 2
        return (indices[:, None] * per + np.arange(per, dtype=np.int64)[None, :]).reshape(-1)

    def forward(self, input_tensor, training=False):
        """``training`` defaults to False like the reference's ``call`` (Keras passes True by itself inside ``fit``; torch has
        no counterpart).  A network in ``model.train()`` mode that holds batch-norm layers and is called without the argument
        would silently normalise with -- and never update -- the moving statistics: that combination warns once; write
        ``model(x, training=True)`` in a training loop (``training=None``: follow ``self.training``)."""
        if training is False and self.training and not getattr(self, "_warned_training", False):
            if any(getattr(layer, "use_bn", False) for layer in self):
                import warnings

                warnings.warn("HealpyGCNN is in train() mode but forward() was called with the default training=False: its "
                              "batch-norm layers use (and do not update) the moving statistics.  Pass training=True "
                              "(or training=None to follow module.training).", stacklevel=2)
                self._warned_training = True
        x = input_tensor
        layers = list(self)
        i = 0
        while i < len(layers):
            layer = layers[i]
            if isinstance(layer, (gnn.Chebyshev, gnn.GCNN_ResidualLayer)):
                # a graph convolution followed by HealpyPool(p = 1): one pass where the kernels can (inference, a first layer:
                # the strip kernel stores the pooled map and the full-resolution output is never written), else the two layers
                nxt = layers[i + 1] if i + 1 < len(layers) else None
                if (isinstance(layer, gnn.Chebyshev) and isinstance(nxt, hp_nn.HealpyPool) and nxt.p == 1 and not training
                        and isinstance(x, torch.Tensor)):
                    y = layer.forward_pool(x, nxt.pool_type)
                    if y is not None:
                        x = y
                        i += 2
                        continue
                x = layer(x, training=training)
            else:
                if not isinstance(x, torch.Tensor):
                    x = torch.as_tensor(np.asarray(x), dtype=torch.float32)
                x = layer(x)
            i += 1
        return x

    call = forward


__all__ = ["HealpyGCNN"]
