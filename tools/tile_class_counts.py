"""Diagnostic (GPU box): tiles by class of BASELINE configs[4] (partial sky) -- how many the strips take, how many stay on the
table-driven structured kernel and why.  python3 tools/tile_class_counts.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'deepsphere-cosmo-tf2_amd'))
import torch, bench
from deepsphere import _native
cols, vals, _ = bench.build_laplacian_masked(1024, torch.device('cuda', 0))
K, Fin, Fout, N = 5, 64, 64, 16
p = _native.LaplacianPlan(cols, vals, device=0); p.prepare(K, Fin)
print('default: struct, bfs =', p.tile_counts(K), 'strip tiles', p.strip_tiles(Fin, Fout, K, _native.PREC_BF16X3, N=N), 'tiles total', (cols.shape[0] + 255) // 256)
q = _native.LaplacianPlan(cols, vals, device=0, options={_native.OPT_TABLES: 0}); q.prepare(K, Fin)
print('no tables: struct (= class R), bfs =', q.tile_counts(K), 'strip tiles', q.strip_tiles(Fin, Fout, K, _native.PREC_BF16X3, N=N))
pairs = p.strip_pairs(K)
import numpy as np
w = pairs[:, 2]; h = pairs[:, 7] - pairs[:, 6]
print('quad strips:', len(pairs), 'widths histogram', np.bincount(w // 8)[:9], 'heights min/median/max', h.min(), int(np.median(h)), h.max())
