#!/usr/bin/env python3
"""Rows and stored entries per kernel class of a bench configuration's two sides (classes of 16 up to 96
entries: the short-row kernels; longer rows up to one chunk; chunked rows):
    python tools/debug/class_counts.py [config]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'recommendation-models_amd')):
    sys.path.insert(0, p)
import numpy as np
import bench
from alsmi import synth, device as dev

cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'c2']
device = dev.require_gpu()
prob = synth.sharded_problem(20242, cfg['n'], cfg['m'], cfg['nnz'], 1, 0, device)
kp = dev.padded_rank(cfg['rank'])
for name, side in (('user', prob['csr']), ('item', prob['csc'])):
    deg = np.diff(side[0].cpu().numpy())
    print('%s side: %d rows, %d entries, padded rank %d' % (name, deg.size, deg.sum(), kp))
    for lo, hi in ((0, 0), (1, 16), (17, 32), (33, 48), (49, 64), (65, 80), (81, 96), (97, 128),
                   (129, 256), (257, 2048), (2049, 1 << 40)):
        sel = (deg >= lo) & (deg <= hi)
        ent = int(deg[sel].sum())
        print('   %5d..%-7s rows %9d  entries %11d  gathered %7.2f GB' % (
            lo, hi if hi < (1 << 40) else 'max', sel.sum(), ent, ent * kp * 4 / 1e9))
