#!/usr/bin/env python3
"""cProfile of one ALS.fit on 20 M unsorted host triplets (second call; the first pays one-off
set-up), to see where the wall time of the estimator surface goes."""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, 'recommendation-models_amd')):
    sys.path.insert(0, _p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from alsmi import ALS, synth  # noqa: E402
from alsmi import device as dev  # noqa: E402

d = dev.require_gpu()
n, m, nnz = 138493, 26744, 20000000
u, i, v = synth.user_block(20242, 0, n, m, nnz, d)
u, i, v = u.cpu().numpy(), i.cpu().numpy(), v.double().cpu().numpy()
p = np.random.default_rng(0).permutation(nnz)
u, i, v = u[p], i[p], v[p]
ALS(rank=64, alpha=0.1, random_state=42, max_iter=10, dtype='float32').fit((u, i), v, shape=(n, m))
pr = cProfile.Profile()
pr.enable()
ALS(rank=64, alpha=0.1, random_state=42, max_iter=10, dtype='float32').fit((u, i), v, shape=(n, m))
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
