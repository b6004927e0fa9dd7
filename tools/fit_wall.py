#!/usr/bin/env python3
"""End-to-end wall time of ALS.fit on host triplets at config-2 size (what a user of the
estimator sees, PCIe and host work included).  Not the headline metric."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, 'recommendation-models_amd')):
    sys.path.insert(0, _p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from alsmi import ALS, synth  # noqa: E402
from alsmi import device as dev  # noqa: E402

device = dev.require_gpu()
n, m, nnz = 138493, 26744, 20000000
users, items, vals = synth.user_block(20242, 0, n, m, nnz, device)
users, items, vals = users.cpu().numpy(), items.cpu().numpy(), vals.double().cpu().numpy()
perm = np.random.default_rng(0).permutation(nnz)          # unsorted triplets, as a user would pass
users, items, vals = users[perm], items[perm], vals[perm]
for dtype in ('float32', 'float64'):
    for _ in range(2):
        als = ALS(rank=64, alpha=0.1, random_state=42, max_iter=10, dtype=dtype)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        als.fit((users, items), vals, shape=(n, m))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print('%s: fit of 10 sweeps %.2f s wall (second run), train rmse %.5f -> %.5f, score %.5f'
          % (dtype, dt, als.rmse_history_[0], als.rmse_history_[-1], als.reconstruction_err_))
