#!/usr/bin/env python3
"""Wall time of DirtyIMC.fit through the estimator surface: IMC with Newton-CG on the side
features, then ALS on the residual (src/dimc.py:116-162).  Host work and PCIe included; not the
headline metric.  Default: per-sample feature rows, the reference's input form.  --attribute
passes the user and item attribute tables with the index pairs instead (nothing of size
samples x p is ever formed); --config5-slice is one GPU's share of the 100 M-sample
configuration at 8 GPUs in that form (250 000 x 200 000, 12.5 M samples, p = 128, q = 256).

    python tools/bench_dimc.py [--nnz 2000000] [--attribute] [--config5-slice]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, 'recommendation-models_amd')):
    sys.path.insert(0, _p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from alsmi import DirtyIMC, synth  # noqa: E402
from alsmi import device as dev  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--users', type=int, default=40000)
ap.add_argument('--items', type=int, default=8000)
ap.add_argument('--nnz', type=int, default=2000000)
ap.add_argument('--p', type=int, default=32)
ap.add_argument('--q', type=int, default=64)
ap.add_argument('--rank', type=int, default=32)
ap.add_argument('--attribute', action='store_true')
ap.add_argument('--config5-slice', action='store_true')
ap.add_argument('--imc-alpha', type=float, default=None)
args = ap.parse_args()
if args.config5_slice:
    args.users, args.items, args.nnz, args.p, args.q, args.rank = 250000, 200000, 12500000, 128, 256, 8
    args.attribute = True
    if args.imc_alpha is None:
        args.imc_alpha = 1e-9  # (the reference's cost counts the regulariser once per sample)
device = dev.require_gpu()
n, m, nnz = args.users, args.items, args.nnz
users, items, vals = synth.user_block(20244, 0, n, m, nnz, device)
users, items = users.cpu().numpy(), items.cpu().numpy()
vals = vals.double().cpu().numpy()
rng = np.random.default_rng(1)
XA = rng.standard_normal((n, args.p)) / np.sqrt(args.p)   # noisy side features
YA = rng.standard_normal((m, args.q)) / np.sqrt(args.q)
x, y_ = (XA, YA) if args.attribute else (XA[users], YA[items])  # tables, or a row per sample
out = {}
for rep in range(2):
    kw = {} if args.imc_alpha is None else {'imc_alpha': args.imc_alpha}
    est = DirtyIMC(rank=args.rank, tol=0.001, random_state=7, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    est.fit((x, y_, users, items), vals, shape=(n, m))
    torch.cuda.synchronize()
    out['fit_s'] = time.perf_counter() - t0
out.update(workload='DirtyIMC n=%d m=%d nnz=%d p=%d q=%d rank=%d, %s'
           % (n, m, nnz, args.p, args.q, args.rank,
              'attribute tables + index pairs' if args.attribute else 'per-sample dense features'),
           als_sweeps=int(est.als.n_iter_), score=float(est.reconstruction_err_),
           imc_rmse=float(est.imc.reconstruction_err_))
print(json.dumps(out))
