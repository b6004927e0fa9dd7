#!/usr/bin/env python3
"""Wall time of DirtyIMC.fit through the estimator surface (per-sample feature rows, the
reference's input form, src/dimc.py:116-162): IMC with Newton-CG on the side features, then ALS
on the residual.  Host work and PCIe included; not the headline metric.

    python tools/bench_dimc.py [--nnz 2000000]
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
args = ap.parse_args()
device = dev.require_gpu()
n, m, nnz = args.users, args.items, args.nnz
users, items, vals = synth.user_block(20244, 0, n, m, nnz, device)
users, items = users.cpu().numpy(), items.cpu().numpy()
vals = vals.double().cpu().numpy()
rng = np.random.default_rng(1)
XA = rng.standard_normal((n, args.p)) / np.sqrt(args.p)   # noisy side features
YA = rng.standard_normal((m, args.q)) / np.sqrt(args.q)
x, y_ = XA[users], YA[items]                               # one dense row per sample
out = {}
for rep in range(2):
    est = DirtyIMC(rank=args.rank, tol=0.001, random_state=7)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    est.fit((x, y_, users, items), vals, shape=(n, m))
    torch.cuda.synchronize()
    out['fit_s'] = time.perf_counter() - t0
out.update(workload='DirtyIMC n=%d m=%d nnz=%d p=%d q=%d rank=%d, per-sample dense features'
           % (n, m, nnz, args.p, args.q, args.rank),
           als_sweeps=int(est.als.n_iter_), score=float(est.reconstruction_err_),
           imc_score=float(est.imc.score((x, y_), vals)) if hasattr(est.imc, 'score') else None)
print(json.dumps(out))
