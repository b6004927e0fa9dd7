#!/usr/bin/env python3
"""Test aid (not collected by pytest): measured accuracy of the fp32 half-sweep against the fp64
oracle, run by hand on the GPU box:
worst and median relative error per factor vector for a few seeded problems."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'recommendation-models_amd'), os.path.join(ROOT, 'tests')):  # noqa
    sys.path.insert(0, p)

from problems import random_problem  # noqa: E402
from alsmi import device as dev  # noqa: E402
from oracle import c_oracle as co  # noqa: E402

device = dev.require_gpu()
CASES = [(11, 300, 200, 6000, 10, 0), (12, 400, 300, 30000, 32, 0), (13, 2000, 1500, 150000, 64, 0),
         (14, 500, 3000, 200000, 64, 256), (15, 3000, 800, 60000, 50, 0),
         (16, 1500, 1200, 150000, 128, 0), (17, 600, 2500, 200000, 100, 256)]
for seed, n, m, nnz, rank, chunk in CASES:
    P = random_problem(seed, n, m, nnz, rank)
    for alpha in (0.1, 0.001):
        side = dev.SparseSide(P['indptr'], P['indices'], P['vals'], n, m, 'float32', device)
        plan = dev.SweepPlan(side, rank, chunk=chunk)
        V = dev.factors_to_device(P['V'], 'float32', device)
        U = V.new_zeros((n, dev.padded_rank(rank)))
        plan.half_sweep(V, U, alpha)
        got = dev.factors_to_host(U, rank)
        want = co.half_sweep(P['indptr'], P['indices'], P['vals'].astype(np.float64),
                             np.ascontiguousarray(dev.factors_to_host(V, rank).T), alpha,
                             threads=8).T
        num = np.linalg.norm(got - want, axis=0)
        den = np.maximum(np.linalg.norm(want, axis=0), 1e-30)
        rel = num / den
        print('rank %3d n=%5d nnz=%7d alpha=%-6g  worst %.2e  median %.2e' %
              (rank, n, nnz, alpha, rel.max(), np.median(rel)))
        plan.close()
