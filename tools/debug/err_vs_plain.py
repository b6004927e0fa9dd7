#!/usr/bin/env python3
"""Do alsmi_half_sweep and alsmi_half_sweep_err write bit-identical rows, and is each of them
bit-reproducible?  Prints the rows that differ with their lengths.
    python tools/debug/err_vs_plain.py [rank] [n] [m] [nnz] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'recommendation-models_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np
import torch
from alsmi import device as dev
from problems import random_problem
rank, n, m, nnz, seed = (int(a) for a in (sys.argv[1:] + ['100', '171', '527', '9011', '3'][len(sys.argv) - 1:]))
device = dev.require_gpu()
P = random_problem(seed, n, m, nnz, rank, long_row=(min(94, n - 1), min(285, m)))
side = dev.SparseSide(P['indptr'], P['indices'], P['vals'], n, m, 'float32', device)
plan = dev.SweepPlan(side, rank)
Vd = dev.factors_to_device(P['V'], 'float32', device)
kp = dev.padded_rank(rank)
deg = np.diff(P['indptr'])
def plain():
    U = torch.full((n, kp), 3.0, dtype=Vd.dtype, device=device); plan.half_sweep(Vd, U, 0.01); return U
def err():
    cur = torch.full((n, kp), 0.5, dtype=Vd.dtype, device=device)
    U = torch.zeros_like(cur); plan.half_sweep_err(Vd, cur, U, 0.01); return U
a, b, c, d = plain(), plain(), err(), err()
for name, x, y in (('plain vs plain', a, b), ('err vs err', c, d), ('plain vs err', a, c)):
    rows = torch.nonzero((x != y).any(1)).flatten().cpu().numpy()
    print(name, 'differing rows:', len(rows), 'lengths', sorted(set(deg[rows].tolist()))[:20],
          'max rel diff %.3g' % (float(((x - y).abs().max() / x.abs().max())) if len(rows) else 0.0))
