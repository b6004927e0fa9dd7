#!/usr/bin/env python3
"""Where does the fp32 normal-equation dump differ from numpy?  Prints, for the first bad row,
the tiles (16 x 16 blocks in the kernel's interleaved order) with their largest relative error.
    python tools/debug/normal_eq_diff.py [rank [chunk]]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'recommendation-models_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np
from alsmi import device as dev
from problems import random_problem
rank = int(sys.argv[1]) if len(sys.argv) > 1 else 128
device = dev.require_gpu()
n, m, nnz = 30, 200, 1500
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 0
P = random_problem(5, n, m, nnz, rank, long_row=(3, 170)) if chunk else random_problem(5, n, m, nnz, rank)
side = dev.SparseSide(P['indptr'], P['indices'], P['vals'], n, m, 'float32', device)
plan = dev.SweepPlan(side, rank, chunk=chunk)
V = dev.factors_to_device(P['V'], 'float32', device)
Vq = dev.factors_to_host(V, rank)
kp = dev.padded_rank(rank)
A, b = plan.normal_equations(V, 0.1)
A = A.cpu().numpy().astype(np.float64)
nv = kp // 16
for r in range(n):
    lo, hi = P['indptr'][r], P['indptr'][r + 1]
    if hi == lo:
        continue
    S = Vq[:, P['indices'][lo:hi]]
    Aref = np.zeros((kp, kp)); Aref[:rank, :rank] = S.dot(S.T) + 0.1 * (hi - lo) * np.eye(rank); Aref[rank:, rank:] = np.eye(kp - rank)
    err = np.abs(A[r] - Aref).max() / np.abs(Aref).max()
    print('row %d entries %d rel err %.3g' % (r, hi - lo, err))
    if err > 2e-6:
        D = np.abs(A[r] - Aref) / np.abs(Aref).max()
        T = np.zeros((nv, nv))
        for e in range(nv):
            for f in range(nv):
                T[e, f] = D[e::nv, f::nv].max()
        np.set_printoptions(linewidth=200, precision=2)
        print(T)
        i, j = np.unravel_index(np.argmax(D), D.shape)
        print('worst', i, j, A[r][i, j], Aref[i, j], 'ratio', A[r][i, j] / Aref[i, j])
        break
