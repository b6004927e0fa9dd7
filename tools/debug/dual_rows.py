"""Debug aid: per-row error of the rank-100 (padded 128) fp32 half-sweep against the numpy oracle."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'recommendation-models_amd'), os.path.join(ROOT, 'tests')]
import torch
from problems import random_problem
from alsmi import device as dev
from oracle import als_oracle as ao
device = dev.require_gpu()
seed, n, m, nnz, rank, chunk, extra = (5, 30, 200, 1500, 100, 64, dict(long_row=(3, 170)))
P = random_problem(seed, n, m, nnz, rank, **extra)
side = dev.SparseSide(P['indptr'], P['indices'], P['vals'], n, m, 'float32', device)
plan = dev.SweepPlan(side, rank, chunk=chunk)
V = dev.factors_to_device(P['V'], 'float32', device)
Vq = dev.factors_to_host(V, rank)
U = torch.full((n, 128), 7.0, dtype=V.dtype, device=device)
plan.half_sweep(V, U, 0.1)
got = dev.factors_to_host(U, rank)
want = ao.half_sweep(P['indptr'], P['indices'], P['vals'], Vq, rank, 0.1)
deg = np.diff(P['indptr'])
np.set_printoptions(linewidth=200, precision=3, suppress=True)
for r in range(n):
    e = np.linalg.norm(got[:, r] - want[:, r]) / max(np.linalg.norm(want[:, r]), 1e-30)
    print(r, deg[r], '%.2e' % e)
r = int(np.argmax([np.linalg.norm(got[:, r] - want[:, r]) for r in range(n)]))
print('worst row', r, 'deg', deg[r])
print('got ', got[:40, r])
print('want', want[:40, r])
full = U[r].cpu().numpy()
# where does each wanted value appear in the output row?
for d in range(16):
    j = np.argmin(np.abs(full - want[d, r]))
    print(d, '->', j, full[j], want[d, r])
