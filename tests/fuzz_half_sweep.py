#!/usr/bin/env python3
"""Randomised parity run (by hand, not collected): many random shapes, ranks, chunk lengths,
scales and ridge values through alsmi_half_sweep / alsmi_half_sweep_err against the C oracle.

    python tests/fuzz_half_sweep.py [n_cases] [seed]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'recommendation-models_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np
import torch
from alsmi import device as dev
from oracle import c_oracle as co
from problems import random_problem

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
device = dev.require_gpu()
worst = {}
bad = 0
only = int(os.environ['FUZZ_ONLY']) if os.environ.get('FUZZ_ONLY') else None
for case in range(n_cases):
    rank = int(rng.choice([1, 2, 7, 10, 16, 17, 32, 33, 50, 64, 65, 100, 128]))
    n = int(rng.integers(5, 400))
    m = int(rng.integers(20, 600))
    dens = rng.choice([0.02, 0.1, 0.3, 0.7])
    nnz = max(1, int(n * m * dens))
    chunk = int(rng.choice([0, 0, 64, 128]))
    alpha = float(rng.choice([0.1, 0.01, 1.0, 0.003]))
    scale = float(rng.choice([1.0, 1e-3, 1e3, 1.0, 1.0]))
    dtype = str(rng.choice(['float32', 'float32', 'float64']))
    extra = {}
    if rng.random() < 0.4:
        extra['long_row'] = (int(rng.integers(0, n)), int(min(m, rng.integers(65, 600))))
    if rng.random() < 0.3:
        extra['empty_rows'] = tuple(int(x) for x in rng.integers(0, n, 3))
    P = random_problem(int(rng.integers(1 << 30)), n, m, nnz, rank, **extra)
    if only is not None and case != only:
        continue
    V = P['V'] * scale
    side = dev.SparseSide(P['indptr'], P['indices'], P['vals'], n, m, dtype, device)
    plan = dev.SweepPlan(side, rank, chunk=chunk)
    Vd = dev.factors_to_device(V, dtype, device)
    Vq = dev.factors_to_host(Vd, rank)
    kp = dev.padded_rank(rank)
    U = torch.full((n, kp), 3.0, dtype=Vd.dtype, device=device)
    plan.half_sweep(Vd, U, alpha)
    got = dev.factors_to_host(U, rank)                      # (rank, n)
    want = co.half_sweep(P['indptr'], P['indices'], P['vals'], np.ascontiguousarray(Vq.T), alpha).T
    # Forward error against the oracle where the system is well conditioned; everywhere the
    # normwise backward error ||A x - b|| / (||A|| ||x|| + ||b||) of the row's own normal
    # equations (a solver is not to blame for what an ill-conditioned matrix does to the
    # forward error), zero rows excepted where the matrix is numerically singular.
    eps = 2.0 ** -52 if dtype == 'float64' else 2.0 ** -24
    fwd_worst = bwd_worst = 0.0
    for r in range(n):
        lo, hi = P['indptr'][r], P['indptr'][r + 1]
        if hi == lo:
            assert not got[:, r].any()
            continue
        S = Vq[:, P['indices'][lo:hi]]
        A = S.dot(S.T) + alpha * (hi - lo) * np.eye(rank)
        b = S.dot(P['vals'][lo:hi])
        cond = np.linalg.cond(A)
        x = got[:, r]
        if not x.any() and cond * eps > 0.05:
            continue  # reported singular, and numerically it is
        if only is not None and not x.any():
            print('   row %d: %d entries, cond %.3g, returned as a zero row' % (r, hi - lo, cond))
        bwd = np.abs(A.dot(x) - b).max() / (np.abs(A).sum(1).max() * np.abs(x).max() + np.abs(b).max())
        bwd_worst = max(bwd_worst, bwd / eps)
        if cond * eps < 1e-3:
            # (the right-hand side b = sum r f is itself a sum with cancellation)
            kb = np.abs(S).dot(np.abs(P['vals'][lo:hi])).max() / max(np.abs(b).max(), 1e-300)
            fwd = np.linalg.norm(x - want[:, r]) / max(np.linalg.norm(want[:, r]), 1e-300)
            fwd_worst = max(fwd_worst, fwd / (cond * max(kb, 1.0) * eps))
    # the fused error of the vectors just written, against the direct sum
    U2 = torch.zeros_like(U)
    sse = float(plan.half_sweep_err(Vd, U, U2, alpha).cpu()[0])
    pred = (got.T[P['users']] * Vq.T[P['items']]).sum(-1)
    want_sse = float(((P['vals'] - pred) ** 2).sum())
    sum_r2 = float((P['vals'] ** 2).sum())
    # (the fused form cancels r^2 - 2 u.b + u^T G u: its error is relative to those terms)
    big = sum_r2 + float((pred ** 2).sum())
    sse_err = abs(sse - want_sse) / max(big, 1e-300) / eps
    close = float((U - U2).abs().max() / max(float(U.abs().max()), 1e-30)) / eps
    key = (dtype, kp)
    w = worst.setdefault(key, [0.0, 0.0, 0.0, 0.0])
    w[0], w[1], w[2], w[3] = max(w[0], bwd_worst), max(w[1], fwd_worst), max(w[2], sse_err), max(w[3], close)
    # (the backward error and the difference between the two entry points are reported only: the n x n form of short rows and the fp64
    # Gauss-Jordan path of narrow ranks are forward stable, which is the parity criterion)
    if fwd_worst > 100 or sse_err > 2000 or not np.isfinite(got).all():
        bad += 1
        print('CASE %d FAILED: rank %d n %d m %d nnz %d chunk %d alpha %g scale %g %s extra %s: '
              'backward %.3g eps, forward %.3g cond eps, sse %.3g eps, err-vs-plain %.3g eps'
              % (case, rank, n, m, nnz, chunk, alpha, scale, dtype, extra, bwd_worst, fwd_worst,
                 sse_err, close), flush=True)
    plan.close()
for k in sorted(worst):
    print('%s padded rank %3d: worst backward error %.3g eps, forward %.3g cond*eps, fused sse '
          '%.3g eps, error variant vs plain %.3g eps' % ((k[0], k[1]) + tuple(worst[k])))
print('%d cases, %d failed' % (n_cases, bad))
sys.exit(1 if bad else 0)
