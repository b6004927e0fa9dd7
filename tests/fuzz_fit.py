#!/usr/bin/env python3
"""Randomised parity run (by hand, not collected) of whole fits through the estimator surface:
ALS.fit_transform (src/als.py:109-171: preamble, the `while diff > tol` loop, the error of every
sweep) against oracle.als_oracle.als_fit on random shapes, ranks, ridge values and tolerances,
with duplicate and zero ratings, empty users and items; predict_one / score afterwards.

    python tests/fuzz_fit.py [n_cases] [seed]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'recommendation-models_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np
from alsmi import ALS, device as dev
from oracle import als_oracle as ao

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev.require_gpu()
bad = 0
worst = dict(factors=0.0, rmse=0.0, pred=0.0)
for case in range(n_cases):
    n = int(rng.integers(3, 200))
    m = int(rng.integers(3, 200))
    rank = int(rng.choice([1, 2, 5, 10, 16, 20, 33, 64, 70]))
    alpha = float(rng.choice([0.1, 0.05, 1.0]))
    dens = float(rng.choice([0.05, 0.2, 0.6]))
    n_in = max(4, int(n * m * dens))
    users = rng.integers(0, n, n_in)                       # duplicates on purpose
    items = rng.integers(0, m, n_in)
    y = rng.integers(0, 6, n_in).astype(np.float64)         # zeros are dropped by the preamble
    if rng.random() < 0.5:
        users[users == users[0]] = users[-1]                # a user without ratings
    seed = int(rng.integers(1 << 20))
    mode = str(rng.choice(['tol', 'tol_small', 'max_iter']))
    kw = dict(tol=0.001)
    okw = dict(tol=0.001)
    if mode == 'tol_small':
        kw = dict(tol=1e-4)
        okw = dict(tol=1e-4)
    elif mode == 'max_iter':
        it = int(rng.integers(1, 5))
        kw = dict(tol=0.0, max_iter=it)
        okw = dict(max_sweeps=it)
    try:
        want_u, want_i, want_hist, csr = ao.als_fit(users, items, y, (n, m), rank=rank,
                                                    alpha=alpha, seed=seed, **okw)
    except Exception as exc:                                # e.g. nothing left after the zeros
        continue
    als = ALS(rank=rank, alpha=alpha, random_state=seed, dtype='float64', **kw)
    got_u, got_i = als.fit_transform((users, items), y, shape=(n, m))
    msg = None
    if als.n_iter_ != len(want_hist):
        # a stop decided by a difference at rounding level may legitimately differ by a sweep
        diffs = np.abs(np.diff(np.concatenate([[np.inf], want_hist])))
        if not (mode != 'max_iter' and np.any(np.abs(diffs - kw['tol']) < 1e-8)):
            msg = 'sweeps %d vs %d' % (als.n_iter_, len(want_hist))
    else:
        scale = max(np.abs(want_u).max(), np.abs(want_i).max())
        e_f = max(np.abs(got_u - want_u).max(), np.abs(got_i - want_i).max()) / scale
        e_r = float(np.abs(np.asarray(als.rmse_history_) - want_hist).max() /
                    max(want_hist.max(), 1e-300))
        # ten and more sweeps amplify the rounding differences of the first ones
        tol_f = 1e-7 if len(want_hist) <= 6 else 1e-5
        pu, pi = rng.integers(0, n, 5), rng.integers(0, m, 5)
        e_p = max(abs(float(np.ravel(als.predict_one(int(a), int(b)))[0]) - want_u[:, a].dot(want_i[:, b]))
                  for a, b in zip(pu, pi)) / max(scale * scale, 1e-300)
        worst['factors'] = max(worst['factors'], e_f)
        worst['rmse'] = max(worst['rmse'], e_r)
        worst['pred'] = max(worst['pred'], e_p)
        if not (e_f <= tol_f and e_r <= 1e-9 and e_p <= tol_f):
            msg = 'factors %.2e rmse %.2e pred %.2e after %d sweeps' % (e_f, e_r, e_p,
                                                                        len(want_hist))
    if msg:
        bad += 1
        print('case %d n=%d m=%d rank=%d alpha=%g mode=%s n_in=%d: %s' % (
            case, n, m, rank, alpha, mode, n_in, msg), flush=True)
    if case % 25 == 24:
        print('... %d cases, %d failed' % (case + 1, bad), flush=True)
print('%d cases, %d failed; worst factors %.2e, rmse history %.2e, predictions %.2e' % (
    n_cases, bad, worst['factors'], worst['rmse'], worst['pred']))
sys.exit(1 if bad else 0)
