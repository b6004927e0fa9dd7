#!/usr/bin/env python3
"""Randomised parity run (by hand, not collected): random shapes, attribute widths, sample
patterns and dtypes through the IMC evaluations (cost / grad / hessp / predictions / the low-rank
passes, src/imc.py:43-96) against the numpy oracle.

    python tests/fuzz_imc.py [n_cases] [seed]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'recommendation-models_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np
from alsmi import device as dev
from alsmi.imc import ImcProblem
from oracle import imc_oracle as io

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev.require_gpu()
only = int(os.environ['FUZZ_ONLY']) if os.environ.get('FUZZ_ONLY') else None
bad = 0
worst = {}


def rel(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    return float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-300))


for case in range(n_cases):
    n = int(rng.integers(1, 300))
    m = int(rng.integers(1, 300))
    p = int(rng.choice([1, 2, 3, 15, 16, 17, 31, 33, 64, 65, 100, 129, 200, 256]))
    q = int(rng.choice([1, 2, 3, 15, 16, 17, 31, 33, 64, 65, 100, 129, 200, 256]))
    dens = float(rng.choice([0.005, 0.05, 0.3, 0.9]))
    nnz = max(1, min(n * m, int(n * m * dens)))
    dtype = str(rng.choice(['float64', 'float64', 'float32']))
    lam = float(rng.choice([0.0, 0.01, 0.3]))
    r = int(rng.choice([1, 4, 10, 16, 32, 33]))
    flat = rng.choice(n * m, size=nnz, replace=False)
    users, items = flat // m, flat % m
    if rng.random() < 0.3 and n > 3:                          # some users without samples
        keep = ~np.isin(users, rng.integers(0, n, 3))
        if keep.any():
            users, items = users[keep], items[keep]
    if rng.random() < 0.3:                                    # one user rated everything
        u = int(rng.integers(0, n))
        keep = users != u
        users = np.concatenate([users[keep], np.full(m, u)])
        items = np.concatenate([items[keep], np.arange(m)])
    perm = rng.permutation(users.size)                        # caller order is arbitrary
    users, items = users[perm], items[perm]
    vals = rng.standard_normal(users.size) * float(rng.choice([1.0, 5.0, 1e-2]))
    X = rng.standard_normal((n, p)) / np.sqrt(p)
    Y = rng.standard_normal((m, q)) / np.sqrt(q)
    Z, S = rng.standard_normal((p, q)), rng.standard_normal((p, q))
    U, V = rng.standard_normal((p, r)), rng.standard_normal((q, r))
    if only is not None and case != only:
        continue
    if dtype == 'float32':                                    # what the device holds
        Xq, Yq = X.astype(np.float32).astype(np.float64), Y.astype(np.float32).astype(np.float64)
        tol = 2e-4
    else:
        Xq, Yq, tol = X, Y, 1e-10
    prob = ImcProblem(X, Y, users, items, vals, dtype=dtype)
    errs = {
        'cost': rel(prob.cost(Z, lam), io.cost_indexed(Z, Xq, Yq, users, items, vals, lam)),
        'grad': rel(prob.grad(Z, lam), io.grad_indexed(Z, Xq, Yq, users, items, vals, lam)),
        'hessp': rel(prob.hessp(S, lam), io.hessp_indexed(S, Xq, Yq, users, items, lam)),
        'pred': rel(prob.predict_samples(Z), io.preds_indexed(Z, Xq, Yq, users, items)),
    }
    sse, gU, gV = prob.low_rank_eval(U, V)
    W = U.dot(V.T)
    G = io.grad_indexed(W, Xq, Yq, users, items, vals, 0.0)
    errs['lr_sse'] = rel(sse, 2.0 * io.cost_indexed(W, Xq, Yq, users, items, vals, 0.0))
    errs['lr_gU'] = rel(gU, G.dot(V))
    errs['lr_gV'] = rel(gV, G.T.dot(U))
    # a second evaluation on the same problem (workspaces are kept across calls)
    errs['grad2'] = rel(prob.grad(S, lam), io.grad_indexed(S, Xq, Yq, users, items, vals, lam))
    for k, e in errs.items():
        key = (dtype, k)
        worst[key] = max(worst.get(key, 0.0), e)
    fails = {k: e for k, e in errs.items() if not e <= tol}
    if fails or only is not None:
        bad += bool(fails)
        print('case %d n=%d m=%d p=%d q=%d r=%d nnz=%d %s lam=%g: %s' % (
            case, n, m, p, q, r, users.size, dtype, lam,
            ' '.join('%s=%.2e' % kv for kv in (fails or errs).items())), flush=True)
    if case % 50 == 49:
        print('... %d cases, %d failed' % (case + 1, bad), flush=True)
print('%d cases, %d failed' % (n_cases, bad))
for key in sorted(worst):
    print('  worst %-8s %-7s %.2e' % (key[0], key[1], worst[key]))
sys.exit(1 if bad else 0)
