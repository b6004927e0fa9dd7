#!/usr/bin/env python3
"""Randomised parity run (by hand, not collected) of the entry points around the half-sweep:
alsmi_build_csr (the reference's lil fill, src/als.py:137-140), alsmi_csr_to_csc (scipy's
tocsc, src/als.py:275-279), alsmi_sddmm (predictions and squared error, src/als.py:161-165,
347-364) and alsmi_topk_items, each against numpy / scipy / the oracle on random shapes.

    python tests/fuzz_aux.py [n_cases] [seed]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'recommendation-models_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np
import scipy.sparse as sps
import torch
from alsmi import device as dev
from oracle import als_oracle as ao

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
gpu = dev.require_gpu()
bad = 0
worst = {}


def fail(case, what, detail):
    global bad
    bad += 1
    print('case %d %s: %s' % (case, what, detail), flush=True)


for case in range(n_cases):
    n = int(rng.integers(1, 400))
    m = int(rng.integers(1, 400))
    dtype = str(rng.choice(['float64', 'float32']))
    td = torch.float64 if dtype == 'float64' else torch.float32
    eps = 2.0 ** -52 if dtype == 'float64' else 2.0 ** -23
    # ---- triplets -> CSR: duplicates, zeros, negative indices, empty input ----------------
    n_in = int(rng.choice([0, 1, 5, 100, 3000, 40000]))
    users = rng.integers(-n, n, n_in)
    items = rng.integers(-m, m, n_in)
    y = rng.integers(0, 5, n_in).astype(np.float64) * 0.5
    want = ao.build_csr(np.where(users < 0, users + n, users), np.where(items < 0, items + m, items),
                        y, (n, m))
    got = dev.build_csr(users, items, y, (n, m), dtype, gpu)
    if not (np.array_equal(got[0].cpu().numpy(), want[0]) and
            np.array_equal(got[1].cpu().numpy(), want[1]) and
            np.array_equal(got[2].cpu().numpy(), want[2].astype(dtype))):
        fail(case, 'build_csr', 'n=%d m=%d n_in=%d %s' % (n, m, n_in, dtype))
        continue
    # ---- CSR -> CSC -----------------------------------------------------------------------
    indptr, indices, vals = got
    nnz = int(indices.numel())
    csr = sps.csr_matrix((want[2], want[1], want[0]), shape=(n, m))
    csc = csr.tocsc()
    if nnz:
        cptr, cidx, cvals = dev.csr_to_csc(indptr, indices, vals, (n, m))
        if not (np.array_equal(cptr.cpu().numpy(), csc.indptr) and
                np.array_equal(cidx.cpu().numpy(), csc.indices) and
                np.array_equal(cvals.cpu().numpy(), csc.data.astype(dtype))):
            fail(case, 'csr_to_csc', 'n=%d m=%d nnz=%d %s' % (n, m, nnz, dtype))
            continue
    # ---- SDDMM over the stored entries and over random pairs --------------------------------
    rank = int(rng.choice([1, 2, 7, 10, 16, 17, 32, 50, 64, 100, 128]))
    kp = dev.padded_rank(rank)
    U = rng.standard_normal((rank, n)) * float(rng.choice([1.0, 1e-2, 30.0]))
    V = rng.standard_normal((rank, m))
    Ud, Vd = dev.factors_to_device(U, dtype, gpu), dev.factors_to_device(V, dtype, gpu)
    Uq, Vq = dev.factors_to_host(Ud, rank), dev.factors_to_host(Vd, rank)
    n_pairs = int(rng.choice([1, 3, 63, 64, 65, 1000, 20000]))
    pu, pi = rng.integers(0, n, n_pairs), rng.integers(0, m, n_pairs)
    pv = rng.standard_normal(n_pairs)
    to = lambda a, t: torch.from_numpy(np.ascontiguousarray(a)).to(gpu, t)
    pred, sse = dev.sddmm(rank, to(pu, torch.int32), to(pi, torch.int32), Ud, Vd, vals=to(pv, td))
    want_pred = np.einsum('kp,kp->p', Uq[:, pu], Vq[:, pi])
    pvq = pv.astype(dtype).astype(np.float64)
    scale = np.abs(Uq[:, pu] * Vq[:, pi]).sum(axis=0).max() + 1e-300
    e_pred = float(np.abs(pred.cpu().numpy().astype(np.float64) - want_pred).max() / scale / eps)
    want_sse = float(((pvq - want_pred) ** 2).sum())
    e_sse = abs(float(sse.cpu()[0]) - want_sse) / max(want_sse, 1e-300) / eps
    worst[(dtype, 'pred')] = max(worst.get((dtype, 'pred'), 0.0), e_pred)
    worst[(dtype, 'sse')] = max(worst.get((dtype, 'sse'), 0.0), e_sse)
    if not (e_pred <= 4 * np.sqrt(rank) + 4 and e_sse <= 64 * np.sqrt(rank)):
        fail(case, 'sddmm', 'rank=%d pairs=%d %s pred %.1f eps sse %.1f eps' % (
            rank, n_pairs, dtype, e_pred, e_sse))
        continue
    # ---- top-k ------------------------------------------------------------------------------
    k = int(min(m, rng.choice([1, 2, 10, 100, 1024])))
    n_sel = int(min(n, rng.choice([1, 7, 64, 300])))
    sel = rng.choice(n, size=n_sel, replace=False)
    items_k, scores_k = dev.topk_items(rank, to(sel, torch.int32), Ud, Vd, k)
    S = Uq[:, sel].T.dot(Vq)                                   # [n_sel x m]
    got_items = items_k.cpu().numpy().astype(np.int64)
    got_scores = scores_k.cpu().numpy().astype(np.float64)
    tol = (4 * np.sqrt(rank) + 4) * eps * (np.abs(Uq[:, sel]).T.dot(np.abs(Vq))).max()
    ok = True
    for r in range(n_sel):
        # the reported scores are the scores of the reported items, in non-increasing order, no
        # item twice, and no unreported item beats the last reported one (up to rounding)
        if len(set(got_items[r])) != k or np.abs(S[r, got_items[r]] - got_scores[r]).max() > tol:
            ok = False
        if (np.diff(got_scores[r]) > 0).any():
            ok = False
        rest = np.delete(S[r], got_items[r])
        if rest.size and rest.max() > got_scores[r, -1] + 2 * tol:
            ok = False
    if not ok:
        fail(case, 'topk', 'rank=%d n_sel=%d m=%d k=%d %s' % (rank, n_sel, m, k, dtype))
    if case % 50 == 49:
        print('... %d cases, %d failed' % (case + 1, bad), flush=True)
print('%d cases, %d failed' % (n_cases, bad))
for key in sorted(worst):
    print('  worst %-8s %-5s %.2f eps' % (key[0], key[1], worst[key]))
sys.exit(1 if bad else 0)
