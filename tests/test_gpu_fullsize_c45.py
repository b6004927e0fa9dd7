"""BASELINE configs 4 and 5 at their FULL size on one GPU.

Config 4: ALS rank 128 on 10 M users x 1 M items with 1 B stored entries (fp32).  The byte
offsets of the entries pass 2^31 in the CSR, the CSC and the row-id array of the error pass (4 GB
per array), the user table is 5.1 GB.
The oracle cannot sweep a billion entries, but a row's update depends only on that row's
entries and the other side's factors: the C restatement runs on a SAMPLE of more than 300 rows
per side (the longest, every short-row class, rows whose entries lie beyond byte offset 2^31), fed
with exactly the source rows those rows touch.

Config 5: DirtyIMC (src/dimc.py:116-162) on 2 M users x 200 000 items with 100 M samples, user x 128
and item x 256 attribute tables in the attribute + index form, ALS rank 64 on the residual.
Checked through what does not depend on the size: the IMC stage's predictions of a 1 % user
sample against a host computation, its cost / gradient / Hessian-vector product on that
sample against the C restatement, the squared error of all 100 M samples against the sum over
the predictions, the decomposition prediction = IMC part + ALS part, score = stored error."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

C4 = dict(n_block=1250000, m=1000000, nnz_block=125000000, blocks=8, rank=128)


def row_rel_err(got, want):
    return np.linalg.norm(got - want, axis=1) / np.maximum(np.linalg.norm(want, axis=1), 1e-30)


def pick_rows(ptr_h, seed):
    """More than 300 rows: the 3 longest, 20 of the next-longest thousand, 40 of every short-row
    class (<= 16, 17-32, ... 81-96 entries), 60 between 97 entries and one chunk, 60 whose
    entries lie beyond byte offset 2^31 of the 4-byte index and value arrays, 40 at random, the first and the last row."""
    rng = np.random.default_rng(seed)
    deg = np.diff(ptr_h)
    order = np.argsort(-deg, kind='stable')
    picks = [order[:3], rng.choice(order[3:1003], 20, replace=False), [0, deg.size - 1]]
    for lo, hi in ((1, 16), (17, 32), (33, 48), (49, 64), (65, 80), (81, 96), (97, 2048)):
        pool = np.flatnonzero((deg >= lo) & (deg <= hi))
        if pool.size:
            picks.append(rng.choice(pool, min(60 if lo == 97 else 40, pool.size), replace=False))
    beyond = np.flatnonzero(ptr_h[:-1] > (1 << 29))
    assert beyond.size > 1000, 'the structure does not pass byte offset 2^31'
    picks.append(rng.choice(beyond, 60, replace=False))
    picks.append(rng.choice(deg.size, 40, replace=False))
    rows = np.unique(np.concatenate([np.asarray(p, dtype=np.int64) for p in picks]))
    if rows.size < 320:   # (a side without short rows: the item side at this size) top up
        rows = np.unique(np.concatenate([rows, rng.choice(deg.size, 400 - rows.size, False)]))
    return rows, deg, order[:3]


def oracle_rows(torch, co, side, rows, ptr_h, src_table, rank, alpha, heavy):
    """The C restatement on the given rows of a device-side structure.  Only the source rows
    those rows touch are brought to the host (renumbered); the heaviest rows are spread over
    the oracle's thread chunks of 16 rows."""
    heavy_set = set(int(h) for h in heavy)
    layout = [int(r) for r in rows if int(r) not in heavy_set]
    for k, h in enumerate(heavy):          # positions 0, 16, 32: one OpenMP chunk each
        layout.insert(min(16 * k, len(layout)), int(h))
    layout = np.asarray(layout, dtype=np.int64)
    idx_parts, val_parts, degs = [], [], []
    for r in layout:
        lo, hi = int(ptr_h[r]), int(ptr_h[r + 1])
        idx_parts.append(side.indices[lo:hi])
        val_parts.append(side.vals[lo:hi])
        degs.append(hi - lo)
    cat = torch.cat(idx_parts).to(torch.int64)
    uniq, inv = torch.unique(cat, return_inverse=True)
    small = src_table[uniq][:, :rank].to(torch.float64).cpu().numpy()
    remap = inv.to(torch.int32).cpu().numpy()
    vals = torch.cat(val_parts).to(torch.float64).cpu().numpy()
    sub_ptr = np.concatenate(([0], np.cumsum(degs))).astype(np.int64)
    want = co.half_sweep(sub_ptr, remap, vals, small, alpha, threads=16)
    return layout, want, np.asarray(degs)


def test_c4_full_1gpu():
    import torch
    from alsmi import device as dev
    from alsmi import synth
    from alsmi.engine import AlsEngine
    from oracle import c_oracle as co
    device = dev.require_gpu()
    t0 = time.perf_counter()
    P = synth.sharded_problem(20242, C4['n_block'], C4['m'], C4['nnz_block'], 1, 0, device,
                              n_blocks=C4['blocks'])
    rank = C4['rank']
    n, m = P['shape']
    assert (n, m) == (10000000, 1000000) and int(P['csr'][1].numel()) == 1000000000
    eng = AlsEngine(P['csr'], P['csc'], P['shape'], rank, 0.1, dtype='float32')
    del P
    assert eng.nnz == 1000000000 and eng.plan_i.info()['n_chunks'] > 1000
    g = torch.Generator(device=device)
    g.manual_seed(42)
    V0 = torch.zeros_like(eng.V)
    V0[:, :rank] = torch.rand((m, rank), generator=g, device=device)
    V0[:, 0] = 3.5

    def one_sweep():
        eng.V.copy_(V0)
        eng.U.zero_()
        eng.half_sweep_users()
        U1 = eng.U.clone()
        eng.half_sweep_items()
        return U1, eng.V.clone()

    U1, V1 = one_sweep()
    print('c4 full: set-up + first sweep %.1f s' % (time.perf_counter() - t0))
    t0 = time.perf_counter()
    classes = ((0, 0), (1, 16), (17, 32), (33, 48), (49, 64), (65, 80), (81, 96), (97, 2048),
               (2049, 1 << 40))
    seen, exists = {}, {}
    for side, table, src, seed in ((eng.csr, U1, V0, 3), (eng.csc, V1, U1, 4)):
        ptr_h = side.indptr_host
        assert int(ptr_h[-1]) == 1000000000
        rows, deg, heavy = pick_rows(ptr_h, seed)
        assert rows.size > 300
        layout, want, d = oracle_rows(torch, co, side, rows, ptr_h, src, rank, 0.1, heavy)
        got = table[torch.from_numpy(layout).to(device)][:, :rank].to(torch.float64).cpu().numpy()
        err = row_rel_err(got, want)
        for lo, hi in classes:
            seen[(lo, hi)] = seen.get((lo, hi), 0) + int(((d >= lo) & (d <= hi)).sum())
            exists[(lo, hi)] = exists.get((lo, hi), 0) + int(((deg >= lo) & (deg <= hi)).sum())
        assert err.max() < 1e-4, (err.max(), d[np.argmax(err)], layout[np.argmax(err)])
        assert (ptr_h[layout] > (1 << 29)).sum() >= 60
        empty = layout[d == 0]
        assert not table[torch.from_numpy(empty).to(device)].any()
    # every kernel class that occurs at this size is in the sample (users have at least 20
    # entries and the least popular item about 100, so there are no rows of 16 entries or
    # fewer here; the one-GPU slice of this configuration in test_gpu_fullsize.py has them)
    assert all(seen[c] >= min(20, exists[c]) for c in classes), (seen, exists)
    assert sum(1 for c in classes if exists[c]) >= 7
    print('c4 full: oracle on the sampled rows %.1f s, rows per class %s'
          % (time.perf_counter() - t0, seen))
    assert not U1[:, rank:].any() if rank < eng.kp else True
    # the error the next user half derives for the factors it replaces against the standalone
    # pass over all 10^9 entries (a 4 GB row-id array): 1e-5 on the RMSE
    rmse = eng.train_rmse()
    fused = float(np.sqrt(float(eng.half_sweep_users_err().cpu()[0]) / eng.nnz))
    assert np.isfinite(rmse) and abs(fused - rmse) <= 1e-5 * rmse, (fused, rmse)
    assert torch.isfinite(eng.U).all() and torch.isfinite(eng.V).all()
    # run-to-run bit equality of both half-sweeps at this size
    U2, V2 = one_sweep()
    assert torch.equal(U1, U2) and torch.equal(V1, V2)
    del U1, U2, V1, V2, V0, eng
    torch.cuda.empty_cache()


def test_dimc_c5_full_1gpu():
    import warnings
    import torch
    from alsmi import DirtyIMC
    from alsmi import device as dev
    from alsmi import synth
    from alsmi.imc import ImcProblem
    from oracle import c_oracle as co
    device = dev.require_gpu()
    n, m, ns, p, q, rank = 2000000, 200000, 100000000, 128, 256, 64
    t0 = time.perf_counter()
    users_t, items_t, _ = synth.user_block(20245, 0, n, m, ns, device)
    g = torch.Generator(device=device)
    g.manual_seed(9)
    X = torch.randn((n, p), generator=g, device=device, dtype=torch.float64) / np.sqrt(p)
    Y = torch.randn((m, q), generator=g, device=device, dtype=torch.float64) / np.sqrt(q)
    Zs = torch.randn((p, 16), generator=g, device=device, dtype=torch.float64).mm(
        torch.randn((16, q), generator=g, device=device, dtype=torch.float64))
    A = X.mm(Zs)
    P8 = torch.randn((n, 8), generator=g, device=device, dtype=torch.float64) * 0.5
    Q8 = torch.randn((m, 8), generator=g, device=device, dtype=torch.float64) * 0.5
    vals_t = torch.empty(ns, dtype=torch.float64, device=device)
    resid_sq = 0.0
    for s in range(0, ns, 1 << 22):
        e = min(ns, s + (1 << 22))
        side = (A[users_t[s:e]] * Y[items_t[s:e]]).sum(-1)
        resid = (P8[users_t[s:e]] * Q8[items_t[s:e]]).sum(-1)
        resid_sq += float((resid * resid).sum())
        vals_t[s:e] = side + resid
    vals_t += torch.randn(ns, generator=g, device=device, dtype=torch.float64) * 0.1
    resid_var = resid_sq / ns
    users, items, vals = users_t.cpu().numpy(), items_t.cpu().numpy(), vals_t.cpu().numpy()
    Xh, Yh = X.cpu().numpy(), Y.cpu().numpy()
    del A, X, Y, P8, Q8, vals_t, users_t, items_t
    torch.cuda.empty_cache()
    print('c5 full: generated in %.1f s' % (time.perf_counter() - t0))
    args = (Xh, Yh, users, items)
    # (a tiny imc_alpha: the reference's cost counts the regulariser once per sample,
    # src/imc.py:47, its gradient once, so at 10^8 samples any larger value makes every step
    # look uphill to the line search)
    d = DirtyIMC(rank=rank, tol=1e-3, random_state=1, imc_alpha=1e-10, als_alpha=0.05)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        Z, u, i = d.fit_transform(args, vals, shape=(n, m))
    torch.cuda.synchronize()
    fit_s = time.perf_counter() - t0
    print('c5 full: DirtyIMC.fit_transform %.2f s, %d ALS sweeps' % (fit_s, d.als.n_iter_))
    assert fit_s < 10.0
    assert Z.shape == (p, q) and u.shape == (rank, n) and i.shape == (rank, m)
    assert np.isfinite(Z).all() and np.isfinite(u).all() and np.isfinite(i).all()
    total_rmse = float(np.sqrt(np.mean(vals ** 2)))
    imc_rmse = d.imc.reconstruction_err_
    dimc_rmse = -d.reconstruction_err_
    # IMC explains the side-feature part, ALS on the residual most of the rest
    assert imc_rmse < 1.15 * np.sqrt(resid_var + 0.01) and imc_rmse < 0.6 * total_rmse
    assert dimc_rmse < 0.75 * imc_rmse

    # ---- the IMC stage at full size: predictions of all 10^8 samples ---------------------
    prob = ImcProblem(Xh, Yh, users, items, vals)
    assert prob.order is None                      # the samples come user by user
    t0 = time.perf_counter()
    _, sse_dev, pred_dev = prob._eval_dev(0, Z, want_out=False, want_sse=True, want_pred=True)
    sse = float(sse_dev.cpu()[0])
    # the squared error of the launch against the sum over its own predictions (fp64, torch)
    want_sse = float(((prob.ratings - pred_dev) ** 2).sum())
    np.testing.assert_allclose(sse, want_sse, rtol=1e-11)
    np.testing.assert_allclose(np.sqrt(sse / ns), imc_rmse, rtol=1e-9)
    # a 1 % sample of the users (the first 20 000: the samples are grouped by user, so they
    # are a prefix of the sample arrays): predictions against the host, then cost, gradient and
    # Hessian-vector product of that sub-problem against the C restatement
    nu = n // 100
    cut = int(np.searchsorted(users, nu))
    assert cut > 500000
    Ah = Xh[:nu].dot(Z)
    want_pred = np.einsum('ij,ij->i', Ah[users[:cut]], Yh[items[:cut]])
    got_pred = pred_dev[:cut].cpu().numpy()
    assert np.abs(got_pred - want_pred).max() <= 1e-11 * np.abs(want_pred).max()
    # ... and of the LAST 1 % of the users (the far end of every array)
    lo_u = n - nu
    lo_s = int(np.searchsorted(users, lo_u))
    Ah = Xh[lo_u:].dot(Z)
    want_pred = np.einsum('ij,ij->i', Ah[users[lo_s:] - lo_u], Yh[items[lo_s:]])
    got_pred = pred_dev[lo_s:].cpu().numpy()
    assert np.abs(got_pred - want_pred).max() <= 1e-11 * np.abs(want_pred).max()
    del pred_dev
    sub = ImcProblem(Xh[:nu], Yh, users[:cut], items[:cut], vals[:cut])
    rng = np.random.default_rng(3)
    S = rng.standard_normal((p, q)) * 0.1
    sub_ptr = np.concatenate(([0], np.cumsum(np.bincount(users[:cut], minlength=nu)))).astype(np.int64)
    for mode, M in ((0, Z), (1, S)):
        got, got_sse, _ = sub._eval(mode, M, want_sse=(mode == 0))
        want, want_s = co.imc_eval(mode, sub_ptr, items[:cut].astype(np.int32), vals[:cut],
                                   Xh[:nu], Yh, M, threads=16)
        assert np.abs(got - want).max() <= 1e-10 * np.abs(want).max(), mode
        if mode == 0:
            np.testing.assert_allclose(got_sse, want_s, rtol=1e-11)
    # the whole problem's gradient and Hessian-vector product are sums over the samples: the
    # 1 % sub-problem plus the other 99 % (a second problem on the remaining users) give the
    # full launch's result
    rest = ImcProblem(Xh[nu:], Yh, users[cut:] - nu, items[cut:], vals[cut:])
    for mode, M in ((0, Z), (1, S)):
        full, _, _ = prob._eval(mode, M)
        a, _, _ = sub._eval(mode, M)
        b, _, _ = rest._eval(mode, M)
        assert np.abs(full - (a + b)).max() <= 1e-10 * np.abs(full).max(), mode
    print('c5 full: IMC stage checks %.1f s' % (time.perf_counter() - t0))
    del prob, sub, rest
    torch.cuda.empty_cache()

    # ---- the estimator's decomposition, on a subset given in another order ---------------
    sel = np.random.default_rng(0).choice(ns, 500000, replace=False)
    subset = (Xh, Yh, users[sel], items[sel])
    pred = d.predict(subset)
    uv = d.als._pair_predictions(users[sel], items[sel])
    xzy = np.einsum('ij,ij->i', Xh[users[sel]].dot(Z), Yh[items[sel]])
    np.testing.assert_allclose(pred, uv + xzy, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(-np.sqrt(np.mean((vals[sel] - pred) ** 2)),
                               d.score(subset, vals[sel]), rtol=1e-12)
    full = d.score(args, vals)
    np.testing.assert_allclose(full, d.reconstruction_err_, rtol=1e-9)
    # ---- the ALS stage at full size ---------------------------------------------------------
    # The fit ends with an item half-sweep, so every stored item row solves the normal
    # equations built from the stored user factors and the residual r - x^T Z y
    # (src/dimc.py:155-160, src/als.py:243-247): 300 sampled items (the most rated ones
    # included) against the C restatement fed with the residual computed on the host.
    t0 = time.perf_counter()
    rng = np.random.default_rng(1)
    counts = np.bincount(items, minlength=m)
    sel_items = np.unique(np.concatenate([np.argsort(-counts)[:5], rng.choice(m, 295, False)]))
    mask = np.isin(items, sel_items)
    su, si, sv = users[mask], items[mask], vals[mask]
    resid = sv - np.einsum('ij,ij->i', Xh[su].dot(Z), Yh[si])
    keep = resid != 0
    su, si, resid = su[keep], si[keep], resid[keep]
    order = np.argsort(si, kind='stable')          # item-major, users ascending inside an item
    su, si, resid = su[order], si[order], resid[order]
    ptr = np.concatenate(([0], np.cumsum(np.bincount(np.searchsorted(sel_items, si),
                                                     minlength=sel_items.size)))).astype(np.int64)
    want_i = co.half_sweep(ptr, su.astype(np.int32), resid, np.ascontiguousarray(u.T), 0.05,
                           threads=16)
    err = row_rel_err(i[:, sel_items].T, want_i)
    assert err.max() < 1e-8, (err.max(), counts[sel_items][np.argmax(err)])
    assert counts[sel_items].max() > 100000
    print('c5 full: ALS stage check %.1f s' % (time.perf_counter() - t0))
