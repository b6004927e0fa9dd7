"""Parity at the benchmark's full size (BASELINE config 2: 138 493 x 26 744, 20 M entries,
rank 64) on sampled rows and through size-independent properties.

The oracle cannot sweep the whole matrix in seconds, but a row's update depends only on that
row's entries and the other side's factors, so the C restatement is run on a SAMPLE of rows
(the longest ones, which take the chunked path, the shortest, and random ones) and compared
row for row; the squared-error pass is checked on a row range the same way.  Whole-matrix
properties: every updated row satisfies its normal equations, and the training error never
increases from one sweep to the next."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_USERS, N_ITEMS, NNZ, RANK = 138493, 26744, 20000000, 64


@pytest.fixture(scope='module')
def problem():
    import torch
    from alsmi import device as dev
    from alsmi import synth
    from alsmi.engine import AlsEngine
    device = dev.require_gpu()
    prob = synth.sharded_problem(20242, N_USERS, N_ITEMS, NNZ, 1, 0, device)
    host = {k: tuple(t.cpu().numpy() for t in prob[k]) for k in ('csr', 'csc')}
    rs = np.random.RandomState(42)
    V0 = rs.rand(RANK, N_ITEMS)
    return dict(device=device, torch=torch, prob=prob, host=host, V0=V0, AlsEngine=AlsEngine)


def sample_rows(indptr, n_random, seed):
    deg = np.diff(indptr)
    order = np.argsort(-deg, kind='stable')
    rng = np.random.default_rng(seed)
    picks = np.concatenate([order[:24], order[-8:], rng.choice(deg.size, n_random, replace=False)])
    return np.unique(picks), deg


def oracle_rows(co, csr, rows, src_rows, alpha):
    indptr, indices, vals = csr
    out = np.zeros((rows.size, src_rows.shape[1]))
    for k, r in enumerate(rows):
        lo, hi = int(indptr[r]), int(indptr[r + 1])
        sub_ptr = np.array([0, hi - lo], dtype=np.int64)
        out[k] = co.half_sweep(sub_ptr, indices[lo:hi], vals[lo:hi].astype(np.float64), src_rows,
                               alpha, threads=1)[0]
    return out


def row_rel_err(got, want):
    return np.linalg.norm(got - want, axis=1) / np.maximum(np.linalg.norm(want, axis=1), 1e-30)


@pytest.mark.parametrize('dtype,tol', [('float32', 1e-4), ('float64', 1e-10)])
def test_sampled_rows_match_oracle_at_full_size(problem, dtype, tol):
    from oracle import c_oracle as co
    torch, P = problem['torch'], problem['prob']
    eng = problem['AlsEngine'](P['csr'], P['csc'], P['shape'], RANK, 0.1, dtype=dtype)
    assert eng.plans_u[0].info()['n_chunks'] > 0 and eng.plan_i.info()['n_chunks'] > 1000
    eng.set_factors(np.zeros((RANK, N_USERS)), problem['V0'])
    Vq = eng.V[:, :RANK].to(torch.float64).cpu().numpy()

    eng.half_sweep_users()
    U = eng.U[:, :RANK].to(torch.float64).cpu().numpy()
    rows, deg = sample_rows(problem['host']['csr'][0], 150, 1)
    assert deg[rows].max() > 2048            # the chunked path is in the sample
    want = oracle_rows(co, problem['host']['csr'], rows, Vq, 0.1)
    assert row_rel_err(U[rows], want).max() < tol

    eng.half_sweep_items()
    V = eng.V[:, :RANK].to(torch.float64).cpu().numpy()
    rows, deg = sample_rows(problem['host']['csc'][0], 150, 2)
    assert deg[rows].max() > 20000
    want = oracle_rows(co, problem['host']['csc'], rows, U, 0.1)
    assert row_rel_err(V[rows], want).max() < tol
    empty = np.flatnonzero(deg == 0)
    assert not V[empty].any()

    # squared error: the kernel over all 20 M entries against the oracle over the same entries
    indptr, indices, vals = problem['host']['csr']
    sse = float(eng.train_sse_async().cpu()[0])
    want_sse = co.sse(indptr, indices, vals.astype(np.float64), U, V, threads=8)
    np.testing.assert_allclose(sse, want_sse, rtol=2e-6 if dtype == 'float32' else 1e-11)


def test_whole_matrix_properties_fp32(problem):
    torch, P = problem['torch'], problem['prob']
    eng = problem['AlsEngine'](P['csr'], P['csc'], P['shape'], RANK, 0.1, dtype='float32')
    eng.set_factors(np.zeros((RANK, N_USERS)), problem['V0'])
    rmse = []
    for _ in range(4):
        eng.sweep()
        rmse.append(eng.train_rmse())
    assert all(np.isfinite(rmse)) and all(b <= a * (1 + 1e-6) for a, b in zip(rmse, rmse[1:]))
    # the error a user half-sweep derives from its own normal equations (for the factors it
    # replaces) against the standalone pass over all entries, on converging factors where
    # its three terms cancel most: the 1e-5 RMSE bar
    for keep_old in (True, False):
        fused = float(eng.half_sweep_users_err(keep_old=keep_old).cpu()[0])
        fused_rmse = float(np.sqrt(fused / eng.nnz))
        assert abs(fused_rmse - rmse[-1]) <= 1e-5 * rmse[-1], (fused_rmse, rmse[-1])
        if keep_old:
            eng.commit_users()
        eng.half_sweep_items()
        rmse.append(eng.train_rmse())
    assert torch.isfinite(eng.U).all() and torch.isfinite(eng.V).all()
    assert not eng.U[:, RANK:].any() and not eng.V[:, RANK:].any()
    # normal equations: (S S^T + alpha n I) x = S r for a block of rows, in fp64 on the host
    eng.half_sweep_users()
    indptr, indices, vals = problem['host']['csr']
    U = eng.U[:, :RANK].to(torch.float64).cpu().numpy()
    V = eng.V[:, :RANK].to(torch.float64).cpu().numpy()
    worst = 0.0
    for r in range(5000, 5200):
        lo, hi = int(indptr[r]), int(indptr[r + 1])
        S = V[indices[lo:hi]]
        b = S.T.dot(vals[lo:hi].astype(np.float64))
        res = S.T.dot(S.dot(U[r])) + 0.1 * (hi - lo) * U[r] - b
        worst = max(worst, np.linalg.norm(res) / max(np.linalg.norm(b), 1e-30))
    assert worst < 2e-5


def test_sharded_ingest_at_full_size(problem):
    """AlsEngine.from_shards on the 20 M triplets in shuffled order (one rank: no exchange, the
    ordering and the pointer arrays are what is tested at this size) gives the generator's own
    CSR and CSC bit for bit."""
    torch, P = problem['torch'], problem['prob']
    indptr, indices, vals = P['csr']
    users = torch.repeat_interleave(torch.arange(N_USERS, device=indptr.device),
                                    indptr[1:] - indptr[:-1])
    g = torch.Generator(device=indptr.device)
    g.manual_seed(1)
    perm = torch.randperm(NNZ, generator=g, device=indptr.device)
    eng = problem['AlsEngine'].from_shards(users[perm], indices[perm], vals[perm],
                                           P['shape'], RANK, 0.1, dtype='float32')
    assert eng.nnz == NNZ
    for side, want in ((eng.csr, P['csr']), (eng.csc, P['csc'])):
        assert np.array_equal(side.indptr_host, want[0].cpu().numpy())
        assert torch.equal(side.indices, want[1]) and torch.equal(side.vals, want[2])
    # the per-item means of the initial factors (src/als.py:144-148)
    cptr, _, cvals = P['csc']
    cnt = (cptr[1:] - cptr[:-1]).to(torch.float64)
    sums = torch.segment_reduce(cvals.to(torch.float64), 'sum', offsets=cptr, initial=0.0)
    want = torch.where(cnt > 0, sums / cnt.clamp_min(1), torch.zeros_like(sums))
    assert torch.equal(eng.item_means(), want)


# ---- one GPU's share of the 1 B-entry rank-128 configuration -------------------------------
# 1.25 M users x 1 M items, 125 M entries, rank 128 (BASELINE config 4 cut over 8 GPUs), fp32.
# Most rows here are shorter than the rank and take the neighbour-by-neighbour kernels (<= 32
# and <= 64 entries); the popular items are cut into chunks.  The sample covers all of them.
S_USERS, S_ITEMS, S_NNZ, S_RANK = 1250000, 1000000, 125000000, 128


def device_rows(csr, rows):
    """(indptr, indices, vals) of the given rows only, copied from device tensors."""
    import torch
    indptr, indices, vals = csr
    ptr_h = indptr.cpu().numpy()
    out = []
    for r in rows:
        lo, hi = int(ptr_h[r]), int(ptr_h[r + 1])
        out.append((indices[lo:hi].cpu().numpy(), vals[lo:hi].to(torch.float64).cpu().numpy()))
    return ptr_h, out


def test_rank128_slice_sampled_rows_match_oracle():
    import torch
    from alsmi import device as dev
    from alsmi import synth
    from alsmi.engine import AlsEngine
    from oracle import c_oracle as co
    device = dev.require_gpu()
    P = synth.sharded_problem(20242, S_USERS, S_ITEMS, S_NNZ, 1, 0, device)
    eng = AlsEngine(P['csr'], P['csc'], P['shape'], S_RANK, 0.1, dtype='float32')
    assert eng.plan_i.info()['n_chunks'] > 100
    g = torch.Generator(device=device)
    g.manual_seed(42)
    eng.V[:, :S_RANK] = torch.rand((S_ITEMS, S_RANK), generator=g, device=device)
    eng.V[:, 0] = 3.5
    Vq = eng.V[:, :S_RANK].to(torch.float64).cpu().numpy()

    def check(side_csr, table, src64, seed):
        ptr_h, _ = device_rows(side_csr, [])
        rows, deg = sample_rows(ptr_h, 200, seed)
        _, data = device_rows(side_csr, rows)
        got = table[torch.from_numpy(rows).to(device)][:, :S_RANK].to(torch.float64).cpu().numpy()
        want = np.zeros_like(got)
        for k, (idx, val) in enumerate(data):
            want[k] = co.half_sweep(np.array([0, idx.size], dtype=np.int64), idx, val, src64, 0.1,
                                    threads=1)[0]
        err = row_rel_err(got, want)
        d = deg[rows]
        # every kernel is in the sample: <= 32, <= 64 and <= 96 entries (neighbour form), up to
        # one chunk (rank form) and the chunked rows
        for lo, hi in ((1, 32), (33, 64), (65, 96), (97, 2048)):
            assert ((d >= lo) & (d <= hi)).any(), (lo, hi)
        assert err.max() < 1e-4, (err.max(), d[np.argmax(err)])
        return d

    eng.half_sweep_users()
    d = check(P['csr'], eng.U, Vq, 3)
    U64 = eng.U[:, :S_RANK].to(torch.float64).cpu().numpy()
    eng.half_sweep_items()
    d = check(P['csc'], eng.V, U64, 4)
    assert d.max() > 4096
    del U64, Vq
    # the error the next user half derives for the factors it replaces against the standalone
    # pass over all 125 M entries (1e-5 on the RMSE), and pad columns stay zero
    rmse = eng.train_rmse()
    fused = float(np.sqrt(float(eng.half_sweep_users_err().cpu()[0]) / eng.nnz))
    assert abs(fused - rmse) <= 1e-5 * rmse, (fused, rmse)
    assert torch.isfinite(eng.U).all() and torch.isfinite(eng.V).all()


def test_rank128_source_table_above_4gib():
    """At padded rank 128 the fp32 gather addresses the source table with 64 bits: rows that
    lie beyond the 4 GiB mark of a 4.6 GB table give the same vectors as the same rows packed
    into a small table (all three kernels: chunked, rank form, neighbour form)."""
    import torch
    from alsmi import device as dev
    from oracle import c_oracle as co
    device = dev.require_gpu()
    rank, kp, n_src = 128, 128, 9000000          # 9 M x 512 B = 4.6 GB
    rng = np.random.default_rng(11)
    degs = [5, 30, 50, 64, 100, 700, 3000]
    used = np.unique(np.concatenate([rng.integers(8400000, n_src, 3500),     # beyond 4 GiB
                                     rng.integers(0, n_src, 500)]))
    rows_idx = [np.sort(rng.choice(used, d, replace=False)) for d in degs]
    indptr = np.concatenate(([0], np.cumsum(degs))).astype(np.int64)
    indices = np.concatenate(rows_idx).astype(np.int32)
    vals = rng.integers(1, 6, indices.size).astype(np.float64)
    small = rng.standard_normal((used.size, rank)).astype(np.float32)
    table = torch.zeros((n_src, kp), dtype=torch.float32, device=device)
    table[torch.from_numpy(used).to(device)] = torch.from_numpy(small).to(device)
    side = dev.SparseSide(indptr, indices, vals, len(degs), n_src, 'float32', device)
    plan = dev.SweepPlan(side, rank, chunk=1024)
    assert plan.info()['n_chunks'] >= 3
    out = torch.zeros((len(degs), kp), dtype=torch.float32, device=device)
    plan.half_sweep(table, out, 0.1)
    got = out.to(torch.float64).cpu().numpy()
    remap = np.searchsorted(used, indices).astype(np.int32)
    want = co.half_sweep(indptr, remap, vals, small.astype(np.float64), 0.1, threads=1)
    assert row_rel_err(got, want).max() < 1e-4
    plan.close()
    del table
    torch.cuda.empty_cache()


# ---- BASELINE config 3: IMC with dense side features at the full shape ----------------------
# users x 128 and items x 256 attributes over the 138 493 x 26 744 / 20 M-entry ratings.  The C
# restatement covers the whole problem on the host cores in seconds, so the p x q results and
# the squared error are compared directly, in fp64 and in fp32.

@pytest.fixture(scope='module')
def imc_problem(problem):
    torch, device = problem['torch'], problem['device']
    P = problem['prob']
    p, q = 128, 256
    g = torch.Generator(device=device)
    g.manual_seed(3)
    X = torch.randn((N_USERS, p), generator=g, device=device, dtype=torch.float64) / np.sqrt(p)
    Y = torch.randn((N_ITEMS, q), generator=g, device=device, dtype=torch.float64) / np.sqrt(q)
    # fp32-exact inputs, so that both builds see the same numbers
    X, Y = X.float().double(), Y.float().double()
    M = (torch.randn((p, q), generator=g, device=device, dtype=torch.float64) * 0.1).float().double()
    from oracle import c_oracle as co
    indptr, items, vals = problem['host']['csr']
    Xh, Yh, Mh = X.cpu().numpy(), Y.cpu().numpy(), M.cpu().numpy()
    vals64 = vals.astype(np.float64)
    want = {mode: co.imc_eval(mode, indptr, items, vals64, Xh, Yh, Mh, threads=16)
            for mode in (0, 1)}
    # predictions of a sample of users (the longest, the shortest, random ones)
    rows, _ = sample_rows(indptr, 60, 5)
    A = Xh[rows].dot(Mh)
    preds = {}
    for k, r in enumerate(rows):
        lo, hi = int(indptr[r]), int(indptr[r + 1])
        preds[int(r)] = (lo, hi, Yh[items[lo:hi]].dot(A[k]))
    return dict(X=X, Y=Y, M=Mh, want=want, preds=preds, csr=P['csr'])


@pytest.mark.parametrize('dtype,tol_out,tol_sse,tol_pred', [('float64', 1e-10, 1e-11, 1e-11),
                                                            ('float32', 2e-4, 2e-6, 2e-5)])
def test_imc_evaluations_at_full_size(imc_problem, dtype, tol_out, tol_sse, tol_pred):
    from alsmi.imc import ImcProblem
    I = imc_problem
    indptr, items, vals = I['csr']
    prob = ImcProblem.from_device_csr(indptr, items, vals, I['X'], I['Y'], dtype=dtype)
    out0, sse, pred = prob._eval(0, I['M'], want_sse=True, want_pred=True)
    out1, _, _ = prob._eval(1, I['M'])
    want0, want_sse = I['want'][0]
    want1, _ = I['want'][1]
    assert np.abs(out0 - want0).max() <= tol_out * np.abs(want0).max()
    assert np.abs(out1 - want1).max() <= tol_out * np.abs(want1).max()
    np.testing.assert_allclose(sse, want_sse, rtol=tol_sse)
    for r, (lo, hi, want_p) in I['preds'].items():
        scale = max(np.abs(want_p).max(), 1e-30) if hi > lo else 1.0
        assert np.abs(pred[lo:hi] - want_p).max() <= tol_pred * scale, r
    # linearity of the Hessian-vector part in its argument
    out2, _, _ = prob._eval(1, 2.0 * I['M'])
    assert np.abs(out2 - 2.0 * out1).max() <= (1e-12 if dtype == 'float64' else 1e-5) * np.abs(out1).max()


def test_dimc_config5_slice_properties():
    """One GPU's share of BASELINE config 5 (100 M samples over 8 GPUs): 250 000 users x
    200 000 items, 12.5 M samples, user x 128 and item x 256 attribute matrices in the
    attribute form (the per-sample form would be 38 GB), planted X Z Y^T plus a rank-8
    residual plus noise.  Too large for the oracle: size-independent properties."""
    import torch
    import warnings
    from alsmi import DirtyIMC, IMC
    from alsmi import device as dev
    from alsmi import synth
    device = dev.require_gpu()
    n, m, ns, p, q = 250000, 200000, 12500000, 128, 256
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
    side = torch.empty(ns, dtype=torch.float64, device=device)
    resid = torch.empty(ns, dtype=torch.float64, device=device)
    for s in range(0, ns, 1 << 20):
        e = min(ns, s + (1 << 20))
        side[s:e] = (A[users_t[s:e]] * Y[items_t[s:e]]).sum(-1)
        resid[s:e] = (P8[users_t[s:e]] * Q8[items_t[s:e]]).sum(-1)
    noise = torch.randn(ns, generator=g, device=device, dtype=torch.float64) * 0.1
    vals = (side + resid + noise).cpu().numpy()
    side_var, resid_var = float(side.var()), float(resid.var())
    users, items = users_t.cpu().numpy(), items_t.cpu().numpy()
    Xh, Yh = X.cpu().numpy(), Y.cpu().numpy()
    del A, side, resid, noise, X, Y
    args = (Xh, Yh, users, items)
    # (a tiny imc_alpha: the reference's cost counts the regulariser once per sample,
    # src/imc.py:47, while its gradient counts it once, so at 12.5 M samples any larger value
    # makes every step look uphill to the line search)
    d = DirtyIMC(rank=8, tol=1e-3, random_state=1, imc_alpha=1e-9, als_alpha=0.05)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        Z, u, i = d.fit_transform(args, vals, shape=(n, m))
    assert d._attribute_form(*args)
    assert Z.shape == (p, q) and u.shape == (8, n) and i.shape == (8, m)
    assert np.isfinite(Z).all() and np.isfinite(u).all() and np.isfinite(i).all()
    total_rmse = float(np.sqrt(np.mean(vals ** 2)))
    imc_rmse = d.imc.reconstruction_err_
    dimc_rmse = -d.reconstruction_err_
    # IMC explains the side-feature part, ALS on the residual most of the rest
    assert imc_rmse < 1.15 * np.sqrt(resid_var + 0.01) and imc_rmse < 0.6 * total_rmse
    assert dimc_rmse < 0.75 * imc_rmse
    # score and predict agree with the stored error, on a subset given in another order
    sel = np.random.default_rng(0).choice(ns, 200000, replace=False)
    sub = (Xh, Yh, users[sel], items[sel])
    pred = d.predict(sub)
    np.testing.assert_allclose(-np.sqrt(np.mean((vals[sel] - pred) ** 2)),
                               d.score(sub, vals[sel]), rtol=1e-12)
    full = d.score(args, vals)
    np.testing.assert_allclose(full, d.reconstruction_err_, rtol=1e-9)


@pytest.mark.parametrize('rank,dtype', [(64, 'float32'), (128, 'float32'), (128, 'float64')])
def test_half_sweeps_are_bit_reproducible_at_full_size(problem, rank, dtype):
    """Two runs of both half-sweeps (plain and error-summing kernels) over the whole
    138 493 x 26 744 / 20 M-entry matrix give bit-identical factors and error sums: 330 000
    row solves per run, enough to see an event of one row in 10^5.  (fp64 at rank 128: the kernel
    whose matrix instructions are partly inline assembly with their own wait states.)"""
    torch, P = problem['torch'], problem['prob']
    from alsmi import device as dev
    device = problem['device']
    kp = dev.padded_rank(rank)
    td = torch.float64 if dtype == 'float64' else torch.float32
    g = torch.Generator(device=device)
    g.manual_seed(1)
    for tri, n_rows, n_cols in ((P['csr'], N_USERS, N_ITEMS), (P['csc'], N_ITEMS, N_USERS)):
        side = dev.SparseSide(tri[0], tri[1], tri[2], n_rows, n_cols, dtype, device)
        plan = dev.SweepPlan(side, rank)
        src = torch.zeros((n_cols, kp), device=device, dtype=td)
        src[:, :rank] = torch.rand((n_cols, rank), generator=g, device=device, dtype=td)
        cur = torch.zeros((n_rows, kp), device=device, dtype=td)
        cur[:, :rank] = torch.rand((n_rows, rank), generator=g, device=device, dtype=td)
        outs = []
        for _ in range(2):
            a = torch.zeros((n_rows, kp), device=device, dtype=td)
            plan.half_sweep(src, a, 0.1)
            b = torch.zeros((n_rows, kp), device=device, dtype=td)
            sse = float(plan.half_sweep_err(src, cur, b, 0.1).cpu()[0])
            outs.append((a, b, sse))
        assert torch.equal(outs[0][0], outs[1][0])
        assert torch.equal(outs[0][1], outs[1][1])
        assert outs[0][2] == outs[1][2]
        plan.close()
