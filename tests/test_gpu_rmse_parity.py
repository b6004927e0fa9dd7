"""RMSE parity at the benchmark shape (BASELINE config 2: 138 493 x 26 744, 20 M stored entries,
rank 64): the reference ends every sweep with the training RMSE over all entries
(src/als.py:161-167) and scores held-out pairs through ALS.score (src/als.py:383-388).

From the same initial factors (src/als.py:143-152) ten sweeps run in fp32 (the throughput mode:
products on the fp16 matrix cores from a scaled two-way split) and in fp64 (the reference's
arithmetic); after every sweep the training RMSE and the RMSE on 1 % held-out entries of the
same planted-factor generator must agree within 1e-5 relative.  On the factors after sweep
ten — where the normal matrices are closest to rank deficiency — one more half-sweep per
side is compared row by row with the C restatement of src/als.py:218-250 fed with those very
factors (fp32: 1e-4 per row)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_USERS, N_ITEMS, NNZ, HELD, RANK, ALPHA, SWEEPS = 138493, 26744, 20000000, 200000, 64, 0.1, 10


def row_rel_err(got, want):
    return np.linalg.norm(got - want, axis=1) / np.maximum(np.linalg.norm(want, axis=1), 1e-30)


@pytest.fixture(scope='module')
def split_problem():
    import torch
    from alsmi import device as dev
    from alsmi import synth
    device = dev.require_gpu()
    users, items, vals = synth.user_block(20242, 0, N_USERS, N_ITEMS, NNZ + HELD, device)
    g = torch.Generator(device=device)
    g.manual_seed(5)
    held = torch.zeros(NNZ + HELD, dtype=torch.bool, device=device)
    held[torch.randperm(NNZ + HELD, generator=g, device=device)[:HELD]] = True
    tu, ti, tv = users[~held], items[~held].to(torch.int32), vals[~held]
    indptr = synth.csr_from_sorted(tu, N_USERS)
    csr = (indptr, ti, tv)
    csc = dev.csr_to_csc(indptr, ti, tv, (N_USERS, N_ITEMS))
    # initial item factors of the reference: uniform, row 0 = mean rating of the item
    sums = torch.zeros(N_ITEMS, dtype=torch.float64, device=device)
    cnts = torch.zeros(N_ITEMS, dtype=torch.float64, device=device)
    sums.index_add_(0, ti.to(torch.int64), tv.to(torch.float64))
    cnts.index_add_(0, ti.to(torch.int64), torch.ones_like(tv, dtype=torch.float64))
    V0 = synth.init_factors(42, RANK, N_ITEMS, sums.cpu().numpy(), cnts.cpu().numpy())
    test = (users[held].to(torch.int32), items[held].to(torch.int32), vals[held])
    return dict(torch=torch, device=device, csr=csr, csc=csc, V0=V0, test=test)


def run_sweeps(P, dtype):
    from alsmi.engine import AlsEngine
    torch = P['torch']
    eng = AlsEngine(P['csr'], P['csc'], (N_USERS, N_ITEMS), RANK, ALPHA, dtype=dtype)
    assert eng.nnz == NNZ
    eng.set_factors(np.zeros((RANK, N_USERS)), P['V0'])
    tu, ti, tv = P['test']
    tv = tv.to(eng.U.dtype)
    train, test, fused = [], [], []
    for s in range(SWEEPS):
        if s == 0:
            eng.half_sweep_users()
        else:   # the error of the factors of sweep s, as ALS.fit obtains it inside the loop
            fused.append(float(np.sqrt(float(eng.half_sweep_users_err().cpu()[0]) / eng.nnz)))
        eng.half_sweep_items()
        train.append(eng.train_rmse())
        _, sse = eng.backend.sddmm(RANK, tu, ti, eng.U, eng.V, tv, False)
        test.append(float(np.sqrt(float(sse.cpu()[0]) / HELD)))
    return eng, np.array(train), np.array(test), np.array(fused)


def test_rmse_trajectories_fp32_vs_fp64_and_converged_rows(split_problem):
    from oracle import c_oracle as co
    P = split_problem
    torch = P['torch']
    eng64, train64, test64, fused64 = run_sweeps(P, 'float64')
    del eng64
    eng, train32, test32, fused32 = run_sweeps(P, 'float32')
    print('train RMSE fp64', train64)
    print('test  RMSE fp64', test64)
    rel_train = np.abs(train32 - train64) / train64
    rel_test = np.abs(test32 - test64) / test64
    print('max relative difference fp32 vs fp64: train %.2e, held-out %.2e'
          % (rel_train.max(), rel_test.max()))
    assert np.isfinite(train32).all() and np.isfinite(test32).all()
    assert rel_train.max() <= 1e-5, rel_train
    assert rel_test.max() <= 1e-5, rel_test
    # the error the loop itself sees (taken from the next user half-sweep's normal equations)
    assert np.abs(fused32 - train32[:-1]).max() <= 1e-5 * train32.min()
    assert np.abs(fused64 - train64[:-1]).max() <= 1e-9 * train64.min()
    # the fit does what the planted model allows: held-out error near the noise floor of the
    # generator (N(0, .5) noise + rounding to half stars), training error below it
    assert train64[-1] < train64[0] and test64[-1] < 0.9 and train64[-1] < test64[-1]

    # ---- sampled rows against the oracle on the factors after sweep ten --------------------
    csr_h = tuple(t.cpu().numpy() for t in P['csr'])
    csc_h = tuple(t.cpu().numpy() for t in P['csc'])
    for side, (indptr, indices, vals), src, plan_call, table in (
            ('user', csr_h, eng.V, eng.half_sweep_users, lambda: eng.U),
            ('item', csc_h, eng.U, eng.half_sweep_items, lambda: eng.V)):
        src_h = src[:, :RANK].to(torch.float64).cpu().numpy()
        plan_call()
        got = table()[:, :RANK].to(torch.float64).cpu().numpy()
        deg = np.diff(indptr)
        order = np.argsort(-deg, kind='stable')
        rng = np.random.default_rng(11)
        picks = [order[:16], rng.choice(deg.size, 250, replace=False)]
        for lo, hi in ((1, 16), (17, 32), (33, 48), (49, 64), (65, 128)):
            pool = np.flatnonzero((deg >= lo) & (deg <= hi))
            if pool.size:
                picks.append(rng.choice(pool, min(30, pool.size), replace=False))
        rows = np.unique(np.concatenate(picks))
        sub_ptr = np.concatenate(([0], np.cumsum(deg[rows]))).astype(np.int64)
        sel = np.concatenate([np.arange(indptr[r], indptr[r + 1]) for r in rows])
        want = co.half_sweep(sub_ptr, indices[sel], vals[sel].astype(np.float64), src_h, ALPHA,
                             threads=16)
        err = row_rel_err(got[rows], want)
        print('%s side after sweep %d: %d sampled rows, worst relative row error %.2e'
              % (side, SWEEPS, rows.size, err.max()))
        assert err.max() < 1e-4, (side, err.max(), deg[rows][np.argmax(err)])
