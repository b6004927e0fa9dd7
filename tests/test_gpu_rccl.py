"""The multi-rank code path on RCCL with the one GPU at hand: a process group of ONE rank
whose collectives are forced on (ALSMI_FORCE_COLLECTIVES=1), so that every transfer of the
slab gathers (the batch of point-to-point transfers, here with the rank itself, and the
broadcasts of ALSMI_GATHER=bcast), the all-reduce of the squared error and the exchange of the
slab bounds is a real RCCL call, asynchronous beside the kernels exactly as with eight ranks.  The factors must
equal those of the plain single-process path bit for bit (a row's arithmetic does not depend
on how the rows are cut into slabs)."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.fixture(scope='module')
def rccl_group():
    import torch
    import torch.distributed as dist
    from alsmi import device as dev
    device = dev.require_gpu()
    assert not dist.is_initialized()
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % free_port(), rank=0,
                            world_size=1, device_id=device)
    old = os.environ.get('ALSMI_FORCE_COLLECTIVES')
    os.environ['ALSMI_FORCE_COLLECTIVES'] = '1'
    try:
        yield dist.group.WORLD
    finally:
        if old is None:
            del os.environ['ALSMI_FORCE_COLLECTIVES']
        else:
            os.environ['ALSMI_FORCE_COLLECTIVES'] = old
        torch.cuda.synchronize()
        dist.destroy_process_group()


@pytest.mark.parametrize('mode', ['p2p', 'bcast'])
@pytest.mark.parametrize('dtype', ['float32', 'float64'])
def test_forced_gathers_on_rccl_match_the_plain_path(rccl_group, dtype, mode, monkeypatch):
    import torch
    monkeypatch.setenv('ALSMI_GATHER', mode)
    import torch.distributed as dist
    from alsmi import device as dev
    from alsmi import synth
    from alsmi.engine import AlsEngine
    device = dev.require_gpu()
    n, m, nnz, rank = 20000, 5000, 1000000, 64
    P = synth.sharded_problem(777, n, m, nnz, 1, 0, device)
    V0 = np.random.RandomState(3).rand(rank, m)

    def run(group):
        eng = AlsEngine(P['csr'], P['csc'], P['shape'], rank, 0.1, dtype=dtype, group=group,
                        user_slabs=4 if group else None, item_slabs=2 if group else None)
        eng.set_factors(np.zeros((rank, n)), V0)
        eng.time_gathers = eng.collective
        eng.half_sweep_users()
        eng.half_sweep_items()
        sse = []
        for keep_old in (True, False, False):      # three sweeps on the fused schedule
            sse.append(float(eng.half_sweep_users_err(keep_old=keep_old).cpu()[0]))
            if keep_old:
                eng.commit_users()
            eng.half_sweep_items()
        sse.append(float(eng.train_sse_async().cpu()[0]))
        return eng, sse

    plain, sse_plain = run(None)
    assert not plain.collective and len(plain.plans_u) == 1
    forced, sse_forced = run(rccl_group)
    assert dist.get_backend(rccl_group) == 'nccl'
    assert forced.collective and len(forced.plans_u) == 4 and len(forced.plans_i) == 2
    assert not forced._host_bounce                 # the production branch, not the rehearsal one
    assert torch.equal(forced.U, plain.U) and torch.equal(forced.V, plain.V)
    # per-slab error partials are added in another order than the single plan's: equal to
    # rounding of the fp64 sum
    np.testing.assert_allclose(sse_forced, sse_plain, rtol=1e-12)
    assert forced.nnz == plain.nnz == nnz
    # 4 half-sweeps with gathers each on both sides were timed
    assert len(forced._gather_stamps) == 8
    assert forced.gather_exposed_ms() >= 0.0
    # the whole-table gather used when the factors are initialised
    before = forced.V.clone()
    forced._all_gather_rows(forced.V, forced.item_bounds)
    assert torch.equal(forced.V, before)


def test_estimators_run_on_a_forced_one_rank_group(rccl_group):
    """ALS.fit and IMC pick the group up by themselves (default_group) and give the results of
    the plain path."""
    from alsmi import ALS
    from alsmi.engine import default_group
    assert default_group() is rccl_group
    rng = np.random.default_rng(5)
    n, m, nnz = 300, 200, 6000
    flat = rng.choice(n * m, size=nnz, replace=False)
    users, items = flat // m, flat % m
    vals = rng.integers(1, 6, nnz).astype(np.float64)
    a = ALS(rank=8, random_state=2, max_iter=4).fit((users, items), vals, shape=(n, m))
    os.environ['ALSMI_FORCE_COLLECTIVES'] = '0'
    try:
        assert default_group() is None
        b = ALS(rank=8, random_state=2, max_iter=4).fit((users, items), vals, shape=(n, m))
    finally:
        os.environ['ALSMI_FORCE_COLLECTIVES'] = '1'
    assert np.array_equal(a.user_feats, b.user_feats)
    assert np.array_equal(a.item_feats, b.item_feats)
    np.testing.assert_allclose(a.rmse_history_, b.rmse_history_, rtol=1e-12)


def test_fit_sharded_equals_fit_on_rccl_and_without_a_group(rccl_group):
    """ALS.fit_sharded (every rank brings a part of the triplets; the entries travel by
    all-to-all to the owners of their user and of their item) against ALS.fit on the same
    ratings: through the forced one-rank RCCL group, and with no group at all."""
    import torch
    import torch.distributed as dist
    from alsmi import ALS
    from alsmi import device as dev
    from alsmi.engine import AlsEngine, default_group
    device = dev.require_gpu()
    rng = np.random.default_rng(8)
    n, m, nnz = 900, 400, 30000
    flat = rng.permutation(n * m)[:nnz]                  # distinct pairs in no particular order
    users, items = flat // m, flat % m
    vals = rng.integers(1, 11, nnz).astype(np.float64) / 2
    ref = ALS(rank=12, random_state=4, max_iter=5, dtype='float64').fit((users, items), vals,
                                                                        shape=(n, m))
    assert default_group() is rccl_group and dist.get_backend(rccl_group) == 'nccl'
    a = ALS(rank=12, random_state=4, max_iter=5, dtype='float64').fit_sharded(
        users, items, vals, (n, m))
    os.environ['ALSMI_FORCE_COLLECTIVES'] = '0'
    try:
        b = ALS(rank=12, random_state=4, max_iter=5, dtype='float64').fit_sharded(
            torch.from_numpy(users).to(device), torch.from_numpy(items).to(device),
            torch.from_numpy(vals).to(device), (n, m))
    finally:
        os.environ['ALSMI_FORCE_COLLECTIVES'] = '1'
    for est in (a, b):
        assert np.array_equal(est.user_feats, ref.user_feats)
        assert np.array_equal(est.item_feats, ref.item_feats)
        np.testing.assert_allclose(est.rmse_history_, ref.rmse_history_, rtol=1e-12)
        np.testing.assert_allclose(est.reconstruction_err_, ref.reconstruction_err_, rtol=1e-12)
        np.testing.assert_allclose(est._predict((users[:50], items[:50])),
                                   ref._predict((users[:50], items[:50])), rtol=1e-13)
    # the slices the exchange builds are those of the device CSR / CSC builders
    eng = AlsEngine.from_shards(torch.from_numpy(users).to(device),
                                torch.from_numpy(items).to(device),
                                torch.from_numpy(vals).to(device), (n, m), 12, 0.1,
                                group=rccl_group)
    csr = dev.build_csr(users, items, vals, (n, m), np.float64, device)
    csc = dev.csr_to_csc(csr[0], csr[1], csr[2], (n, m))
    for side, want in ((eng.csr, csr), (eng.csc, csc)):
        assert np.array_equal(side.indptr_host, want[0].cpu().numpy())
        assert torch.equal(side.indices, want[1]) and torch.equal(side.vals, want[2])
    with pytest.raises(ValueError):
        ALS(rank=4, max_iter=1).fit_sharded(users[:10], items[:10], np.zeros(10), (n, m))


def test_dimc_fit_sharded_equals_fit(rccl_group):
    """DirtyIMC.fit_sharded (src/dimc.py:145-162 with every rank bringing its own samples:
    IMC evaluations all-reduced, the residual's entries exchanged into the ALS row ranges)
    against DirtyIMC.fit on the same samples, through the forced one-rank RCCL group."""
    import warnings
    from alsmi import DirtyIMC
    rng = np.random.default_rng(21)
    n, m, p, q, ns = 400, 300, 6, 9, 12000
    flat = rng.permutation(n * m)[:ns]
    users, items = flat // m, flat % m
    X, Y = rng.standard_normal((n, p)), rng.standard_normal((m, q))
    Zs = rng.standard_normal((p, q)) * 0.3
    P8, Q8 = rng.standard_normal((n, 3)) * 0.5, rng.standard_normal((m, 3)) * 0.5
    vals = np.einsum('ij,ij->i', X[users].dot(Zs), Y[items]) + \
        np.einsum('ij,ij->i', P8[users], Q8[items]) + rng.standard_normal(ns) * 0.05
    kw = dict(rank=4, tol=1e-4, random_state=3, imc_alpha=1e-6, als_alpha=0.05)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        ref = DirtyIMC(**kw).fit((X, Y, users, items), vals, shape=(n, m))
        got = DirtyIMC(**kw).fit_sharded((X, Y, users, items), vals, (n, m))
    # (the IMC stage sums over the users in another slicing once the users without samples are
    # dropped, so the optimiser's iterates agree to rounding, not bit for bit)
    np.testing.assert_allclose(got.imc.Z, ref.imc.Z, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(got.imc.reconstruction_err_, ref.imc.reconstruction_err_, rtol=1e-8)
    assert got.als.n_iter_ == ref.als.n_iter_
    np.testing.assert_allclose(got.als.user_feats, ref.als.user_feats, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(got.als.item_feats, ref.als.item_feats, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(got.reconstruction_err_, ref.reconstruction_err_, rtol=1e-6)
    sel = rng.choice(ns, 500, replace=False)
    sub = (X, Y, users[sel], items[sel])
    np.testing.assert_allclose(got.predict(sub), ref.predict(sub), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(got.score(sub, vals[sel]), ref.score(sub, vals[sel]), rtol=1e-5)
