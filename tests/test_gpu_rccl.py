"""The multi-rank code path on RCCL with the one GPU at hand: a process group of ONE rank
whose collectives are forced on (ALSMI_FORCE_COLLECTIVES=1), so that every broadcast of the
slab gathers, the all-reduce of the squared error and the exchange of the slab bounds is a
real RCCL call, asynchronous beside the kernels exactly as with eight ranks.  The factors must
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


@pytest.mark.parametrize('dtype', ['float32', 'float64'])
def test_forced_gathers_on_rccl_match_the_plain_path(rccl_group, dtype):
    import torch
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
