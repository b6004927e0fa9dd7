"""The multi-rank path of alsmi.engine.AlsEngine (row sharding, all-gather of the
updated factor rows, all-reduce of the squared error) under gloo with 2 and 3 CPU
processes, the arithmetic supplied by the oracle backend."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from problems import random_problem

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def reference_run(P, rank, alpha, sweeps):
    from oracle import als_oracle as ao
    uf, itf = np.zeros_like(P['U']), P['V'].copy()
    hist = []
    for _ in range(sweeps):
        uf = ao.half_sweep(P['indptr'], P['indices'], P['vals'], itf, rank, alpha)
        itf = ao.half_sweep(P['cptr'], P['cidx'], P['cvals'], uf, rank, alpha)
        hist.append(ao.train_sse(P['indptr'], P['indices'], P['vals'], uf, itf))
    return uf, itf, hist


def worker(rank_id, world, port, case, out_dir):
    for p in (os.path.dirname(HERE), os.path.join(os.path.dirname(HERE),
                                                  'recommendation-models_amd'), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from alsmi.engine import AlsEngine
    from oracle_backend import OracleBackend
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank_id, world_size=world)
    try:
        seed, n, m, nnz, rank, extra = case
        P = random_problem(seed, n, m, nnz, rank, **extra)
        eng = AlsEngine.from_global(P['indptr'], P['indices'], P['vals'], (n, m), rank, 0.1,
                                    group=dist.group.WORLD, backend=OracleBackend(),
                                    user_slabs=4, item_slabs=2)
        assert eng.nnz == int(P['indptr'][-1])
        eng.set_factors(np.zeros_like(P['U']), P['V'])
        hist = []
        eng.half_sweep_users()
        eng.half_sweep_items()
        hist.append(float(eng.train_sse_async()[0]))
        # second sweep on the fused schedule: its user half reports the error of sweep 1
        # and parks the new rows until they are committed
        before = eng.U.clone()
        fused = float(eng.half_sweep_users_err(keep_old=True)[0])
        assert abs(fused - hist[0]) <= 1e-10 * abs(hist[0])
        assert torch.equal(eng.U, before)
        eng.commit_users()
        eng.half_sweep_items()
        hist.append(float(eng.train_sse_async()[0]))
        # a parked half-sweep that is dropped leaves the factors alone
        before = eng.U.clone()
        again = float(eng.half_sweep_users_err(keep_old=True)[0])
        eng.discard_users()
        assert abs(again - hist[1]) <= 1e-10 * abs(hist[1])
        assert torch.equal(eng.U, before)
        # third sweep in place (the schedule of bench.py and of ALS.fit(max_iter=...)): the
        # user half runs slab by slab with the slabs' all-gathers in flight beside it
        assert len(eng.plans_u) == 4
        inplace = float(eng.half_sweep_users_err()[0])
        assert abs(inplace - hist[1]) <= 1e-10 * abs(hist[1])
        eng.half_sweep_items()
        hist.append(float(eng.train_sse_async()[0]))
        uf, itf = eng.get_factors()
        np.savez(os.path.join(out_dir, 'rank%d.npz' % rank_id), uf=uf, itf=itf, hist=hist,
                 ub=eng.user_bounds, ib=eng.item_bounds)
    finally:
        dist.destroy_process_group()


CASES = [
    (51, 40, 30, 500, 4, {}),
    # skewed: one very long user row makes the entry-balanced ranges uneven in rows
    (52, 37, 90, 700, 10, dict(long_row=(3, 80), empty_rows=(0, 36))),
]


@pytest.mark.parametrize('mode', ['p2p', 'bcast'])
@pytest.mark.parametrize('world', [2, 3])
@pytest.mark.parametrize('case', CASES, ids=['even', 'skewed'])
def test_sharded_sweeps_match_single_process(tmp_path, world, case, mode, monkeypatch):
    """Both forms of the row gather: the batch of pairwise transfers and the broadcast per
    owning rank (the spawned ranks inherit ALSMI_GATHER)."""
    monkeypatch.setenv('ALSMI_GATHER', mode)
    port = free_port()
    mp.spawn(worker, args=(world, port, case, str(tmp_path)), nprocs=world, join=True)
    seed, n, m, nnz, rank, extra = case
    P = random_problem(seed, n, m, nnz, rank, **extra)
    uf, itf, hist = reference_run(P, rank, 0.1, 3)
    outs = [np.load(os.path.join(str(tmp_path), 'rank%d.npz' % r)) for r in range(world)]
    for o in outs:
        # every rank ends with the full, identical factor tables
        np.testing.assert_allclose(o['uf'], uf, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(o['itf'], itf, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(o['hist'], hist, rtol=1e-10)
        assert np.array_equal(o['uf'], outs[0]['uf']) and np.array_equal(o['itf'], outs[0]['itf'])
        assert o['ub'][0] == 0 and o['ub'][-1] == n and np.all(np.diff(o['ub']) >= 0)
        assert o['ib'][0] == 0 and o['ib'][-1] == m and np.all(np.diff(o['ib']) >= 0)


def shard_worker(rank_id, world, port, case, out_dir):
    for p in (os.path.dirname(HERE), os.path.join(os.path.dirname(HERE),
                                                  'recommendation-models_amd'), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from alsmi.engine import AlsEngine
    from oracle_backend import OracleBackend
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank_id, world_size=world)
    try:
        seed, n, m, nnz, rank, extra = case
        P = random_problem(seed, n, m, nnz, rank, **extra)
        # this rank's share of the triplets: a shuffled deal, unrelated to the row ranges
        owner = np.random.default_rng(seed).permutation(P['users'].size) % world
        if world > 2:
            owner[owner == world - 1] = 0       # with 3 ranks the last one brings nothing
        deal = owner == rank_id
        mine = [torch.from_numpy(np.ascontiguousarray(P[k][deal])) for k in
                ('users', 'items', 'vals')]
        eng = AlsEngine.from_shards(mine[0], mine[1], mine[2], (n, m), rank, 0.1,
                                    group=dist.group.WORLD, backend=OracleBackend())
        ref = AlsEngine.from_global(P['indptr'], P['indices'], P['vals'], (n, m), rank, 0.1,
                                    group=dist.group.WORLD, backend=OracleBackend())
        # the same row ranges and the same local slices as the cut of the whole matrix
        assert np.array_equal(eng.user_bounds, ref.user_bounds)
        assert np.array_equal(eng.item_bounds, ref.item_bounds)
        assert eng.nnz == ref.nnz == int(P['indptr'][-1])
        for a, b in ((eng.csr, ref.csr), (eng.csc, ref.csc)):
            assert np.array_equal(a.indptr_host, b.indptr_host)
            assert np.array_equal(a.indices_np, b.indices_np)
            assert np.array_equal(a.vals_np, b.vals_np)
        means = eng.item_means().numpy()
        cnt = np.bincount(P['items'], minlength=m)
        want = np.where(cnt > 0, np.bincount(P['items'], weights=P['vals'], minlength=m) /
                        np.maximum(cnt, 1), 0.0)
        np.testing.assert_allclose(means, want, rtol=1e-13)
        eng.set_factors(np.zeros_like(P['U']), P['V'])
        for _ in range(2):
            eng.sweep()
        uf, itf = eng.get_factors()
        np.savez(os.path.join(out_dir, 'shard%d.npz' % rank_id), uf=uf, itf=itf,
                 n_mine=int(deal.sum()))
        # a pair stored twice (on two ranks) is refused by every rank that receives it
        if world == 2:
            dup = [torch.cat([t, t[:1]]) if rank_id == 0 else
                   torch.from_numpy(np.ascontiguousarray(P[k][deal | ~deal][:0])) for t, k in
                   zip(mine, ('users', 'items', 'vals'))]
            if rank_id == 1:    # the other rank holds every entry once more
                dup = [torch.from_numpy(np.ascontiguousarray(P[k])) for k in
                       ('users', 'items', 'vals')]
            try:
                AlsEngine.from_shards(dup[0], dup[1], dup[2], (n, m), rank, 0.1,
                                      group=dist.group.WORLD, backend=OracleBackend())
                raised = False
            except ValueError:
                raised = True
            flag = torch.tensor([int(raised)])
            dist.all_reduce(flag)
            assert int(flag[0]) >= 1
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_ingest_builds_the_same_slices(tmp_path, world):
    case = CASES[1]
    port = free_port()
    mp.spawn(shard_worker, args=(world, port, case, str(tmp_path)), nprocs=world, join=True)
    seed, n, m, nnz, rank, extra = case
    P = random_problem(seed, n, m, nnz, rank, **extra)
    uf, itf, _ = reference_run(P, rank, 0.1, 2)
    outs = [np.load(os.path.join(str(tmp_path), 'shard%d.npz' % r)) for r in range(world)]
    assert sum(int(o['n_mine']) for o in outs) == P['users'].size
    for o in outs:
        np.testing.assert_allclose(o['uf'], uf, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(o['itf'], itf, rtol=1e-9, atol=1e-12)


def test_split_by_nnz():
    from alsmi.engine import split_by_nnz
    indptr = np.array([0, 0, 10, 10, 11, 40, 40], dtype=np.int64)
    b = split_by_nnz(indptr, 2)
    assert b[0] == 0 and b[-1] == 6 and len(b) == 3
    assert np.all(np.diff(split_by_nnz(indptr, 4)) >= 0)
    assert list(split_by_nnz(indptr, 1)) == [0, 6]
    # more parts than rows still gives monotone bounds ending at n
    b = split_by_nnz(np.array([0, 5], dtype=np.int64), 3)
    assert b[0] == 0 and b[-1] == 1 and np.all(np.diff(b) >= 0)


# ---- IMC on several ranks: samples sharded by user range, all-reduced evaluations -----------

def imc_worker(rank_id, world, port, out_dir):
    for p in (os.path.dirname(HERE), os.path.join(os.path.dirname(HERE),
                                                  'recommendation-models_amd'), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import warnings
    import torch.distributed as dist
    import alsmi.imc as imc_mod
    from oracle_backend import make_oracle_imc_problem
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank_id, world_size=world)
    try:
        Problem = make_oracle_imc_problem()
        imc_mod.ImcProblem = Problem          # the estimator builds its problems from this name
        X, Y, users, items, r, Z, S = imc_case()
        prob = Problem(X, Y, users, items, r, group=dist.group.WORLD)
        assert 0 < prob.n_local_samples < r.size     # really sharded
        res = dict(cost=prob.cost(Z, 0.3), grad=prob.grad(Z, 0.3), hessp=prob.hessp(S, 0.3),
                   z0=prob.init_z(), pred=prob.predict_samples(Z), n_local=prob.n_local_samples)
        # the estimator end to end on the sharded problem: fit, training error, score
        import scipy.sparse as sps
        R = sps.csr_matrix((r, (users, items)), shape=(X.shape[0], Y.shape[0]))
        est = imc_mod.IMC(alpha=0.3)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            est.fit((X, Y), R)
        res.update(fit_Z=est.Z, fit_err=est.reconstruction_err_, score=est.score((X, Y), R))
        np.savez(os.path.join(out_dir, 'imc%d.npz' % rank_id), **res)
    finally:
        dist.destroy_process_group()


def imc_local_worker(rank_id, world, port, out_dir):
    for p in (os.path.dirname(HERE), os.path.join(os.path.dirname(HERE),
                                                  'recommendation-models_amd'), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import warnings
    import torch.distributed as dist
    import alsmi.imc as imc_mod
    from oracle_backend import make_oracle_imc_problem
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank_id, world_size=world)
    try:
        Problem = make_oracle_imc_problem()
        imc_mod.ImcProblem = Problem
        X, Y, users, items, r, Z, S = imc_case()
        # this rank's own samples: a shuffled deal (users end up split between ranks); with
        # three ranks the last one brings none
        owner = np.random.default_rng(3).permutation(r.size) % world
        if world > 2:
            owner[owner == world - 1] = 0
        mine = owner == rank_id
        prob = Problem.from_local_samples(X, Y, users[mine], items[mine], r[mine],
                                          group=dist.group.WORLD)
        assert prob.n_samples == r.size and prob.n_local_samples == int(mine.sum())
        res = dict(cost=prob.cost(Z, 0.3), grad=prob.grad(Z, 0.3), hessp=prob.hessp(S, 0.3),
                   z0=prob.init_z(), pred=prob.predict_samples(Z), mine=mine)
        est = imc_mod.IMC(alpha=0.3)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            est.fit_sharded((X, Y), users[mine], items[mine], r[mine])
        res.update(fit_Z=est.Z, fit_err=est.reconstruction_err_)
        np.savez(os.path.join(out_dir, 'imcl%d.npz' % rank_id), **res)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_imc_on_each_ranks_own_samples(tmp_path, world):
    import warnings
    from oracle import imc_oracle as io
    port = free_port()
    mp.spawn(imc_local_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    X, Y, users, items, r, Z, S = imc_case()
    outs = [np.load(os.path.join(str(tmp_path), 'imcl%d.npz' % k)) for k in range(world)]
    assert sum(int(o['mine'].sum()) for o in outs) == r.size
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fit_Z, _, _ = io.fit_imc(X[users], Y[items], r, alpha=0.3)
    fit_err = io.imc_score(fit_Z, X[users], Y[items], r)
    preds = io.preds_indexed(Z, X, Y, users, items)
    for o in outs:
        np.testing.assert_allclose(o['cost'], io.cost_indexed(Z, X, Y, users, items, r, 0.3),
                                   rtol=1e-12)
        np.testing.assert_allclose(o['grad'], io.grad_indexed(Z, X, Y, users, items, r, 0.3),
                                   rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(o['hessp'], io.hessp_indexed(S, X, Y, users, items, 0.3),
                                   rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(o['z0'], io.init_z(X, Y, users, items, r), rtol=1e-9,
                                   atol=1e-12)
        # a rank predicts its own samples, in the order it gave them
        np.testing.assert_allclose(o['pred'], preds[o['mine']], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(o['fit_Z'], fit_Z, rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(o['fit_err'], fit_err, rtol=1e-6)
        assert np.array_equal(o['fit_Z'], outs[0]['fit_Z'])


def imc_case():
    rng = np.random.default_rng(17)
    n, m, p, q, ns = 60, 45, 7, 19, 700
    flat = rng.choice(n * m, size=ns, replace=False)
    users, items = flat // m, flat % m
    users[users == 4] = 5                      # a user without samples
    keep = np.unique(users * m + items, return_index=True)[1]
    users, items = users[keep], items[keep]
    X, Y = rng.standard_normal((n, p)), rng.standard_normal((m, q))
    return (X, Y, users, items, rng.standard_normal(users.size), rng.standard_normal((p, q)),
            rng.standard_normal((p, q)))


@pytest.mark.parametrize('world', [2, 3])
def test_imc_sharded_evaluations_match_single_process(tmp_path, world):
    import warnings
    from oracle import imc_oracle as io
    port = free_port()
    mp.spawn(imc_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    X, Y, users, items, r, Z, S = imc_case()
    outs = [np.load(os.path.join(str(tmp_path), 'imc%d.npz' % k)) for k in range(world)]
    assert sum(int(o['n_local']) for o in outs) == r.size
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fit_Z, _, _ = io.fit_imc(X[users], Y[items], r, alpha=0.3)
    fit_err = io.imc_score(fit_Z, X[users], Y[items], r)
    for o in outs:
        np.testing.assert_allclose(o['cost'], io.cost_indexed(Z, X, Y, users, items, r, 0.3),
                                   rtol=1e-12)
        np.testing.assert_allclose(o['grad'], io.grad_indexed(Z, X, Y, users, items, r, 0.3),
                                   rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(o['hessp'], io.hessp_indexed(S, X, Y, users, items, 0.3),
                                   rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(o['z0'], io.init_z(X, Y, users, items, r), rtol=1e-9,
                                   atol=1e-12)
        np.testing.assert_allclose(o['pred'], io.preds_indexed(Z, X, Y, users, items),
                                   rtol=1e-11, atol=1e-12)
        # every rank drives the same optimiser trajectory and scores all samples
        np.testing.assert_allclose(o['fit_Z'], fit_Z, rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(o['fit_err'], fit_err, rtol=1e-6)
        np.testing.assert_allclose(o['score'], fit_err, rtol=1e-6)
        assert np.array_equal(o['fit_Z'], outs[0]['fit_Z'])
