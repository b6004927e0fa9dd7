#!/usr/bin/env python3
"""Generate golden input/output vectors by running the upstream Python reference.

Run from any directory OUTSIDE the reference checkout (the reference's
``ALS.update_user`` writes files into the cwd, ``src/als.py:412-413``)::

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python3 /root/repo/tests/golden/make_golden.py

The reference is imported from ``/root/reference`` (read-only) and never
copied.  Only DATA (inputs and the reference's outputs) is written, as small
``.npz`` files next to this script.  The reference is not available on the
GPU box; the committed fixtures are what the tests there compare against.

One process-local compatibility patch is applied before fitting:
``src/als.py:268,274`` call ``scipy.sparse.sputils.get_index_dtype/upcast``
which recent scipy only keeps in the private ``_sputils`` module.
"""
import os
import sys
import warnings

import numpy as np
import scipy.sparse as sps

REF = os.environ.get('ALSMI_REFERENCE', '/root/reference')
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)

import scipy.sparse._sputils as _sputils  # noqa: E402

if not hasattr(sps, 'sputils'):
    import types
    sps.sputils = types.ModuleType('sputils')
for _name in ('get_index_dtype', 'upcast'):
    if not hasattr(sps.sputils, _name):
        setattr(sps.sputils, _name, getattr(_sputils, _name))

from src import ALS, IMC, DirtyIMC  # noqa: E402
from src.imc import _cost, _cost_hess, _cost_prime, _fit_imc  # noqa: E402
from src.utils import root_mean_squared_error  # noqa: E402


def save(name, **arrays):
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **arrays)
    print('wrote %s (%.1f KB)' % (path, os.path.getsize(path) / 1024))


def random_ratings(seed, n, m, nnz, levels=(1, 2, 3, 4, 5)):
    """nnz distinct positions, ratings drawn from `levels`, CSR order."""
    rng = np.random.default_rng(seed)
    flat = np.sort(rng.choice(n * m, size=nnz, replace=False))
    users = (flat // m).astype(np.int32)
    items = (flat % m).astype(np.int32)
    vals = rng.choice(np.asarray(levels, dtype=np.float64), size=nnz)
    return users, items, vals


def als_replay(users, items, vals, shape, rank, alpha, seed, sweeps):
    """Replay ALS.fit_transform's preamble (src/als.py:143-152) and run a
    fixed number of sweeps exactly the way its loop body does (:155-165)."""
    from sklearn.utils.validation import check_random_state
    als = ALS(rank=rank, alpha=alpha, random_state=seed)
    data = sps.csr_matrix((vals, (users, items)), shape=shape)
    random_state = check_random_state(seed)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        item_avg = data.sum(0) / (data != 0).sum(0)
    item_avg[np.isnan(item_avg)] = 0
    als.item_feats = random_state.rand(rank, data.shape[1])
    als.item_feats[0] = item_avg
    als.user_feats = np.zeros((rank, data.shape[0]))
    als.data = data
    init_item = als.item_feats.copy()
    u_hist, i_hist, rmse_hist = [], [], []
    for _ in range(sweeps):
        als._update_parallel(np.array_split(np.arange(shape[0]), 1))
        als._update_parallel(np.array_split(np.arange(shape[1]), 1),
                             user=False)
        uu, ii = data.nonzero()
        pred = (als.user_feats.T[uu] * als.item_feats.T[ii]).sum(-1)
        rmse_hist.append(root_mean_squared_error(data.data, pred))
        u_hist.append(als.user_feats.copy())
        i_hist.append(als.item_feats.copy())
    return init_item, u_hist, i_hist, np.array(rmse_hist), als


def gen_als_unit():
    """The 2x3 matrix of tests/test_als.py:19-26 through the public API."""
    R = sps.csr_matrix(np.array([[3, 0, 4], [4, 5, 0]]))
    users, items = R.nonzero()
    als = ALS(2, tol=0.000001, random_state=42)
    sweeps = [0]
    orig = als._update_parallel

    def counting(arrays, user=True):
        sweeps[0] += 1
        return orig(arrays, user=user)
    als._update_parallel = counting
    u, i = als.fit_transform((users, items), R)
    als2 = ALS(2, tol=0.000001, random_state=42)
    u2, i2 = als2.fit_transform((users, items), R.data, shape=R.shape)
    save('als_unit',
         R=R.toarray(), users=users, items=items, r=R.data,
         user_feats=u, item_feats=i, user_feats_dense_y=u2,
         item_feats_dense_y=i2,
         reconstruction_err=np.float64(als.reconstruction_err_),
         half_sweeps=np.int64(sweeps[0]),
         predict=als._predict((users, items)),
         predict_one_0_1=als.predict_one(0, 1),
         predict_all_0=als.predict_all(0),
         score=np.float64(als.score((users, items), R.data)),
         score_sparse_y=np.float64(als.score((users, items), R)))


def gen_als_case(name, seed, n, m, nnz, rank, alpha, sweeps, keep,
                 levels=(1, 2, 3, 4, 5), drop_rows=(), drop_cols=()):
    users, items, vals = random_ratings(seed, n, m, nnz, levels)
    if len(drop_rows) or len(drop_cols):
        mask = ~(np.isin(users, drop_rows) | np.isin(items, drop_cols))
        users, items, vals = users[mask], items[mask], vals[mask]
    init_item, u_hist, i_hist, rmse, als = als_replay(
        users, items, vals, (n, m), rank, alpha, 42, sweeps)
    out = dict(users=users, items=items, vals=vals,
               shape=np.array([n, m], dtype=np.int64),
               rank=np.int64(rank), alpha=np.float64(alpha),
               seed=np.int64(42), init_item_feats=init_item, rmse=rmse,
               keep=np.array(keep, dtype=np.int64))
    for s in keep:
        out['user_feats_%d' % s] = u_hist[s - 1]
        out['item_feats_%d' % s] = i_hist[s - 1]
    rng = np.random.default_rng(seed + 1)
    pu = rng.integers(0, n, size=257).astype(np.int32)
    pi = rng.integers(0, m, size=257).astype(np.int32)
    out['pair_users'] = pu
    out['pair_items'] = pi
    out['pair_pred'] = als._predict((pu, pi))
    out['score_train'] = np.float64(als.score((users, items), vals))
    save(name, **out)


def gen_als_fit_tol():
    """Public fit() with the tol-driven stopping rule (src/als.py:154-169)
    on a small random matrix; also records how many sweeps it ran."""
    n, m, nnz = 60, 40, 600
    users, items, vals = random_ratings(7, n, m, nnz)
    out = {}
    for tag, y, kw in (('sparse', sps.csr_matrix((vals, (users, items)),
                                                 shape=(n, m)), {}),
                       ('dense', vals, {'shape': (n, m)})):
        als = ALS(rank=5, alpha=0.1, tol=0.001, random_state=3)
        count = [0]
        orig = als._update_parallel

        def counting(arrays, user=True, _o=orig, _c=count):
            _c[0] += 1
            return _o(arrays, user=user)
        als._update_parallel = counting
        u, i = als.fit_transform((users, items), y, **kw)
        out['user_feats_' + tag] = u
        out['item_feats_' + tag] = i
        out['err_' + tag] = np.float64(als.reconstruction_err_)
        out['half_sweeps_' + tag] = np.int64(count[0])
    save('als_fit_tol', users=users, items=items, vals=vals,
         shape=np.array([n, m], dtype=np.int64), **out)


def imc_test_data():
    user_atts = np.array([
        [1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 0],
        [0, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 1, 0]])
    item_atts = np.array([
        [1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0],
        [0, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0]])
    R = sps.csr_matrix(np.array([[300, 6000], [9000, 12000]]))
    rows, cols = R.nonzero()
    return user_atts, item_atts, R, rows, cols


def gen_imc_unit():
    """tests/test_imc.py:33-47 data, per-sample inputs (the form the current
    IMC.fit_transform accepts, src/imc.py:241-242)."""
    X, Y, R, rows, cols = imc_test_data()
    x_h, y_h, b_h = X[rows], Y[cols], R.data
    args = (x_h, y_h, b_h, 0.01, (15, 20))
    zeros = np.zeros(300)
    hess_p = np.arange(300)
    out = dict(X=X, Y=Y, R=R.toarray(), rows=rows, cols=cols,
               cost0=np.float64(_cost(zeros, *args)),
               grad0=_cost_prime(zeros, *args),
               hess0=_cost_hess(zeros, hess_p, *args))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        Zd, succ, msg = _fit_imc(x_h, y_h, b_h)
        out['fit_default_Z'] = Zd
        out['fit_default_success'] = np.bool_(succ)
        imc0 = IMC(alpha=0.01)
        Z0 = imc0.fit_transform((x_h, y_h), b_h)
        out['imc0_Z'] = Z0
        out['imc0_err'] = np.float64(imc0.reconstruction_err_)
        out['imc0_score'] = np.float64(imc0.score((x_h, y_h), b_h))
        out['imc0_predict_one'] = np.float64(imc0.predict_one((X[0], Y[0])))
        out['imc0_predict_all'] = imc0.predict_all((X, Y))
        imck = IMC(n_components=2)
        out['imck_Z'] = imck.fit_transform((x_h, y_h), b_h)
        out['imck_n_components_'] = np.int64(imck.n_components_)
        imcb = IMC(alpha=0.01, method='BFGS')
        out['imcb_Z'] = imcb.fit_transform((x_h, y_h), b_h)
        out['imcb_err'] = np.float64(imcb.reconstruction_err_)
    save('imc_unit', **out)


def gen_imc_random():
    """cost / gradient / Hessian-vector on random dense inputs, both as
    per-sample matrices and as the index form they were built from."""
    rng = np.random.default_rng(11)
    n, m, p, q, nnz = 50, 30, 12, 20, 400
    users, items, vals = random_ratings(12, n, m, nnz,
                                        levels=(0.5, 1.5, 2.5, 3.5, 5.0))
    X = rng.standard_normal((n, p)) / np.sqrt(p)
    Y = rng.standard_normal((m, q)) / np.sqrt(q)
    Z = rng.standard_normal((p, q))
    S = rng.standard_normal((p, q))
    lam = 0.1
    args = (X[users], Y[items], vals, lam, (p, q))
    out = dict(X=X, Y=Y, users=users, items=items, vals=vals, Z=Z, S=S,
               lam=np.float64(lam),
               cost=np.float64(_cost(Z.ravel(), *args)),
               grad=_cost_prime(Z.ravel(), *args),
               hessp=_cost_hess(Z.ravel(), S.ravel(), *args))
    # init Z0 of src/imc.py:156-159
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        imc = IMC(alpha=lam)
        Zfit = imc.fit_transform((X[users], Y[items]), vals)
        out['fit_Z'] = Zfit
        out['fit_err'] = np.float64(imc.reconstruction_err_)
        from scipy.sparse import diags
        out['Z0'] = diags(vals).T.dot(X[users]).T.dot(Y[items]) / vals.size
    save('imc_random', **out)


def gen_dimc():
    """DirtyIMC end to end (src/dimc.py:116-162) on the IMC test data and on
    a small random problem; sparse-y and 1-D-y+shape entry points."""
    X, Y, R, rows, cols = imc_test_data()
    x_h, y_h = X[rows], Y[cols]
    out = dict(X=X, Y=Y, R=R.toarray(), rows=rows, cols=cols)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for tag, y, kw in (('sparse', R, {}),
                           ('dense', R.data, {'shape': R.shape})):
            d = DirtyIMC(rank=2, tol=1e-6, random_state=42, imc_alpha=0.01)
            Z, u, i = d.fit_transform((x_h, y_h, rows, cols), y, **kw)
            out['Z_' + tag] = Z
            out['user_feats_' + tag] = u
            out['item_feats_' + tag] = i
            out['err_' + tag] = np.float64(d.reconstruction_err_)
            out['predict_' + tag] = d.predict((x_h, y_h, rows, cols))
        rng = np.random.default_rng(21)
        n, m, p, q, nnz = 40, 25, 6, 8, 300
        users, items, _ = random_ratings(22, n, m, nnz)
        Xa = rng.standard_normal((n, p)) / np.sqrt(p)
        Ya = rng.standard_normal((m, q)) / np.sqrt(q)
        Zs = rng.standard_normal((p, q))
        pu = rng.standard_normal((n, 3))
        qi = rng.standard_normal((m, 3))
        vals = ((Xa[users].dot(Zs) * Ya[items]).sum(-1) +
                0.5 * (pu[users] * qi[items]).sum(-1) +
                0.1 * rng.standard_normal(nnz))
        d = DirtyIMC(rank=3, tol=1e-3, random_state=5, imc_alpha=0.1,
                     als_alpha=0.1)
        Z, u, i = d.fit_transform((Xa[users], Ya[items], users, items), vals,
                                  shape=(n, m))
        out.update(rnd_X=Xa, rnd_Y=Ya, rnd_users=users, rnd_items=items,
                   rnd_vals=vals, rnd_shape=np.array([n, m], dtype=np.int64),
                   rnd_Z=Z, rnd_user_feats=u, rnd_item_feats=i,
                   rnd_err=np.float64(d.reconstruction_err_),
                   rnd_predict=d.predict((Xa[users], Ya[items], users, items)))
    save('dimc', **out)


def main():
    gen_als_unit()
    # BASELINE config 1: 1000x500, 5 % density, rank 10, alpha 0.1, 10 sweeps
    gen_als_case('als_c1', 20241, 1000, 500, 25000, 10, 0.1, 10, (1, 2, 10))
    # rank 64 with most rows shorter than the rank, empty rows and columns
    gen_als_case('als_k64', 20242, 300, 200, 6000, 64, 0.1, 2, (1, 2),
                 levels=(0.5, 1.0, 2.5, 4.0, 5.0),
                 drop_rows=(0, 17, 299), drop_cols=(3, 199))
    # odd rank, tiny alpha, one very long row
    gen_als_case('als_k3', 20243, 120, 90, 1500, 3, 0.01, 3, (1, 3))
    gen_als_fit_tol()
    gen_imc_unit()
    gen_imc_random()
    gen_dimc()


if __name__ == '__main__':
    main()
