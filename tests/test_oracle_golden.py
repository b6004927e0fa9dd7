"""The numpy oracle against vectors produced by the upstream reference
(tests/golden/make_golden.py) and the reference's own known-answer tests."""
import numpy as np
import pytest

from oracle import als_oracle as ao
from oracle import imc_oracle as io


def test_als_unit_known_answers(golden):
    # reference tests/test_als.py:46-56, :89, :97, :105, :113
    g = golden('als_unit')
    expected_u = np.array([[1.52198247, 1.97980245], [-0.79543119, -1.03470076]])
    expected_i = np.array([[1.5339228, 1.94471076, 1.99660463],
                           [-0.80167146, -1.01636085, -1.04348219]])
    uf, itf, hist, _ = ao.als_fit(g['users'], g['items'], g['r'], (2, 3), rank=2,
                                  alpha=0.1, tol=0.000001, seed=42)
    np.testing.assert_allclose(uf, expected_u, rtol=1e-7)
    np.testing.assert_allclose(itf, expected_i, rtol=1e-7)
    np.testing.assert_allclose(uf, g['user_feats'], rtol=1e-10)
    np.testing.assert_allclose(itf, g['item_feats'], rtol=1e-10)
    assert 2 * hist.size == int(g['half_sweeps'])
    pred = ao.predict_pairs(uf, itf, g['users'], g['items'])
    np.testing.assert_allclose(pred, [2.9722781, 3.86881553, 3.86635418, 4.90177248], rtol=1e-7)
    np.testing.assert_allclose(-ao.rmse(g['r'], pred), -0.10663924910933506, rtol=1e-10)
    np.testing.assert_allclose(ao.predict_pairs(uf, itf, [0], [1]), 3.768260826602256, rtol=1e-10)
    np.testing.assert_allclose(ao.predict_pairs(uf, itf, [0, 0, 0], [0, 1, 2]),
                               [2.972278, 3.76826083, 3.86881553], rtol=1e-7)


@pytest.mark.parametrize('name', ['als_c1', 'als_k64', 'als_k3'])
def test_als_sweeps_match_reference(golden, name):
    g = golden(name)
    n, m = (int(v) for v in g['shape'])
    rank, alpha = int(g['rank']), float(g['alpha'])
    keep = [int(s) for s in g['keep']]
    hist = []
    uf, itf, rm, csr = ao.als_fit(g['users'], g['items'], g['vals'], (n, m), rank=rank,
                                  alpha=alpha, seed=int(g['seed']), max_sweeps=max(keep),
                                  history=hist)
    np.testing.assert_allclose(rm, g['rmse'][:max(keep)], rtol=1e-11)
    for s in keep:
        np.testing.assert_allclose(hist[s - 1][0], g['user_feats_%d' % s], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(hist[s - 1][1], g['item_feats_%d' % s], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(ao.predict_pairs(uf, itf, g['pair_users'], g['pair_items']),
                               g['pair_pred'], rtol=1e-9, atol=1e-12)
    # sparse indexing is bit-exact: CSR of the triplets
    import scipy.sparse as sps
    ref = sps.csr_matrix((g['vals'], (g['users'], g['items'])), shape=(n, m))
    ref.sort_indices()
    assert np.array_equal(csr[0], ref.indptr)
    assert np.array_equal(csr[1], ref.indices)
    assert np.array_equal(csr[2], ref.data)
    csc = ref.tocsc()
    cptr, cidx, cval = ao.csr_to_csc(n, m, *csr)
    assert np.array_equal(cptr, csc.indptr)
    assert np.array_equal(cidx, csc.indices)
    assert np.array_equal(cval, csc.data)


def test_als_init_matches_reference(golden):
    g = golden('als_c1')
    n, m = (int(v) for v in g['shape'])
    csr = ao.build_csr(g['users'], g['items'], g['vals'], (n, m))
    uf, itf = ao.init_factors(n, m, *csr, int(g['rank']), int(g['seed']))
    assert np.array_equal(itf, g['init_item_feats'])
    assert not uf.any()


def test_als_tol_loop(golden):
    g = golden('als_fit_tol')
    n, m = (int(v) for v in g['shape'])
    uf, itf, hist, _ = ao.als_fit(g['users'], g['items'], g['vals'], (n, m), rank=5, alpha=0.1,
                                  tol=0.001, seed=3)
    assert 2 * hist.size == int(g['half_sweeps_sparse']) == int(g['half_sweeps_dense'])
    np.testing.assert_allclose(uf, g['user_feats_sparse'], rtol=1e-9)
    np.testing.assert_allclose(itf, g['item_feats_dense'], rtol=1e-9)
    np.testing.assert_allclose(-hist[-1], g['err_sparse'], rtol=1e-11)


def test_build_csr_lil_semantics():
    # duplicates: last write wins; a final 0 is not stored (src/als.py:137-140)
    users = np.array([0, 1, 0, 1, 2, 2])
    items = np.array([1, 0, 1, 2, 2, 2])
    y = np.array([5.0, 2.0, 3.0, 0.0, 4.0, 0.0])
    indptr, indices, vals = ao.build_csr(users, items, y, (3, 3))
    import scipy.sparse as sps
    ref = sps.lil_matrix((3, 3))
    for idx, (i, j) in enumerate(zip(users, items)):
        ref[i, j] = y[idx]
    ref = ref.tocsr()
    assert np.array_equal(indptr, ref.indptr)
    assert np.array_equal(indices, ref.indices)
    assert np.array_equal(vals, ref.data)


def test_update_one_singular_gives_zeros():
    out = ao.update_one(np.zeros((4, 0)), np.zeros(0), 4, 0.1)
    assert out.shape == (4,) and not out.any()


def imc_unit_args(g):
    x_h, y_h = g['X'][g['rows']], g['Y'][g['cols']]
    b_h = g['R'][g['rows'], g['cols']]
    return x_h, y_h, b_h


def test_imc_known_answers(golden):
    # reference tests/test_imc.py:53-80
    g = golden('imc_unit')
    x_h, y_h, b_h = imc_unit_args(g)
    args = (x_h, y_h, b_h, 0.01, (15, 20))
    assert io.cost(np.zeros(300), *args) == 130545000.0
    np.testing.assert_allclose(io.cost_prime(np.zeros(300), *args)[:10],
                               [-300, 0, 0, 0, -6000, 0, -300, 0, -6000, 0])
    np.testing.assert_allclose(io.cost_hess(np.zeros(300), np.arange(300), *args)[:10],
                               [1.54801e3, 0.01, 0.01, 0.01, 1.56001e3, 0.01, 1.54801e3, 0.01,
                                1.56001e3, 0.01])
    np.testing.assert_allclose(io.cost_prime(np.zeros(300), *args), g['grad0'], rtol=1e-13)
    np.testing.assert_allclose(io.cost_hess(np.zeros(300), np.arange(300), *args), g['hess0'],
                               rtol=1e-13)


def test_imc_fit_matches_reference(golden):
    g = golden('imc_unit')
    x_h, y_h, b_h = imc_unit_args(g)
    Z, succ, _ = io.fit_imc(x_h, y_h, b_h, alpha=0.01)
    np.testing.assert_allclose(Z, g['imc0_Z'], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(io.imc_score(Z, x_h, y_h, b_h), 5.748504798089005, rtol=1e-9)
    np.testing.assert_allclose(g['X'][0].dot(Z).dot(g['Y'][0]), 301.06449075427315, rtol=1e-9)
    np.testing.assert_allclose(g['X'].dot(Z).dot(g['Y'].T), g['imc0_predict_all'], rtol=1e-9)
    Zd, succd, _ = io.fit_imc(x_h, y_h, b_h)
    np.testing.assert_allclose(Zd[0, :10], [-1.04576367e2, -1.35697449e-3, -1.35697449e-3,
                                             -1.35697449e-3, 5.21791002e2, -1.35697449e-3,
                                             -1.04576367e2, -1.35697449e-3, 5.21791002e2,
                                             -1.35697449e-3], rtol=1e-6)
    assert bool(succd) == bool(g['fit_default_success'])


def test_imc_random_both_forms(golden):
    g = golden('imc_random')
    X, Y, users, items, vals = g['X'], g['Y'], g['users'], g['items'], g['vals']
    Z, S, lam = g['Z'], g['S'], float(g['lam'])
    args = (X[users], Y[items], vals, lam, Z.shape)
    np.testing.assert_allclose(io.cost(Z.ravel(), *args), g['cost'], rtol=1e-13)
    np.testing.assert_allclose(io.cost_prime(Z.ravel(), *args), g['grad'], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(io.cost_hess(Z.ravel(), S.ravel(), *args), g['hessp'], rtol=1e-11,
                               atol=1e-11)
    np.testing.assert_allclose(io.cost_indexed(Z, X, Y, users, items, vals, lam), g['cost'],
                               rtol=1e-13)
    np.testing.assert_allclose(io.grad_indexed(Z, X, Y, users, items, vals, lam).ravel(),
                               g['grad'], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(io.hessp_indexed(S, X, Y, users, items, lam).ravel(), g['hessp'],
                               rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(io._xt_diag_y(X, Y, users, items, vals) / vals.size, g['Z0'],
                               rtol=1e-11, atol=1e-13)
    Zf, _, _ = io.fit_imc(X[users], Y[items], vals, alpha=lam)
    np.testing.assert_allclose(Zf, g['fit_Z'], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(io.imc_score(Zf, X[users], Y[items], vals), g['fit_err'], rtol=1e-8)


def test_dimc_matches_reference(golden):
    g = golden('dimc')
    x_h, y_h = g['X'][g['rows']], g['Y'][g['cols']]
    r = g['R'][g['rows'], g['cols']].astype(np.float64)
    Z, uf, itf, err, pred = io.dimc_fit(x_h, y_h, g['rows'], g['cols'], r, (2, 2), rank=2,
                                        als_alpha=0.1, imc_alpha=0.01, tol=1e-6, seed=42)
    np.testing.assert_allclose(err, -0.14143343218755264, rtol=1e-6)
    np.testing.assert_allclose(pred, g['predict_sparse'], rtol=1e-8)
    n, m = (int(v) for v in g['rnd_shape'])
    users, items = g['rnd_users'], g['rnd_items']
    Z, uf, itf, err, pred = io.dimc_fit(g['rnd_X'][users], g['rnd_Y'][items], users, items,
                                        g['rnd_vals'], (n, m), rank=3, als_alpha=0.1,
                                        imc_alpha=0.1, tol=1e-3, seed=5)
    np.testing.assert_allclose(Z, g['rnd_Z'], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(err, g['rnd_err'], rtol=1e-6)
    np.testing.assert_allclose(pred, g['rnd_predict'], rtol=1e-5, atol=1e-6)


def test_low_rank_imc_gradient_by_finite_differences():
    """The rank-constrained objective has no reference counterpart (parity unpinned): its
    restated gradient is checked against central differences of its restated cost."""
    from oracle import imc_oracle as io
    rng = np.random.default_rng(3)
    n, m, p, q, r, ns = 20, 15, 6, 9, 3, 120
    X, Y = rng.standard_normal((n, p)), rng.standard_normal((m, q))
    users, items = rng.integers(0, n, ns), rng.integers(0, m, ns)
    vals = rng.standard_normal(ns)
    U, V = rng.standard_normal((p, r)), rng.standard_normal((q, r))
    f, gu, gv = io.low_rank_cost_grad(U, V, X, Y, users, items, vals, 0.2)
    eps = 1e-6
    for (mat, grad) in ((U, gu), (V, gv)):
        for idx in ((0, 0), (2, 1), (mat.shape[0] - 1, r - 1)):
            keep = mat[idx]
            mat[idx] = keep + eps
            fp = io.low_rank_cost_grad(U, V, X, Y, users, items, vals, 0.2)[0]
            mat[idx] = keep - eps
            fm = io.low_rank_cost_grad(U, V, X, Y, users, items, vals, 0.2)[0]
            mat[idx] = keep
            np.testing.assert_allclose((fp - fm) / (2 * eps), grad[idx], rtol=1e-6, atol=1e-7)
    Z, U2, V2 = io.fit_low_rank(X, Y, users, items, vals, r, 0.2)
    assert np.linalg.matrix_rank(Z) == r
    f_fit = io.low_rank_cost_grad(U2, V2, X, Y, users, items, vals, 0.2)
    assert f_fit[0] < f and np.abs(f_fit[1]).max() < 1e-3 * max(1.0, np.abs(gu).max())
