"""HIP IMC / DirtyIMC path against the oracle and the reference's golden vectors."""
import warnings

import numpy as np
import pytest
import scipy.sparse as sps

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def gpu():
    from alsmi import device as dev
    return dev.require_gpu()


def unit_data(g):
    X, Y = g['X'], g['Y']
    R = sps.csr_matrix(g['R'])
    rows, cols = g['rows'], g['cols']
    return X, Y, R, X[rows], Y[cols], np.asarray(R[rows, cols]).ravel().astype(np.float64)


def test_callbacks_known_answers(gpu, golden):
    """reference tests/test_imc.py:53-80 through the scipy-callback signature."""
    from alsmi.imc import _cost, _cost_hess, _cost_prime
    g = golden('imc_unit')
    _, _, _, x_h, y_h, b_h = unit_data(g)
    args = (x_h, y_h, b_h, 0.01, (15, 20))
    assert _cost(np.zeros(300), *args) == 130545000.0
    np.testing.assert_allclose(_cost_prime(np.zeros(300), *args)[:10],
                               [-300, 0, 0, 0, -6000, 0, -300, 0, -6000, 0])
    np.testing.assert_allclose(_cost_hess(np.zeros(300), np.arange(300), *args)[:10],
                               [1.54801e3, 0.01, 0.01, 0.01, 1.56001e3, 0.01, 1.54801e3, 0.01,
                                1.56001e3, 0.01])
    np.testing.assert_allclose(_cost_prime(np.zeros(300), *args), g['grad0'], rtol=1e-12,
                               atol=1e-9)
    np.testing.assert_allclose(_cost_hess(np.zeros(300), np.arange(300), *args), g['hess0'],
                               rtol=1e-12)


@pytest.mark.parametrize('form', ['samples', 'indexed'])
def test_evaluations_match_reference_vectors(gpu, golden, form):
    from alsmi.imc import ImcProblem, problem_from_samples
    from oracle import imc_oracle as io
    g = golden('imc_random')
    X, Y, users, items, vals = g['X'], g['Y'], g['users'], g['items'], g['vals']
    Z, S, lam = g['Z'], g['S'], float(g['lam'])
    if form == 'samples':
        prob = problem_from_samples(X[users], Y[items], vals)
    else:
        # shuffled sample order: grouping by user must not change anything
        perm = np.random.default_rng(0).permutation(vals.size)
        prob = ImcProblem(X, Y, users[perm], items[perm], vals[perm])
    np.testing.assert_allclose(prob.cost(Z, lam), g['cost'], rtol=1e-12)
    np.testing.assert_allclose(prob.grad(Z, lam).ravel(), g['grad'], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(prob.hessp(S, lam).ravel(), g['hessp'], rtol=1e-10, atol=1e-10)
    z0 = prob.init_z()
    np.testing.assert_allclose(z0, g['Z0'], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(z0, io.init_z(X, Y, users, items, vals), rtol=1e-9, atol=1e-12)
    want = io.preds_indexed(Z, X, Y, users, items)
    got = prob.predict_samples(Z)
    if form == 'indexed':
        want = want[perm]
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize('dtype,tol', [('float64', 1e-10), ('float32', 2e-4)])
def test_users_with_few_entries_and_chunk_borders(gpu, dtype, tol):
    """Mean entries per user below 8: the short-chunk path (four users per wave, 128-entry
    chunks).  2 000 users with 0-5 entries and one with 700, which spans several chunks, so
    head, tail and whole-chunk border parts all occur; sample order shuffled."""
    from alsmi.imc import ImcProblem
    from oracle import imc_oracle as io
    rng = np.random.default_rng(11)
    n, m, p, q = 2000, 900, 12, 40
    deg = rng.integers(0, 6, n)
    deg[777] = 700
    users = np.repeat(np.arange(n), deg)
    items = np.concatenate([rng.choice(m, size=d, replace=False) for d in deg])
    perm = rng.permutation(users.size)
    users, items = users[perm], items[perm]
    vals = rng.standard_normal(users.size)
    X = rng.standard_normal((n, p)) / np.sqrt(p)
    Y = rng.standard_normal((m, q)) / np.sqrt(q)
    Z, S = rng.standard_normal((p, q)), rng.standard_normal((p, q))
    prob = ImcProblem(X, Y, users, items, vals, dtype=dtype)
    if dtype == 'float32':
        X, Y = X.astype(np.float32).astype(np.float64), Y.astype(np.float32).astype(np.float64)
    rel = lambda a, b: np.abs(np.asarray(a) - np.asarray(b)).max() / np.abs(np.asarray(b)).max()
    assert rel(prob.cost(Z, 0.3), io.cost_indexed(Z, X, Y, users, items, vals, 0.3)) <= tol
    assert rel(prob.grad(Z, 0.3), io.grad_indexed(Z, X, Y, users, items, vals, 0.3)) <= tol
    assert rel(prob.hessp(S, 0.3), io.hessp_indexed(S, X, Y, users, items, 0.3)) <= tol
    assert rel(prob.predict_samples(Z), io.preds_indexed(Z, X, Y, users, items)) <= tol


@pytest.mark.parametrize('p,q', [(3, 1), (17, 16), (5, 33), (130, 100), (40, 256)])
def test_evaluations_match_oracle_shapes(gpu, p, q):
    """attribute widths around every padding boundary, empty users, one long user."""
    from alsmi.imc import ImcProblem
    from oracle import imc_oracle as io
    rng = np.random.default_rng(p * 1000 + q)
    n, m, nnz = 70, 50, 900
    flat = rng.choice(n * m, size=nnz, replace=False)
    users, items = flat // m, flat % m
    keep = ~np.isin(users, (0, 13, 69))
    users, items = users[keep], items[keep]
    users = np.concatenate([users, np.full(m, 13)])
    items = np.concatenate([items, np.arange(m)])
    vals = rng.standard_normal(users.size)
    X = rng.standard_normal((n, p)) / np.sqrt(p)
    Y = rng.standard_normal((m, q)) / np.sqrt(q)
    Z, S = rng.standard_normal((p, q)), rng.standard_normal((p, q))
    prob = ImcProblem(X, Y, users, items, vals)
    np.testing.assert_allclose(prob.cost(Z, 0.3), io.cost_indexed(Z, X, Y, users, items, vals, 0.3),
                               rtol=1e-12)
    np.testing.assert_allclose(prob.grad(Z, 0.3), io.grad_indexed(Z, X, Y, users, items, vals, 0.3),
                               rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(prob.hessp(S, 0.3), io.hessp_indexed(S, X, Y, users, items, 0.3),
                               rtol=1e-10, atol=1e-10)


def test_imc_estimator_known_answers(gpu, golden):
    """reference tests/test_imc.py:121-210 in both input forms."""
    from alsmi import IMC
    g = golden('imc_unit')
    X, Y, R, x_h, y_h, b_h = unit_data(g)
    for args in (((x_h, y_h), b_h), ((X, Y), R)):
        imc = IMC(alpha=0.01)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            Z = imc.fit_transform(*args)
        np.testing.assert_allclose(Z, g['imc0_Z'], rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(Z[0, [0, 4, 6, 8]],
                                   [-1.06455138e2, 5.26175380e2, -1.06455138e2, 5.26175380e2],
                                   rtol=1e-7)
        np.testing.assert_allclose(imc.reconstruction_err_, 5.748504798089005, rtol=1e-8)
        np.testing.assert_allclose(imc.score(*args), 5.748504798089005, rtol=1e-8)
        np.testing.assert_allclose(imc.predict_one((X[0], Y[0])), 301.06449075427315, rtol=1e-8)
        np.testing.assert_allclose(imc._predict((X[[0]], Y[[0]])), 301.06449075427315, rtol=1e-8)
        np.testing.assert_allclose(imc.predict_all((X, Y)),
                                   [[301.06449075, 5994.73915285],
                                    [8994.67063528, 11991.34151006]], rtol=1e-8)
    imck = IMC(n_components=2)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        Zk = imck.fit_transform((x_h, y_h), b_h)
    assert imck.n_components_ == int(g['imck_n_components_'])
    np.testing.assert_allclose(Zk[0, :10], [-1.0457636e2, -1.35697449e-3, -1.35697449e-3,
                                            -1.35697449e-3, 5.217910e2, -1.35697449e-3,
                                            -1.0457636e2, -1.35697449e-3, 5.217910e2,
                                            -1.35697449e-3], rtol=1e-6)


def test_fit_imc_and_errors(gpu, golden):
    from sklearn.exceptions import NotFittedError
    from alsmi import IMC
    from alsmi.imc import _fit_imc
    g = golden('imc_unit')
    X, Y, _, x_h, y_h, b_h = unit_data(g)
    with pytest.raises(ValueError) as ctx:
        _fit_imc(x_h, y_h, b_h, n_components=0)
    assert str(ctx.value) == ('Number of components must be a positive integer; '
                              'got (n_components=0).')
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        Z, succ, _ = _fit_imc(x_h, y_h, b_h)
    np.testing.assert_allclose(Z, g['fit_default_Z'], rtol=1e-6, atol=1e-8)
    assert bool(succ) == bool(g['fit_default_success'])
    with pytest.raises(NotFittedError):
        IMC(alpha=0.01).predict_one((X[0], Y[0]))


def test_imc_random_fit(gpu, golden):
    from alsmi import IMC
    g = golden('imc_random')
    X, Y, users, items, vals = g['X'], g['Y'], g['users'], g['items'], g['vals']
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        imc = IMC(alpha=float(g['lam'])).fit((X[users], Y[items]), vals)
    np.testing.assert_allclose(imc.Z, g['fit_Z'], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(imc.reconstruction_err_, g['fit_err'], rtol=1e-8)


def test_dimc_matches_reference(gpu, golden):
    from alsmi import DirtyIMC
    g = golden('dimc')
    X, Y = g['X'], g['Y']
    R = sps.csr_matrix(g['R'])
    rows, cols = g['rows'], g['cols']
    x_h, y_h = X[rows], Y[cols]
    for tag, y, kw in (('sparse', R, {}), ('dense', np.asarray(R[rows, cols]).ravel(),
                                           {'shape': R.shape})):
        d = DirtyIMC(rank=2, tol=1e-6, random_state=42, imc_alpha=0.01)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            Z, u, i = d.fit_transform((x_h, y_h, rows, cols), y, **kw)
        np.testing.assert_allclose(d.reconstruction_err_, -0.14143343218755264, rtol=1e-5)
        np.testing.assert_allclose(d.predict((x_h, y_h, rows, cols)), g['predict_' + tag],
                                   rtol=1e-7)
    with pytest.raises(ValueError) as ctx:
        d.fit((x_h, y_h, rows, cols), np.asarray(R[rows, cols]).ravel())
    assert str(ctx.value) == 'When y is a scalar or 1-D array shape must be' + 'provided.'
    users, items = g['rnd_users'], g['rnd_items']
    n, m = (int(v) for v in g['rnd_shape'])
    d = DirtyIMC(rank=3, tol=1e-3, random_state=5, imc_alpha=0.1, als_alpha=0.1)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        Z, u, i = d.fit_transform((g['rnd_X'][users], g['rnd_Y'][items], users, items),
                                  g['rnd_vals'], shape=(n, m))
    np.testing.assert_allclose(Z, g['rnd_Z'], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(d.reconstruction_err_, g['rnd_err'], rtol=1e-6)
    np.testing.assert_allclose(d.predict((g['rnd_X'][users], g['rnd_Y'][items], users, items)),
                               g['rnd_predict'], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('parts', [2, 3, 5])
def test_sharded_evaluations_add_up(gpu, parts):
    """The multi-rank form of the IMC evaluations: each rank keeps one contiguous user range
    (and only the item rows its samples touch); the partial gradients, Hessian-vector
    products, squared errors and predictions of the shards add up to the unsharded result
    (on several ranks that sum is an all-reduce)."""
    from alsmi.imc import ImcProblem
    rng = np.random.default_rng(7)
    n, m, p, q, ns = 90, 70, 9, 21, 900
    X, Y = rng.standard_normal((n, p)), rng.standard_normal((m, q))
    users, items = rng.integers(0, n, ns), rng.integers(0, m, ns)
    users[users == 5] = 6            # a user without samples
    r = rng.standard_normal(ns)
    Z, S = rng.standard_normal((p, q)), rng.standard_normal((p, q))
    whole = ImcProblem(X, Y, users, items, r)
    g0, sse0, pred0 = whole._eval(0, Z, want_sse=True, want_pred=True)
    h0, _, _ = whole._eval(1, S)
    z0 = whole.init_z()
    g, h, sse, pred, zinit = 0.0, 0.0, 0.0, 0.0, 0.0
    seen = 0
    for i in range(parts):
        part = ImcProblem(X, Y, users, items, r, shard=(i, parts))
        gi, si, pi = part._eval(0, Z, want_sse=True, want_pred=True)
        hi, _, _ = part._eval(1, S)
        zi, _, _ = part._eval(0, np.zeros((p, q)), ratings=-part.ratings)
        g, h, sse, pred, zinit = g + gi, h + hi, sse + si, pred + pi, zinit + zi
        seen += part.n_local_samples
        assert part.Y.shape[0] <= m
    assert seen == ns
    np.testing.assert_allclose(g, g0, rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(h, h0, rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(sse, sse0, rtol=1e-12)
    np.testing.assert_allclose(pred, pred0, rtol=1e-12, atol=1e-13)
    u, sv, v = np.linalg.svd(zinit / ns, False)
    np.testing.assert_allclose(u.dot(np.diag(sv)).dot(v), z0, rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize('p,q', [(3, 1), (17, 16), (5, 33), (130, 100), (40, 256)])
def test_evaluations_match_oracle_shapes_fp32(gpu, p, q):
    """The same shapes in the fp32 build of the kernels (the throughput mode): fp32 products
    and row sums, fp64 squared-error accumulation."""
    from alsmi.imc import ImcProblem
    from oracle import imc_oracle as io
    rng = np.random.default_rng(p * 1000 + q)
    n, m, nnz = 70, 50, 900
    flat = rng.choice(n * m, size=nnz, replace=False)
    users, items = flat // m, flat % m
    keep = ~np.isin(users, (0, 13, 69))
    users, items = users[keep], items[keep]
    users = np.concatenate([users, np.full(m, 13)])
    items = np.concatenate([items, np.arange(m)])
    vals = rng.standard_normal(users.size)
    # inputs that are exact in fp32, so that only the arithmetic differs
    X = (rng.standard_normal((n, p)) / np.sqrt(p)).astype(np.float32).astype(np.float64)
    Y = (rng.standard_normal((m, q)) / np.sqrt(q)).astype(np.float32).astype(np.float64)
    vals = vals.astype(np.float32).astype(np.float64)
    Z = rng.standard_normal((p, q)).astype(np.float32).astype(np.float64)
    S = rng.standard_normal((p, q)).astype(np.float32).astype(np.float64)
    prob = ImcProblem(X, Y, users, items, vals, dtype=np.float32)
    want_g = io.grad_indexed(Z, X, Y, users, items, vals, 0.3)
    want_h = io.hessp_indexed(S, X, Y, users, items, 0.3)
    np.testing.assert_allclose(prob.cost(Z, 0.3), io.cost_indexed(Z, X, Y, users, items, vals, 0.3),
                               rtol=1e-5)
    # tolerance: 1e-4 of the largest entry (sums of ~900 fp32 products with cancellation)
    assert np.abs(prob.grad(Z, 0.3) - want_g).max() <= 1e-4 * np.abs(want_g).max()
    assert np.abs(prob.hessp(S, 0.3) - want_h).max() <= 1e-4 * np.abs(want_h).max()
    want_p = io.preds_indexed(Z, X, Y, users, items)
    assert np.abs(prob.predict_samples(Z) - want_p).max() <= 1e-5 * max(np.abs(want_p).max(), 1)


def test_dimc_attribute_form_matches_sample_form(gpu, golden):
    """DirtyIMC given one feature row per user / item plus sample indices (nothing expanded
    to per-sample rows) against the reference's per-sample form and the oracle
    (src/dimc.py:150-160)."""
    from alsmi import DirtyIMC
    from oracle import imc_oracle as io
    g = golden('dimc')
    users, items = g['rnd_users'], g['rnd_items']
    n, m = (int(v) for v in g['rnd_shape'])
    X, Y, vals = g['rnd_X'], g['rnd_Y'], g['rnd_vals']
    assert X.shape[0] == n and Y.shape[0] == m and users.size not in (n, m)
    outs = []
    for form, args in (('attribute', (X, Y, users, items)),
                       ('sample', (X[users], Y[items], users, items))):
        d = DirtyIMC(rank=3, tol=1e-3, random_state=5, imc_alpha=0.1, als_alpha=0.1)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            Z, u, i = d.fit_transform(args, vals, shape=(n, m))
        assert d._attribute_form(*args) == (form == 'attribute')
        np.testing.assert_allclose(Z, g['rnd_Z'], rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(d.reconstruction_err_, g['rnd_err'], rtol=1e-6)
        np.testing.assert_allclose(d.predict(args), g['rnd_predict'], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(d.score(args, vals), g['rnd_err'], rtol=1e-6)
        outs.append((Z, u, i))
    np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(outs[0][1], outs[1][1], rtol=1e-7, atol=1e-9)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        Zo, uo, io_, err, pred = io.dimc_fit(X[users], Y[items], users, items, vals, (n, m), 3,
                                             0.1, 0.1, 1e-3, 5)
    np.testing.assert_allclose(outs[0][0], Zo, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(outs[0][1], uo, rtol=1e-5, atol=1e-7)
    # sparse y with the attribute form, and a row count that fits neither form
    import scipy.sparse as sps
    R = sps.csr_matrix((vals, (users, items)), shape=(n, m))
    d = DirtyIMC(rank=3, tol=1e-3, random_state=5, imc_alpha=0.1, als_alpha=0.1)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        d.fit((X, Y, users, items), R)
    np.testing.assert_allclose(d.reconstruction_err_, g['rnd_err'], rtol=1e-6)
    with pytest.raises(ValueError):
        DirtyIMC(rank=3).fit((X[:-1], Y, users, items), vals, shape=(n, m))


def test_low_rank_imc_matches_restatement(gpu):
    """IMC(low_rank=True): Z = U V^T with n_components columns.  No reference counterpart
    (parity unpinned): checked against the numpy restatement of the same procedure."""
    from alsmi import IMC
    from oracle import imc_oracle as io
    import scipy.sparse as sps
    rng = np.random.default_rng(8)
    n, m, p, q, r, ns = 80, 60, 12, 20, 3, 1500
    X, Y = rng.standard_normal((n, p)) / np.sqrt(p), rng.standard_normal((m, q)) / np.sqrt(q)
    flat = rng.choice(n * m, size=ns, replace=False)
    users, items = flat // m, flat % m
    Zs = rng.standard_normal((p, r)).dot(rng.standard_normal((r, q)))
    vals = io.preds_indexed(Zs, X, Y, users, items) + 0.01 * rng.standard_normal(ns)
    R = sps.csr_matrix((vals, (users, items)), shape=(n, m))
    # the r-wide evaluation (two passes against projected tables) against the restatement,
    # whole and as the sum of three user-range shards, in both arithmetic types
    from alsmi.imc import ImcProblem
    Ut, Vt = rng.standard_normal((p, r)), rng.standard_normal((q, r))
    f_want, gu_want, gv_want = io.low_rank_cost_grad(Ut, Vt, X, Y, users, items, vals, 0.0)
    for dtype, tol in (('float64', 1e-11), ('float32', 1e-4)):
        sse, gu, gv = ImcProblem(X, Y, users, items, vals, dtype=dtype).low_rank_eval(Ut, Vt)
        np.testing.assert_allclose(0.5 * sse, f_want, rtol=tol)
        assert np.abs(gu - gu_want).max() <= tol * np.abs(gu_want).max()
        assert np.abs(gv - gv_want).max() <= tol * np.abs(gv_want).max()
    parts = [ImcProblem(X, Y, users, items, vals, shard=(i, 3)).low_rank_eval(Ut, Vt)
             for i in range(3)]
    np.testing.assert_allclose(sum(t[0] for t in parts) * 0.5, f_want, rtol=1e-11)
    np.testing.assert_allclose(sum(t[1] for t in parts), gu_want, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(sum(t[2] for t in parts), gv_want, rtol=1e-9, atol=1e-9)
    est = IMC(n_components=r, alpha=0.01, low_rank=True)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        Z = est.fit_transform((X, Y), R)
    assert est.n_components_ == r and est.U_.shape == (p, r) and est.V_.shape == (q, r)
    assert np.linalg.matrix_rank(Z, tol=1e-8) == r
    rows, cols = R.nonzero()
    Zo, Uo, Vo = io.fit_low_rank(X, Y, rows, cols, np.asarray(R[rows, cols]).ravel(), r, 0.01)
    np.testing.assert_allclose(Z, Zo, rtol=1e-4, atol=1e-5)
    # the planted rank-3 model is recovered, and a full-rank fit is no better on the samples
    assert est.reconstruction_err_ < 0.05
    np.testing.assert_allclose(est.score((X, Y), R), est.reconstruction_err_, rtol=1e-9)
    # default (reference behaviour): n_components is recorded, Z stays full rank
    ref = IMC(n_components=r, alpha=0.01)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        Zf = ref.fit_transform((X, Y), R)
    assert ref.n_components_ == r and np.linalg.matrix_rank(Zf, tol=1e-8) > r
