"""HIP ALS path against the oracle and the reference's golden vectors (MI355X)."""
import numpy as np
import pytest
import scipy.sparse as sps

from problems import random_problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def gpu():
    import torch
    from alsmi import device as dev
    return dev.require_gpu(), torch


def rel_err(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


def col_rel_err(a, b):
    """largest per-row (per factor vector) relative error; a, b are (rank, n)."""
    num = np.linalg.norm(a - b, axis=0)
    den = np.maximum(np.linalg.norm(b, axis=0), 1e-30)
    return float((num / den).max())


CASES = [
    # seed, n, m, nnz, rank, chunk, extras
    (1, 40, 30, 300, 2, 0, {}),
    (2, 60, 50, 900, 10, 0, dict(empty_rows=(0, 7, 59), empty_cols=(3,))),
    (3, 50, 300, 2500, 20, 64, dict(long_row=(5, 290))),
    (4, 80, 400, 3000, 64, 128, dict(long_row=(11, 390), empty_rows=(2,))),
    (5, 30, 200, 1500, 100, 64, dict(long_row=(3, 170))),
]


@pytest.mark.parametrize('case', CASES, ids=lambda c: 'k%d' % c[4])
@pytest.mark.parametrize('dtype', ['float64', 'float32'])
def test_normal_equations_and_half_sweep(gpu, case, dtype):
    device, torch = gpu
    from alsmi import device as dev
    from oracle import als_oracle as ao
    seed, n, m, nnz, rank, chunk, extra = case
    P = random_problem(seed, n, m, nnz, rank, **extra)
    alpha = 0.1
    side = dev.SparseSide(P['indptr'], P['indices'], P['vals'], n, m, dtype, device)
    plan = dev.SweepPlan(side, rank, chunk=chunk)
    if chunk:
        assert plan.info()['n_chunks'] > 1
    V = dev.factors_to_device(P['V'], dtype, device)
    Vq = dev.factors_to_host(V, rank)  # the table as the device sees it (fp32-rounded)
    kp = dev.padded_rank(rank)
    A, b = plan.normal_equations(V, alpha)
    A, b = A.cpu().numpy().astype(np.float64), b.cpu().numpy().astype(np.float64)
    tol = 1e-12 if dtype == 'float64' else 2e-6
    for r in range(n):
        lo, hi = P['indptr'][r], P['indptr'][r + 1]
        S = Vq[:, P['indices'][lo:hi]]
        if hi == lo:
            assert not A[r].any() and not b[r].any()
            continue
        Aref = np.zeros((kp, kp))
        Aref[:rank, :rank] = S.dot(S.T) + alpha * (hi - lo) * np.eye(rank)
        Aref[rank:, rank:] = np.eye(kp - rank)
        bref = np.zeros(kp)
        bref[:rank] = S.dot(P['vals'][lo:hi])
        assert rel_err(A[r], Aref) < tol, r
        assert rel_err(b[r], bref) < tol, r
    U = torch.full((n, kp), 7.0, dtype=V.dtype, device=device)
    plan.half_sweep(V, U, alpha)
    got = dev.factors_to_host(U, rank)
    assert not U[:, rank:].any()
    want = ao.half_sweep(P['indptr'], P['indices'], P['vals'], Vq, rank, alpha)
    # north-star bar: 1e-10 relative in fp64, 1e-4 in fp32
    assert col_rel_err(got, want) < (1e-10 if dtype == 'float64' else 1e-4)
    for r in (0, 2, 7, 59):
        if r < n and P['indptr'][r] == P['indptr'][r + 1]:
            assert not got[:, r].any()
    plan.close()


@pytest.mark.parametrize('case', CASES, ids=lambda c: 'k%d' % c[4])
@pytest.mark.parametrize('dtype', ['float64', 'float32'])
def test_half_sweep_with_error_of_replaced_factors(gpu, case, dtype):
    """alsmi_half_sweep_err: same new vectors as alsmi_half_sweep, plus the squared
    training error of the vectors it replaces (src/als.py:161-165), for short rows, rows cut
    into chunks and empty rows, out of place and in place."""
    device, torch = gpu
    from alsmi import device as dev
    from oracle import als_oracle as ao
    seed, n, m, nnz, rank, chunk, extra = case
    P = random_problem(seed, n, m, nnz, rank, **extra)
    alpha = 0.1
    side = dev.SparseSide(P['indptr'], P['indices'], P['vals'], n, m, dtype, device)
    plan = dev.SweepPlan(side, rank, chunk=chunk)
    V = dev.factors_to_device(P['V'], dtype, device)
    Ucur = dev.factors_to_device(P['U'], dtype, device)
    kp = dev.padded_rank(rank)
    plain = torch.full((n, kp), 7.0, dtype=V.dtype, device=device)
    plan.half_sweep(V, plain, alpha)
    want_sse = ao.train_sse(P['indptr'], P['indices'], P['vals'],
                            dev.factors_to_host(Ucur, rank), dev.factors_to_host(V, rank))
    # the error comes from sum r^2 - 2 u.b + u^T G u, whose terms cancel: 2e-5 on the sum of
    # squares is the 1e-5 bar on the RMSE
    tol = 1e-12 if dtype == 'float64' else 2e-5
    out = torch.full((n, kp), 7.0, dtype=V.dtype, device=device)
    keep = Ucur.clone()
    sse = plan.half_sweep_err(V, Ucur, out, alpha)
    assert torch.equal(Ucur, keep)
    # same arithmetic as the plain entry point, but a separately compiled kernel
    assert col_rel_err(dev.factors_to_host(out, rank), dev.factors_to_host(plain, rank)) < \
        (1e-12 if dtype == 'float64' else 1e-5)
    assert not out[:, rank:].any()
    assert abs(float(sse.cpu()[0]) - want_sse) <= tol * want_sse
    # in place, twice: the second call sees the new vectors
    sse2 = plan.half_sweep_err(V, Ucur, Ucur, alpha)
    assert torch.equal(Ucur, out)
    assert float(sse2.cpu()[0]) == float(sse.cpu()[0])
    sse3 = plan.half_sweep_err(V, Ucur, Ucur, alpha)
    want3 = ao.train_sse(P['indptr'], P['indices'], P['vals'],
                         dev.factors_to_host(out, rank), dev.factors_to_host(V, rank))
    assert abs(float(sse3.cpu()[0]) - want3) <= tol * max(want3, 1e-30)
    plan.close()


@pytest.mark.parametrize('rank,kp_rows', [(33, (34, 48)), (40, (41, 48)), (70, (40, 64))])
def test_rows_with_more_entries_than_dimensions_are_not_lost(gpu, rank, kp_rows):
    """Rows whose entry count lies between the rank and the largest size of the n x n
    ("dual") form must take the rank x rank form: with more entries than dimensions F F^T is
    rank deficient and, with large factors and a small ridge, its fp32 factorisation meets
    non-positive pivots and the row came back as zeros although the reference's matrix is
    well conditioned (found by tests/fuzz_half_sweep.py at rank 33).  The third case checks
    that rows the dual form may take (at most `rank` entries) are solved as before."""
    device, torch = gpu
    from alsmi import device as dev
    from oracle import als_oracle as ao
    n, m = 150, 430
    P = random_problem(77 + rank, n, m, n * 43, rank)
    deg = np.diff(P['indptr'])
    assert ((deg >= kp_rows[0]) & (deg <= kp_rows[1])).sum() > 20
    alpha = 0.003
    V = P['V'] * 1000.0
    side = dev.SparseSide(P['indptr'], P['indices'], P['vals'], n, m, 'float32', device)
    plan = dev.SweepPlan(side, rank)
    Vd = dev.factors_to_device(V, 'float32', device)
    Vq = dev.factors_to_host(Vd, rank)
    U = torch.full((n, dev.padded_rank(rank)), 7.0, dtype=Vd.dtype, device=device)
    plan.half_sweep(Vd, U, alpha)
    got = dev.factors_to_host(U, rank)
    want = ao.half_sweep(P['indptr'], P['indices'], P['vals'], Vq, rank, alpha)
    for r in range(n):
        lo, hi = P['indptr'][r], P['indptr'][r + 1]
        if hi - lo < rank:
            continue  # fewer entries than dimensions: conditioned by the ridge alone
        S = Vq[:, P['indices'][lo:hi]]
        cond = np.linalg.cond(S.dot(S.T) + alpha * (hi - lo) * np.eye(rank))
        assert got[:, r].any(), (r, hi - lo, cond)
        err = np.linalg.norm(got[:, r] - want[:, r]) / np.linalg.norm(want[:, r])
        assert err < 50 * cond * 2.0 ** -24, (r, hi - lo, cond, err)
    plan.close()


@pytest.mark.parametrize('rank', [128, 100, 84])
@pytest.mark.parametrize('dtype', ['float32', 'float64'])
def test_rows_of_65_to_96_entries_at_padded_rank_128(gpu, rank, dtype):
    """At padded rank 128 rows of up to 96 entries (and at most `rank` entries) take the
    neighbour-by-neighbour form, those above 64 with two entries per lane: every class of 16
    between 49 and 112 entries against the oracle, with the training error of the vectors the
    rows replace (src/als.py:161-165, 244-247); fp32 on the fp16 matrix cores, fp64 on the fp64
    ones."""
    device, torch = gpu
    from alsmi import device as dev
    from oracle import als_oracle as ao
    n, m = 400, 800
    P = random_problem(500 + rank, n, m, n * 80, rank)
    deg = np.diff(P['indptr'])
    for lo, hi in ((49, 64), (65, 80), (81, 96), (97, 112)):
        assert ((deg >= lo) & (deg <= hi)).sum() >= 8, (lo, hi)
    alpha = 0.05
    side = dev.SparseSide(P['indptr'], P['indices'], P['vals'], n, m, dtype, device)
    plan = dev.SweepPlan(side, rank)
    V = dev.factors_to_device(P['V'], dtype, device)
    Ucur = dev.factors_to_device(P['U'], dtype, device)
    Vq = dev.factors_to_host(V, rank)
    kp = dev.padded_rank(rank)
    assert kp == 128
    plain = torch.full((n, kp), 7.0, dtype=V.dtype, device=device)
    plan.half_sweep(V, plain, alpha)
    want = ao.half_sweep(P['indptr'], P['indices'], P['vals'], Vq, rank, alpha)
    got = dev.factors_to_host(plain, rank)
    assert not plain[:, rank:].any()
    assert col_rel_err(got, want) < (1e-4 if dtype == 'float32' else 1e-10)
    out = torch.full((n, kp), 7.0, dtype=V.dtype, device=device)
    sse = plan.half_sweep_err(V, Ucur, out, alpha)
    assert col_rel_err(dev.factors_to_host(out, rank), got) < (1e-5 if dtype == 'float32' else 1e-12)
    want_sse = ao.train_sse(P['indptr'], P['indices'], P['vals'],
                            dev.factors_to_host(Ucur, rank), Vq)
    assert abs(float(sse.cpu()[0]) - want_sse) <= (2e-5 if dtype == 'float32' else 1e-12) * want_sse
    plan.close()


@pytest.mark.parametrize('rank', [128, 96, 90, 64, 50, 33])
@pytest.mark.parametrize('dtype', ['float32', 'float64'])
def test_row_lengths_at_the_kernel_class_boundaries(gpu, rank, dtype):
    """Rows of exactly 1, 15, 16, 17, ... 95, 96, 97, 127, 128, 129 entries: every boundary
    between the kernels a plan deals its rows to (the neighbour-by-neighbour form in classes of
    16 entries up to min(rank, 48 or 96), the rank x rank form above), each row against the
    oracle (src/als.py:244-247), plain and with the error of the replaced vectors."""
    device, torch = gpu
    from alsmi import device as dev
    from oracle import als_oracle as ao
    lengths = [1, 2, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 79, 80, 81, 95, 96, 97, 111,
               112, 113, 127, 128, 129, 200]
    lengths = lengths + [rank - 1, rank, rank + 1]
    m = 400
    rng = np.random.default_rng(900 + rank)
    n = len(lengths)
    indptr = np.concatenate(([0], np.cumsum(lengths))).astype(np.int64)
    indices = np.concatenate([np.sort(rng.choice(m, k, replace=False)) for k in lengths]) \
        .astype(np.int32)
    vals = rng.integers(1, 11, indices.size).astype(np.float64) / 2
    V0 = rng.standard_normal((rank, m)) * 0.5 + 0.3
    U0 = rng.standard_normal((rank, n)) * 0.5 + 0.3
    alpha = 0.02
    side = dev.SparseSide(indptr, indices, vals, n, m, dtype, device)
    plan = dev.SweepPlan(side, rank)
    V = dev.factors_to_device(V0, dtype, device)
    Ucur = dev.factors_to_device(U0, dtype, device)
    Vq = dev.factors_to_host(V, rank)
    kp = dev.padded_rank(rank)
    out = torch.full((n, kp), 7.0, dtype=V.dtype, device=device)
    plan.half_sweep(V, out, alpha)
    want = ao.half_sweep(indptr, indices, vals, Vq, rank, alpha)
    got = dev.factors_to_host(out, rank)
    err = np.linalg.norm(got - want, axis=0) / np.linalg.norm(want, axis=0)
    tol = 1e-4 if dtype == 'float32' else 1e-10
    assert err.max() < tol, (lengths[int(np.argmax(err))], err.max())
    assert not out[:, rank:].any()
    out2 = torch.full((n, kp), 7.0, dtype=V.dtype, device=device)
    sse = plan.half_sweep_err(V, Ucur, out2, alpha)
    assert col_rel_err(dev.factors_to_host(out2, rank), got) < (1e-5 if dtype == 'float32' else 1e-12)
    want_sse = ao.train_sse(indptr, indices, vals, dev.factors_to_host(Ucur, rank), Vq)
    assert abs(float(sse.cpu()[0]) - want_sse) <= (2e-5 if dtype == 'float32' else 1e-12) * want_sse
    plan.close()


@pytest.mark.parametrize('dtype,rank', [('float32', 64), ('float32', 20), ('float32', 128),
                                        ('float64', 64)])
def test_half_sweep_is_reproducible(gpu, dtype, rank):
    """The same half-sweep run three times gives bit-identical factors and error sums (fixed
    summation order, no timing-dependent path), with long rows cut into chunks included."""
    device, torch = gpu
    from alsmi import device as dev
    P = random_problem(77, 6000, 900, 400000, rank, long_row=(17, 850))
    side = dev.SparseSide(P['indptr'], P['indices'], P['vals'], 6000, 900, dtype, device)
    plan = dev.SweepPlan(side, rank, chunk=256)
    assert plan.info()['n_chunks'] > 1
    V = dev.factors_to_device(P['V'], dtype, device)
    U0 = dev.factors_to_device(P['U'], dtype, device)
    outs, sses = [], []
    for _ in range(3):
        a = torch.zeros_like(U0)
        plan.half_sweep(V, a, 0.1)
        b = torch.zeros_like(U0)
        sses.append(float(plan.half_sweep_err(V, U0, b, 0.1).cpu()[0]))
        outs.append((a, b))
    for a, b in outs[1:]:
        assert torch.equal(a, outs[0][0]) and torch.equal(b, outs[0][1])
    assert sses[1] == sses[0] and sses[2] == sses[0]
    plan.close()


@pytest.mark.parametrize('scale', [1e-9, 1e-4, 1.0, 3e3, 1e7])
def test_fp32_half_sweep_any_factor_magnitude(gpu, scale):
    """The fp32 Gram build splits scaled values into fp16 pairs; the scale follows the largest
    magnitude of the source table, so tiny and huge factors keep fp32-level accuracy (the ridge
    is scaled along so that the conditioning is that of the unit-scale problem)."""
    device, torch = gpu
    from alsmi import device as dev
    from oracle import als_oracle as ao
    rank = 64
    P = random_problem(31, 300, 400, 20000, rank)
    Vs = P['V'] * scale
    Vs[:, 7] *= 1e-5      # one row far below the rest
    Vs[3, :] *= 1e-3      # one dimension far below the rest
    alpha = 0.1 * scale * scale
    side = dev.SparseSide(P['indptr'], P['indices'], P['vals'], 300, 400, 'float32', device)
    plan = dev.SweepPlan(side, rank)
    V = dev.factors_to_device(Vs, 'float32', device)
    U = V.new_zeros((300, dev.padded_rank(rank)))
    plan.half_sweep(V, U, alpha)
    got = dev.factors_to_host(U, rank)
    want = ao.half_sweep(P['indptr'], P['indices'], P['vals'], dev.factors_to_host(V, rank), rank,
                         alpha)
    assert col_rel_err(got, want) < 1e-4
    plan.close()


def test_fp32_half_sweep_one_tiny_dimension(gpu):
    """What "elements below max * 2^-16 lose precision relative to themselves" means for the
    solved vector: one factor DIMENSION uniformly 2^-20 of the others.  Its low fp16 half falls
    into the subnormal range, so its Gram entries keep about 17 instead of 24 bits; the ridge
    dominates that dimension's equation, the vector as a whole stays within the fp32 bar and
    the tiny dimension's own components within 1e-4 (median) / 1e-3 (over all rows, in norm) of
    their values."""
    device, torch = gpu
    from alsmi import device as dev
    from oracle import als_oracle as ao
    rank = 64
    P = random_problem(33, 300, 400, 20000, rank)
    Vs = P['V'].copy()
    Vs[5, :] *= 2.0 ** -20
    side = dev.SparseSide(P['indptr'], P['indices'], P['vals'], 300, 400, 'float32', device)
    plan = dev.SweepPlan(side, rank)
    V = dev.factors_to_device(Vs, 'float32', device)
    U = V.new_zeros((300, dev.padded_rank(rank)))
    plan.half_sweep(V, U, 0.1)
    got = dev.factors_to_host(U, rank)
    want = ao.half_sweep(P['indptr'], P['indices'], P['vals'], dev.factors_to_host(V, rank), rank,
                         0.1)
    assert col_rel_err(got, want) < 1e-4
    comp = np.abs(got[5] - want[5]) / np.maximum(np.abs(want[5]), 1e-30)
    across = np.linalg.norm(got[5] - want[5]) / np.linalg.norm(want[5])
    assert np.median(comp) < 1e-4 and across < 1e-3, (np.median(comp), across)
    # the normal equations the kernel shows are the ones production accumulates (fp16 split):
    # row / column 5 of the Gram matrix to 2^-16 of its own size, the rest to fp32 level
    A, _ = plan.normal_equations(V, 0.1)
    A = A.cpu().numpy().astype(np.float64)
    Vq = dev.factors_to_host(V, rank)
    lo, hi = P['indptr'][7], P['indptr'][8]
    S = Vq[:, P['indices'][lo:hi]]
    Aref = S.dot(S.T) + 0.1 * (hi - lo) * np.eye(rank)
    others = np.delete(np.arange(rank), 5)
    assert rel_err(A[7][np.ix_(others, others)], Aref[np.ix_(others, others)]) < 2e-6
    assert rel_err(A[7][5, others], Aref[5, others]) < 2.0 ** -14
    plan.close()


def test_fp32_half_sweep_zero_source_table(gpu):
    """An all-zero source table (largest magnitude 0) with a positive ridge: zero solution."""
    device, torch = gpu
    from alsmi import device as dev
    P = random_problem(32, 50, 60, 900, 64)
    side = dev.SparseSide(P['indptr'], P['indices'], P['vals'], 50, 60, 'float32', device)
    plan = dev.SweepPlan(side, 64)
    V = torch.zeros((60, 64), dtype=torch.float32, device=device)
    U = torch.ones((50, 64), dtype=torch.float32, device=device)
    plan.half_sweep(V, U, 0.1)
    assert not U.cpu().numpy().any()
    plan.close()


def test_user_slabs_change_nothing(gpu):
    """The engine may cut the user half into row slabs (to overlap their all-gathers with the
    computation on several ranks); a row's result does not depend on the cut."""
    device, torch = gpu
    from alsmi.engine import AlsEngine
    P = random_problem(41, 500, 300, 20000, 20, long_row=(9, 280))
    res = []
    for slabs in (1, 3):
        eng = AlsEngine.from_global(P['indptr'], P['indices'], P['vals'], (500, 300), 20, 0.1,
                                    dtype='float32', user_slabs=slabs, chunk=64)
        assert len(eng.plans_u) == slabs
        eng.set_factors(np.zeros_like(P['U']), P['V'])
        eng.sweep()
        sse = float(eng.half_sweep_users_err().cpu()[0])
        eng.half_sweep_items()
        res.append((eng.U.clone(), eng.V.clone(), sse))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert abs(res[0][2] - res[1][2]) <= 1e-12 * res[0][2]


def test_half_sweep_singular_gives_zeros(gpu):
    """A = 0 (zero source factors, alpha = 0): the reference's inv() raises
    LinAlgError and the row becomes zeros (src/als.py:246-249)."""
    device, torch = gpu
    from alsmi import device as dev
    for dtype, rank in (('float64', 20), ('float32', 64), ('float64', 100)):
        P = random_problem(9, 6, 40, 60, rank)
        side = dev.SparseSide(P['indptr'], P['indices'], P['vals'], 6, 40, dtype, device)
        plan = dev.SweepPlan(side, rank)
        V = dev.factors_to_device(np.zeros_like(P['V']), dtype, device)
        U = torch.ones((6, dev.padded_rank(rank)), dtype=V.dtype, device=device)
        plan.half_sweep(V, U, 0.0)
        assert not U.cpu().numpy().any()
        plan.close()


@pytest.mark.parametrize('dtype', ['float64', 'float32'])
@pytest.mark.parametrize('rank', [3, 10, 64, 128])
def test_sddmm_matches_oracle(gpu, dtype, rank):
    device, torch = gpu
    from alsmi import device as dev
    from oracle import als_oracle as ao
    P = random_problem(20 + rank, 70, 90, 1531, rank)
    U = dev.factors_to_device(P['U'], dtype, device)
    V = dev.factors_to_device(P['V'], dtype, device)
    Uq, Vq = dev.factors_to_host(U, rank), dev.factors_to_host(V, rank)
    users = torch.from_numpy(P['users']).to(device)
    items = torch.from_numpy(P['items']).to(device)
    vals = torch.from_numpy(P['vals'].astype(dtype)).to(device)
    pred, sse = dev.sddmm(rank, users, items, U, V, vals)
    want = ao.predict_pairs(Uq, Vq, P['users'], P['items'])
    tol = 1e-13 if dtype == 'float64' else 2e-6
    np.testing.assert_allclose(pred.cpu().numpy(), want, rtol=tol, atol=tol)
    np.testing.assert_allclose(float(sse.cpu()[0]), np.sum((P['vals'] - want) ** 2),
                               rtol=1e-12 if dtype == 'float64' else 1e-5)
    # prediction only, and the empty call
    pred2, none = dev.sddmm(rank, users[:5], items[:5], U, V)
    assert none is None and torch.equal(pred2, pred[:5])
    e = torch.empty(0, dtype=torch.int32, device=device)
    pred3, sse3 = dev.sddmm(rank, e, e, U, V, vals[:0])
    assert pred3.numel() == 0 and float(sse3.cpu()[0]) == 0.0


def test_expand_rows(gpu):
    device, torch = gpu
    from alsmi import device as dev
    P = random_problem(31, 50, 40, 700, 4, empty_rows=(0, 10, 49))
    side = dev.SparseSide(P['indptr'], P['indices'], P['vals'], 50, 40, 'float64', device)
    assert np.array_equal(side.expand_rows().cpu().numpy(), P['users'])
    lo, hi = P['indptr'][10], P['indptr'][30]
    assert np.array_equal(side.expand_rows(10, 30).cpu().numpy(), P['users'][lo:hi])


def test_als_unit_known_answers(gpu, golden):
    """reference tests/test_als.py: fit, fit_transform, predict*, score, add_user."""
    from alsmi import ALS
    R = sps.csr_matrix(np.array([[3, 0, 4], [4, 5, 0]]))
    users, items = R.nonzero()
    als = ALS(2, tol=0.000001, random_state=42)
    u, i = als.fit_transform((users, items), R)
    expected_u = np.array([[1.52198247, 1.97980245], [-0.79543119, -1.03470076]])
    expected_i = np.array([[1.5339228, 1.94471076, 1.99660463],
                           [-0.80167146, -1.01636085, -1.04348219]])
    np.testing.assert_allclose(expected_u, u, rtol=1e-7)
    np.testing.assert_allclose(expected_i, i, rtol=1e-7)
    g = golden('als_unit')
    np.testing.assert_allclose(u, g['user_feats'], rtol=1e-9)
    np.testing.assert_allclose(i, g['item_feats'], rtol=1e-9)
    assert 2 * als.n_iter_ == int(g['half_sweeps'])
    np.testing.assert_allclose(als.reconstruction_err_, -0.10663924910933506, rtol=1e-9)
    with pytest.raises(ValueError) as ctx:
        als.fit_transform((users, items), R.data)
    assert str(ctx.value) == 'When y is a scalar or 1-D array shape must be' + 'provided.'
    u2, i2 = als.fit_transform((users, items), R.data, shape=R.shape)
    np.testing.assert_allclose(expected_u, u2, rtol=1e-7)
    np.testing.assert_allclose(expected_i, i2, rtol=1e-7)
    np.testing.assert_allclose(als._predict((users, items)),
                               [2.9722781, 3.86881553, 3.86635418, 4.90177248], rtol=1e-7)
    one = als.predict_one(0, 1)
    assert one.shape == (1,)
    np.testing.assert_allclose(one, 3.768260826602256, rtol=1e-9)
    np.testing.assert_allclose(als.predict_all(0), [2.972278, 3.76826083, 3.86881553], rtol=1e-7)
    np.testing.assert_allclose(als.score((users, items), R.data), -0.10663924910933506, rtol=1e-9)
    np.testing.assert_allclose(als.score((users, items), R), -0.10663924910933506, rtol=1e-9)
    # update_user: expected matrix of reference tests/test_als.py:122-124
    als.update_user(0, 1, 5)
    np.testing.assert_allclose(als.user_feats, [[1.69210752, 1.97980245],
                                                [-0.8843431, -1.03470076]], rtol=1e-7)
    als.add_user()
    assert als.data.shape[0] == 3 and als.user_feats.shape[1] == 3


@pytest.mark.parametrize('name', ['als_c1', 'als_k64', 'als_k3'])
def test_als_sweeps_match_reference_fp64(gpu, golden, name):
    from alsmi import ALS
    g = golden(name)
    n, m = (int(v) for v in g['shape'])
    keep = [int(s) for s in g['keep']]
    for s in keep:
        als = ALS(rank=int(g['rank']), alpha=float(g['alpha']), random_state=int(g['seed']),
                  max_iter=s)
        u, i = als.fit_transform((g['users'], g['items']), g['vals'], shape=(n, m))
        assert col_rel_err(u, g['user_feats_%d' % s]) < 1e-9
        assert col_rel_err(i, g['item_feats_%d' % s]) < 1e-9
        np.testing.assert_allclose(als.rmse_history_, g['rmse'][:s], rtol=1e-11)
    np.testing.assert_allclose(als._predict((g['pair_users'], g['pair_items'])), g['pair_pred'],
                               rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(als.score((g['users'], g['items']), g['vals']),
                               g['score_train'], rtol=1e-9)
    # index structure is bit-exact
    ref = sps.csr_matrix((g['vals'], (g['users'], g['items'])), shape=(n, m))
    assert np.array_equal(als.data.indptr, ref.indptr)
    assert np.array_equal(als.data.indices, ref.indices)
    assert np.array_equal(als.data.data, ref.data)


def test_als_c1_fp32(gpu, golden):
    """BASELINE config 1 in the fp32 throughput mode: RMSE within 1e-5 relative
    per sweep, factors after the first half-sweeps within 1e-4."""
    from alsmi import ALS
    g = golden('als_c1')
    n, m = (int(v) for v in g['shape'])
    als = ALS(rank=10, alpha=0.1, random_state=42, max_iter=1, dtype='float32')
    u, i = als.fit_transform((g['users'], g['items']), g['vals'], shape=(n, m))
    assert col_rel_err(u, g['user_feats_1']) < 1e-4
    assert col_rel_err(i, g['item_feats_1']) < 1e-4
    als = ALS(rank=10, alpha=0.1, random_state=42, max_iter=10, dtype='float32')
    als.fit((g['users'], g['items']), g['vals'], shape=(n, m))
    np.testing.assert_allclose(als.rmse_history_, g['rmse'], rtol=1e-5)


def test_als_tol_loop(gpu, golden):
    from alsmi import ALS
    g = golden('als_fit_tol')
    n, m = (int(v) for v in g['shape'])
    als = ALS(rank=5, alpha=0.1, tol=0.001, random_state=3)
    u, i = als.fit_transform((g['users'], g['items']), g['vals'], shape=(n, m))
    assert 2 * als.n_iter_ == int(g['half_sweeps_dense'])
    np.testing.assert_allclose(u, g['user_feats_dense'], rtol=1e-8)
    np.testing.assert_allclose(als.reconstruction_err_, g['err_dense'], rtol=1e-10)
    R = sps.csr_matrix((g['vals'], (g['users'], g['items'])), shape=(n, m))
    als2 = ALS(rank=5, alpha=0.1, tol=0.001, random_state=3).fit((g['users'], g['items']), R)
    np.testing.assert_allclose(als2.item_feats, g['item_feats_sparse'], rtol=1e-8)


def test_fp32_tol_loop_on_exactly_low_rank_ratings(gpu):
    """Ratings that ARE a rank-2 product: the fit becomes near exact, the fused error (a
    difference of three float32 sums) drops into its rounding noise, and the loop falls back
    on the standalone pass instead of taking a square root of a negative number."""
    from alsmi import ALS
    rng = np.random.default_rng(4)
    n, m = 400, 300
    A, B = rng.uniform(0.5, 1.5, (n, 2)), rng.uniform(0.5, 1.5, (m, 2))
    flat = rng.choice(n * m, size=30000, replace=False)
    users, items = flat // m, flat % m
    vals = (A[users] * B[items]).sum(-1)
    als = ALS(rank=2, alpha=1e-7, tol=1e-7, random_state=3, dtype='float32')
    als.fit((users, items), vals, shape=(n, m))
    hist = np.asarray(als.rmse_history_)
    assert np.isfinite(hist).all() and (hist >= 0).all() and als.n_iter_ >= 3
    assert hist[-1] < 1e-3 * np.sqrt(np.mean(vals ** 2))
    # predictions come from the device tables kept by fit; a changed host factor array is seen
    p0 = als.predict_one(3, 5)
    np.testing.assert_allclose(p0, als.user_feats[:, 3].dot(als.item_feats[:, 5]), rtol=1e-5)
    als.user_feats = als.user_feats * 2.0
    np.testing.assert_allclose(als.predict_one(3, 5), 2.0 * p0, rtol=1e-6)


def test_in_place_edits_of_the_host_factors_are_seen(gpu):
    """Predictions come from device copies of user_feats / item_feats that are kept between
    calls; an in-place edit of the host arrays, a replaced array and invalidate() must all
    lead to predictions from the current factors (src/als.py:316-321 reads them per call)."""
    from alsmi import ALS
    rng = np.random.default_rng(8)
    n, m, nnz = 60, 40, 700
    flat = rng.choice(n * m, size=nnz, replace=False)
    users, items = flat // m, flat % m
    vals = rng.integers(1, 6, nnz).astype(np.float64)
    als = ALS(rank=5, random_state=0, max_iter=3).fit((users, items), vals, shape=(n, m))
    pairs = (users[:50], items[:50])
    want = lambda: (als.user_feats[:, pairs[0]] * als.item_feats[:, pairs[1]]).sum(0)
    np.testing.assert_allclose(als._predict(pairs), want(), rtol=1e-12)
    als.item_feats *= 2.0                          # in place: same array object
    np.testing.assert_allclose(als._predict(pairs), want(), rtol=1e-12)
    als.user_feats[:, 0] = 7.0                     # one column, in place
    np.testing.assert_allclose(als._predict((np.zeros(5, int), np.arange(5))),
                               (als.user_feats[:, [0]] * als.item_feats[:, :5]).sum(0), rtol=1e-12)
    als.user_feats = als.user_feats * 0.5          # a new array object
    np.testing.assert_allclose(als._predict(pairs), want(), rtol=1e-12)
    # an edit in the interior escapes the signature (first/last row and column sums): the
    # documented way is invalidate()
    als.item_feats[2, 7] += 1.0
    als.invalidate()
    np.testing.assert_allclose(als._predict(pairs), want(), rtol=1e-12)
    assert als._cached_tables() is not None and als._cached_tables()[2] is als.user_feats
