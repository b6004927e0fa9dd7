"""alsmi_topk_items / ALS.recommend against the ranking of the reference's dense predict_all
scores (src/als.py:347-364)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('dtype,rank,m,k', [('float64', 10, 517, 10), ('float32', 64, 3000, 100),
                                             ('float64', 3, 70, 70), ('float32', 128, 1025, 1),
                                             ('float32', 20, 40000, 1024)])
def test_topk_items_match_dense_ranking(dtype, rank, m, k):
    import torch
    from alsmi import device as dev
    from oracle import als_oracle as ao
    device = dev.require_gpu()
    rng = np.random.default_rng(rank + m)
    n = 90
    U = rng.standard_normal((rank, n))
    V = rng.standard_normal((rank, m))
    V[:, 5] = V[:, 3]                       # exact ties between two items
    V[:, m - 1] = V[:, 3]
    Ud = dev.factors_to_device(U, dtype, device)
    Vd = dev.factors_to_device(V, dtype, device)
    users = rng.integers(0, n, 37).astype(np.int32)
    # small workspace limit: several batches
    items, scores = dev.topk_items(rank, torch.from_numpy(users).to(device), Ud, Vd, k,
                                   max_workspace_bytes=m * 8 * 10)
    items, scores = items.cpu().numpy(), scores.to(torch.float64).cpu().numpy()
    Uq, Vq = dev.factors_to_host(Ud, rank), dev.factors_to_host(Vd, rank)
    want_items, want_scores = ao.topk_items(Uq, Vq, users, k)
    if dtype == 'float64':
        # scores to rounding; the ranking can only differ where two scores agree to rounding
        np.testing.assert_allclose(scores, want_scores, rtol=1e-12, atol=1e-12)
        differ = items != want_items
        assert np.abs(scores[differ] - want_scores[differ]).max(initial=0) < 1e-12
    else:
        np.testing.assert_allclose(scores, want_scores, rtol=2e-5, atol=2e-5)
    # properties that hold exactly: sorted best first, ties by item id, no item twice, and the
    # reported score is the score of the reported item
    assert (np.diff(scores, axis=1) <= 0).all()
    tie = np.diff(scores, axis=1) == 0
    assert (np.diff(items, axis=1)[tie] > 0).all()
    assert all(np.unique(r).size == k for r in items)
    dense = Uq[:, users].T.dot(Vq)
    np.testing.assert_allclose(np.take_along_axis(dense, items.astype(np.int64), axis=1), scores,
                               rtol=1e-12 if dtype == 'float64' else 2e-5, atol=2e-5)
    # nothing outside the list beats its last entry
    for r in range(users.size):
        rest = np.delete(dense[r], items[r])
        if rest.size:
            assert rest.max() <= scores[r, -1] + (1e-12 if dtype == 'float64' else 5e-5)


@pytest.mark.parametrize('dtype,rank,n_sel,m,k', [('float32', 64, 300, 1007, 10),
                                                   ('float64', 128, 150, 333, 33),
                                                   ('float32', 5, 65, 17, 17)])
def test_topk_many_users_partial_tiles(dtype, rank, n_sel, m, k):
    """Several blocks of users (the last one partly filled) and an item count that is no multiple
    of the 16-item tile: every reported score is the score of its item, best first, no item
    twice, nothing outside the list beats its last entry."""
    import torch
    from alsmi import device as dev
    device = dev.require_gpu()
    rng = np.random.default_rng(rank * 7 + n_sel)
    n = n_sel + 13
    Ud = dev.factors_to_device(rng.standard_normal((rank, n)), dtype, device)
    Vd = dev.factors_to_device(rng.standard_normal((rank, m)), dtype, device)
    users = rng.permutation(n)[:n_sel].astype(np.int32)
    items, scores = dev.topk_items(rank, torch.from_numpy(users).to(device), Ud, Vd, k)
    items = items.cpu().numpy().astype(np.int64)
    scores = scores.to(torch.float64).cpu().numpy()
    Uq, Vq = dev.factors_to_host(Ud, rank), dev.factors_to_host(Vd, rank)
    dense = Uq[:, users].T.dot(Vq)
    tol = 1e-12 if dtype == 'float64' else 5e-5
    np.testing.assert_allclose(np.take_along_axis(dense, items, axis=1), scores, rtol=tol, atol=tol)
    assert (np.diff(scores, axis=1) <= 0).all()
    assert all(np.unique(r).size == k for r in items)
    for r in range(n_sel):
        rest = np.delete(dense[r], items[r])
        if rest.size:
            assert rest.max() <= scores[r, -1] + 2 * tol


def test_als_recommend(golden):
    from alsmi import ALS
    import scipy.sparse as sps
    R = sps.csr_matrix(np.array([[3.0, 0, 4], [4, 5, 0]]))
    rows, cols = R.nonzero()
    als = ALS(2, tol=1e-6, random_state=42).fit((rows, cols), R)
    items, scores = als.recommend([0, 1, -1], k=2)
    for u, row, sc in zip((0, 1, 1), items, scores):
        full = als.predict_all(u)
        assert list(row) == list(np.argsort(-full, kind='stable')[:2])
        np.testing.assert_allclose(sc, full[row], rtol=1e-12)
    with pytest.raises(ValueError):
        als.recommend([0], k=4)
    with pytest.raises(IndexError):
        als.recommend([2], k=1)


@pytest.mark.parametrize('dtype', ['float32', 'float64'])
@pytest.mark.parametrize('k', [10, 100])
def test_topk_ties_across_the_kth_score(dtype, k):
    """Many items with EXACTLY the same score, so that the k-th score is shared by items
    inside and outside the list: of the equal items the smallest ids are returned, in
    ascending order (the ordered scan of the select kernel), also when every score is equal."""
    import torch
    from alsmi import device as dev
    device = dev.require_gpu()
    rng = np.random.default_rng(k)
    rank, n, m = 12, 40, 3000
    U = rng.standard_normal((rank, n))
    base = rng.standard_normal(rank)
    V = np.tile(base[:, None], (1, m))             # 200 groups of 15 identical items each
    groups = rng.permutation(m) % 200
    V += 0.01 * rng.standard_normal((rank, 200))[:, groups]
    Ud = dev.factors_to_device(U, dtype, device)
    Vd = dev.factors_to_device(V, dtype, device)
    users = np.arange(n, dtype=np.int32)
    items, scores = dev.topk_items(rank, torch.from_numpy(users).to(device), Ud, Vd, k)
    items, scores = items.cpu().numpy().astype(np.int64), scores.to(torch.float64).cpu().numpy()
    # the kernel's own scores for every item (k = m gives all of them), so that equality is
    # exact equality of what the select saw
    if m <= 1024:
        raise AssertionError('the case needs m above the k limit')
    full_i, full_s = dev.topk_items(rank, torch.from_numpy(users).to(device), Ud, Vd, 1024)
    for r in range(n):
        seen = dict(zip(full_i[r].cpu().numpy().tolist(),
                        full_s[r].to(torch.float64).cpu().numpy().tolist()))
        # group structure: items of one group share one score bit for bit
        by_score = {}
        for it, sc in seen.items():
            by_score.setdefault(sc, []).append(it)
        assert max(len(v) for v in by_score.values()) >= 10
        assert (np.diff(scores[r]) <= 0).all()
        kth = scores[r, -1]
        inside = items[r][scores[r] == kth]
        assert (np.diff(inside) > 0).all()
        # every item with the k-th score that was left out has a larger id than those kept
        same_group = np.flatnonzero(groups == groups[inside[0]])
        np.testing.assert_array_equal(inside, np.sort(same_group)[:inside.size])
        # inside a run of equal scores the ids ascend
        tie = np.diff(scores[r]) == 0
        assert (np.diff(items[r])[tie] > 0).all()
    # all scores equal: the first k item ids
    Vc = dev.factors_to_device(np.tile(base[:, None], (1, m)), dtype, device)
    items, _ = dev.topk_items(rank, torch.from_numpy(users[:3]).to(device), Ud, Vc, k)
    np.testing.assert_array_equal(items.cpu().numpy(), np.tile(np.arange(k), (3, 1)))


@pytest.mark.parametrize('dtype,rank,m,k', [('float32', 64, 20000, 10), ('float64', 10, 9001, 25),
                                             ('float32', 128, 12345, 1), ('float64', 64, 8200, 3)])
def test_topk_filtered_form_matches_dense_ranking(dtype, rank, m, k):
    """Item counts from 8192 on with a small k take the form that never writes the score block
    (threshold from a sample of the item tiles, candidates at or above it, sort): the same
    checks as for the two-kernel form, groups of identical items around the k-th score
    included, plus a table whose scores are ALL equal — every item is then a candidate, the
    slots overflow and the call must fall back to the two-kernel form on the device."""
    import torch
    from alsmi import device as dev
    from oracle import als_oracle as ao
    device = dev.require_gpu()
    rng = np.random.default_rng(rank + m + k)
    n = 150
    U = rng.standard_normal((rank, n))
    V = rng.standard_normal((rank, m))
    V[:, m - 1] = V[:, 7]                      # exact ties, one of them in the last (partial) tile
    V[:, 4000:4040] = V[:, [11]]               # forty identical items
    Ud = dev.factors_to_device(U, dtype, device)
    Vd = dev.factors_to_device(V, dtype, device)
    users = rng.permutation(n)[:97].astype(np.int32)
    items, scores = dev.topk_items(rank, torch.from_numpy(users).to(device), Ud, Vd, k)
    items, scores = items.cpu().numpy().astype(np.int64), scores.to(torch.float64).cpu().numpy()
    Uq, Vq = dev.factors_to_host(Ud, rank), dev.factors_to_host(Vd, rank)
    want_items, want_scores = ao.topk_items(Uq, Vq, users, k)
    tol = 1e-12 if dtype == 'float64' else 2e-5
    np.testing.assert_allclose(scores, want_scores, rtol=tol, atol=tol)
    differ = items != want_items
    assert np.abs(scores[differ] - want_scores[differ]).max(initial=0) < (1e-12 if dtype == 'float64' else 5e-5)
    assert (np.diff(scores, axis=1) <= 0).all()
    tie = np.diff(scores, axis=1) == 0
    assert (np.diff(items, axis=1)[tie] > 0).all()
    assert all(np.unique(r).size == k for r in items)
    dense = Uq[:, users].T.dot(Vq)
    np.testing.assert_allclose(np.take_along_axis(dense, items, axis=1), scores, rtol=tol, atol=2e-5)
    for r in range(users.size):
        rest = np.delete(dense[r], items[r])
        assert rest.max() <= scores[r, -1] + (1e-12 if dtype == 'float64' else 5e-5)
    # users whose best items are the forty identical ones: the smallest ids of the group first
    best_is_group = np.flatnonzero((items[:, 0] >= 4000) & (items[:, 0] < 4040))
    for r in best_is_group:
        run = items[r][scores[r] == scores[r, 0]]
        np.testing.assert_array_equal(run, (np.r_[11, 4000:4040] if 11 < 4000 else [])[:run.size])
    # all scores equal (and a second, two-valued table): slot overflow -> the fallback
    base = rng.standard_normal(rank)
    Vc = dev.factors_to_device(np.tile(base[:, None], (1, m)), dtype, device)
    it2, sc2 = dev.topk_items(rank, torch.from_numpy(users[:5]).to(device), Ud, Vc, k)
    np.testing.assert_array_equal(it2.cpu().numpy(), np.tile(np.arange(k), (5, 1)))
    assert (sc2 == sc2[:, :1]).all()
    # the same users, batched differently (another workspace limit), give the same lists
    again, sc_again = dev.topk_items(rank, torch.from_numpy(users).to(device), Ud, Vd, k,
                                     max_workspace_bytes=m * 8 * 33)
    sc_again = sc_again.to(torch.float64).cpu().numpy()
    if dtype == 'float32':
        np.testing.assert_array_equal(again.cpu().numpy(), items)
        np.testing.assert_array_equal(sc_again, scores)
    else:
        # fp64 lists come from an fp32 filter whose candidates are scored again in fp64, or —
        # when a user's candidate slots overflow, per call — from the dense fp64 block: the
        # two sum a score in different orders, so the last bits may differ between batchings
        np.testing.assert_allclose(sc_again, scores, rtol=1e-14, atol=1e-14)
        differ = again.cpu().numpy() != items
        assert np.abs(sc_again[differ] - scores[differ]).max(initial=0) < 1e-12


@pytest.mark.parametrize('rank,k', [(64, 50), (128, 10), (20, 200)])
def test_topk_fp64_ranks_scores_that_fp32_cannot_tell_apart(rank, k):
    """fp64 lists from 8192 items on come from an fp32 filter plus fp64 scores of the candidates
    (src/als.py:347-364 ranks fp64 scores).  300 items spread over the table whose vectors differ
    by a few 1e-9 are every user's best: fp32 scores cannot order them, the filter must keep them
    all and the fp64 scores of the candidates must give exactly numpy's fp64 ranking."""
    import torch
    from alsmi import device as dev
    device = dev.require_gpu()
    rng = np.random.default_rng(31 + rank)
    n, m = 40, 30000
    U = rng.standard_normal((rank, n))
    V = rng.standard_normal((rank, m)) * 0.2
    group = np.arange(300) * 97 + 5
    centre = U.mean(axis=1) * 3 + rng.standard_normal(rank) * 0.1
    step = rng.standard_normal((rank, 300)) * 3e-9
    V[:, group] = centre[:, None] + step
    U = U * 0.2 + U.mean(axis=1, keepdims=True) * 2        # every user likes the group best
    Ud = dev.factors_to_device(U, 'float64', device)
    Vd = dev.factors_to_device(V, 'float64', device)
    users = np.arange(n, dtype=np.int32)
    items, scores = dev.topk_items(rank, torch.from_numpy(users).to(device), Ud, Vd, k)
    items, scores = items.cpu().numpy().astype(np.int64), scores.cpu().numpy()
    Uq, Vq = dev.factors_to_host(Ud, rank), dev.factors_to_host(Vd, rank)
    dense = Uq.T.dot(Vq)
    order = np.lexsort((np.broadcast_to(np.arange(m), dense.shape), -dense), axis=1)[:, :k]
    assert np.isin(order, group).all()                     # the test is about the group
    want_scores = np.take_along_axis(dense, order, axis=1)
    np.testing.assert_allclose(scores, want_scores, rtol=1e-13)
    differ = items != order
    # neighbours in the ranking are about 1e-9 apart: the lists agree except where two fp64
    # scores agree to rounding
    assert np.abs(scores[differ] - want_scores[differ]).max(initial=0) < 1e-13
    assert differ.mean() < 0.01
    gaps = -np.diff(want_scores, axis=1)
    assert np.median(gaps) < 1e-7 and (np.diff(scores, axis=1) <= 0).all()

