"""Host-side plumbing of alsmi.utils (mirrors the reference's tests/test_utils.py)."""
import numpy as np
import pytest
import scipy.sparse as sps

from alsmi.utils import (DataHolder, _check_x, _check_y, _format_data, root_mean_squared_error,
                         triplets_to_csr)


def data():
    X = np.array([[1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 0],
                  [0, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 1, 0]])
    Y = np.array([[1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0],
                  [0, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0]])
    R = sps.csr_matrix(np.array([[300, 6000], [9000, 12000]]))
    users, items = R.nonzero()
    return X, Y, R, users, items


def test_rmse():
    # reference tests/test_utils.py:34-41
    _, _, R, _, _ = data()
    assert root_mean_squared_error(np.array([301, 6001, 9001, 12001]), R.data) == 1


def test_check_x_and_errors():
    # reference tests/test_utils.py:43-78
    X, Y, _, users, items = data()
    assert _check_x((X, Y)) == (X, Y)
    out = _check_x(DataHolder(X, Y))
    assert out[0] is X and out[1] is Y
    out = _check_x(DataHolder(X, Y, users, items), indices=True)
    assert len(out) == 4 and out[2] is users and out[3] is items
    with pytest.raises(TypeError) as ctx:
        _check_x([X, Y])
    assert str(ctx.value) == 'Type of argument X should be tuple or DataHolder, was list.'
    with pytest.raises(ValueError) as ctx:
        _check_x((X, Y, users))
    assert str(ctx.value) == 'Argument X should be a tuple of length 2.'
    with pytest.raises(ValueError) as ctx:
        _check_x((X, Y), indices=True)
    assert str(ctx.value) == 'Argument X should be a tuple of length 4.'


def test_format_data():
    # reference tests/test_utils.py:80-94
    X, Y, R, users, items = data()
    x, y, r = _format_data((X, Y), R)
    np.testing.assert_equal((x, y, r), (X[users], Y[items], R.data))
    x, y, r = _format_data((X[users], Y[items]), R.toarray())
    np.testing.assert_equal((x, y, r), (X[users], Y[items], R.data))


def test_data_holder_getitem():
    # reference tests/test_utils.py:96-113
    X, Y, _, users, items = data()
    np.testing.assert_equal(DataHolder(X, Y)[0], (X[0], Y[0]))
    np.testing.assert_equal(DataHolder(X, Y, users, items)[0], (X[0], Y[0], users[0], items[0]))
    assert DataHolder(X, Y).shape == X.shape


def test_check_y():
    _, _, R, users, items = data()
    np.testing.assert_equal(_check_y(R, users, items), R.data)
    np.testing.assert_equal(_check_y(np.array([[1, 2], [3, 4]]), None, None), [1, 2, 3, 4])


def test_triplets_to_csr_lil_semantics():
    rng = np.random.default_rng(0)
    n, m, k = 12, 9, 150
    users, items = rng.integers(0, n, k), rng.integers(0, m, k)
    y = rng.choice([0.0, 1.0, 2.0, 3.5], size=k)
    ref = sps.lil_matrix((n, m))
    for idx, (i, j) in enumerate(zip(users, items)):
        ref[i, j] = y[idx]
    ref = ref.tocsr()
    indptr, indices, vals = triplets_to_csr(users, items, y, (n, m))
    assert np.array_equal(indptr, ref.indptr) and np.array_equal(indices, ref.indices)
    assert np.array_equal(vals, ref.data)
    with pytest.raises(IndexError):
        triplets_to_csr([n], [0], [1.0], (n, m))
