"""Host-side input plumbing shared by the estimators.

Mirrors the helper surface of the reference's ``src/utils.py`` (same names,
argument meaning, exception types and messages) plus ``_format_data``, which the
reference's ``tests/test_utils.py:80-94`` still specifies although the function
was dropped from ``src/utils.py``.
"""
import numpy as np
from scipy.sparse import issparse


def root_mean_squared_error(true, pred):
    """sqrt(mean((true - pred)^2)) — src/utils.py:7-24."""
    true = np.asarray(true, dtype=np.float64)
    pred = np.asarray(pred, dtype=np.float64)
    return np.sqrt(np.average((true - pred) ** 2, axis=0))


def _check_x(X, indices=False):
    """Unpack a tuple or DataHolder into its arrays — src/utils.py:27-68."""
    len_ = 4 if indices else 2
    if isinstance(X, tuple):
        if len(X) != len_:
            raise ValueError('Argument X should be a tuple of length {}.'.format(len_))
        out = X
    elif isinstance(X, DataHolder):
        if len_ == 2:
            out = (X.X, X.Y)
        else:
            out = (X.X, X.Y, X.X_ind, X.Y_ind)
    else:
        raise TypeError('Type of argument X should be tuple or DataHolder, was'
                        ' {}.'.format(str(type(X)).split("'")[1]))
    return out


def _check_y(y, rows, cols):
    """True values as a flat array — src/utils.py:71-96 (sparse: y[rows, cols])."""
    if issparse(y):
        out = np.asarray(y.tocsr()[rows, cols]).ravel()
    else:
        out = np.asarray(y).flatten()
    return out


def _format_data(X, y):
    """Per-sample form of (attribute matrices, ratings).

    Sparse ``y``: X holds the user and item ATTRIBUTE matrices and the result is
    ``(X[rows], Y[cols], y.data)`` over the stored entries; dense ``y``: X is
    already per-sample and only the stored (non-zero) values of ``y`` are kept,
    in row-major order (reference tests/test_utils.py:80-94)."""
    x, y_ = _check_x(X)
    if issparse(y):
        y = y.tocsr()
        rows, cols = y.nonzero()
        return x[rows], y_[cols], y.data
    y = np.asarray(y)
    return x, y_, y[y != 0] if y.ndim > 1 else y


class DataHolder(object):
    """Packs user/item arrays (and optionally their indices) so that
    cross-validation splitters can index all of them at once —
    src/utils.py:99-125."""

    def __init__(self, X, Y, X_ind=None, Y_ind=None):
        self.X = X
        self.Y = Y
        if X_ind is not None and Y_ind is not None:
            self.X_ind = X_ind
            self.Y_ind = Y_ind
        self.shape = self.X.shape

    def __getitem__(self, x):
        if hasattr(self, 'X_ind'):
            return self.X[x], self.Y[x], self.X_ind[x], self.Y_ind[x]
        return self.X[x], self.Y[x]


def triplets_to_csr(users, items, y, shape):
    """CSR arrays of ratings given as triplets.

    Same observable result as the reference's element-by-element fill of a
    ``lil_matrix`` followed by ``tocsr()`` (src/als.py:137-140): when a position
    occurs more than once the LAST value wins, a final value of exactly zero is
    not stored, column indices inside a row are sorted."""
    n, m = int(shape[0]), int(shape[1])
    users = np.asarray(users).ravel().astype(np.int64)
    items = np.asarray(items).ravel().astype(np.int64)
    y = np.asarray(y).ravel()
    if not (users.size == items.size == y.size):
        raise ValueError('users, items and y must have the same length')
    if users.size and (users.min() < -n or users.max() >= n or
                       items.min() < -m or items.max() >= m):
        raise IndexError('index out of bounds for shape {}'.format((n, m)))
    users = np.where(users < 0, users + n, users)
    items = np.where(items < 0, items + m, items)
    key = users * m + items
    order = np.argsort(key, kind='stable')
    sorted_key = key[order]
    is_last = np.ones(sorted_key.size, dtype=bool)
    is_last[:-1] = sorted_key[1:] != sorted_key[:-1]
    pick = order[is_last]
    vals = y[pick].astype(np.float64)
    stored = vals != 0
    pick, vals = pick[stored], vals[stored]
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(users[pick], minlength=n), out=indptr[1:])
    return indptr, items[pick].astype(np.int32), vals
