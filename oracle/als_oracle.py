"""CPU restatement of the reference ALS path (numpy, fp64).  TEST INFRASTRUCTURE.

This module is the checker the HIP path is compared against.  It is only ever
imported by ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py``; the product package never imports it and has no CPU
fallback.

Parity is pinned: ``tests/test_oracle_golden.py`` checks every function here
against vectors produced by running the upstream reference itself
(``tests/golden/make_golden.py``) and against the known-answer values of the
reference's own ``tests/test_als.py``.

Every function cites the reference lines it restates (paths relative to the
reference checkout).
"""
import numpy as np


# --------------------------------------------------------------------------
# sparse structure (src/als.py:136-142, :268-280)
# --------------------------------------------------------------------------

def build_csr(users, items, y, shape):
    """Ratings triplets -> CSR with sorted column indices.

    Restates the dense-``y`` branch of ``ALS.fit_transform``
    (src/als.py:137-140): entries are written one by one into a
    ``lil_matrix`` and converted with ``tocsr()``.  Observable semantics:
    duplicates -> the LAST write wins; a final value of exactly 0 is not
    stored; indices inside a row come out sorted.
    """
    n, m = int(shape[0]), int(shape[1])
    users = np.asarray(users).astype(np.int64).ravel()
    items = np.asarray(items).astype(np.int64).ravel()
    y = np.asarray(y).ravel()
    if users.size and (users.min() < 0 or users.max() >= n or
                       items.min() < 0 or items.max() >= m):
        raise IndexError('index out of bounds')
    key = users * m + items
    # stable sort, keep the last occurrence of each key
    order = np.argsort(key, kind='stable')
    ks = key[order]
    last = np.ones(ks.size, dtype=bool)
    last[:-1] = ks[1:] != ks[:-1]
    sel = order[last]
    vals = y[sel].astype(np.float64)
    keep = vals != 0
    sel, vals = sel[keep], vals[keep]
    indices = items[sel].astype(np.int32)
    counts = np.bincount(users[sel], minlength=n)
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(counts, out=indptr[1:])
    return indptr, indices, vals


def csr_to_csc(n, m, indptr, indices, vals):
    """CSR -> CSC exactly as ``scipy.sparse._sparsetools.csr_tocsc`` does for
    src/als.py:275-279: count per column, exclusive scan, scatter rows in
    increasing row order (so row ids inside a column come out sorted)."""
    nnz = int(indptr[n])
    counts = np.bincount(indices[:nnz], minlength=m)
    cptr = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(counts, out=cptr[1:])
    rows = np.repeat(np.arange(n, dtype=np.int32), np.diff(indptr))
    order = np.argsort(indices[:nnz], kind='stable')
    return cptr, rows[order].astype(np.int32), np.asarray(vals)[order]


# --------------------------------------------------------------------------
# initialisation (src/als.py:143-152)
# --------------------------------------------------------------------------

def init_factors(n, m, indptr, indices, vals, rank, seed):
    """``item_feats = RandomState(seed).rand(rank, m)`` with row 0 replaced by
    the per-item mean rating (NaN -> 0); ``user_feats = 0``.  Layout is the
    reference's: (rank, n) and (rank, m), C order."""
    nnz = int(indptr[n])
    sums = np.bincount(indices[:nnz], weights=vals[:nnz], minlength=m)
    cnts = np.bincount(indices[:nnz], minlength=m)
    with np.errstate(invalid='ignore', divide='ignore'):
        item_avg = sums / cnts
    item_avg[np.isnan(item_avg)] = 0
    if isinstance(seed, np.random.RandomState):
        rs = seed
    else:
        rs = np.random.RandomState(seed)
    item_feats = rs.rand(rank, m)
    item_feats[0] = item_avg
    user_feats = np.zeros((rank, n))
    return user_feats, item_feats


# --------------------------------------------------------------------------
# the update (src/als.py:218-250)
# --------------------------------------------------------------------------

def update_one(submat, row, rank, alpha):
    """One normal-equation solve: ``inv(S S^T + alpha n I) (S r)``; a singular
    matrix (e.g. an empty row) gives zeros (src/als.py:243-250)."""
    num_ratings = row.size
    reg_sums = submat.dot(submat.T) + alpha * num_ratings * np.eye(rank)
    feature_sums = submat.dot(row[np.newaxis].T)
    try:
        col = np.linalg.inv(reg_sums).dot(feature_sums)
    except np.linalg.LinAlgError:
        col = np.zeros((1, rank))
    return col.ravel()


def half_sweep(indptr, indices, vals, src_feats, rank, alpha,
               row_begin=0, row_end=None):
    """Update every row of one side from the other side's factors
    (src/als.py:173-216 with make_item_submats/make_user_submats, :252-300).

    ``indptr/indices/vals`` is the CSR of the side being updated (pass the CSC
    of the ratings for the item side).  ``src_feats`` is (rank, n_src).
    Returns (rank, row_end-row_begin)."""
    n_rows = indptr.size - 1
    if row_end is None:
        row_end = n_rows
    out = np.zeros((rank, row_end - row_begin))
    for r in range(row_begin, row_end):
        lo, hi = int(indptr[r]), int(indptr[r + 1])
        submat = src_feats[:, indices[lo:hi]]
        out[:, r - row_begin] = update_one(submat, vals[lo:hi], rank, alpha)
    return out


# --------------------------------------------------------------------------
# predictions and error (src/als.py:161-165, :316-321, :383-388;
# src/utils.py:22-24)
# --------------------------------------------------------------------------

def predict_pairs(user_feats, item_feats, users, items):
    return (user_feats.T[users] * item_feats.T[items]).sum(-1)


def rmse(true, pred):
    true = np.asarray(true, dtype=np.float64)
    pred = np.asarray(pred, dtype=np.float64)
    return np.sqrt(np.mean((true - pred) ** 2)) if true.size else np.nan


def train_sse(indptr, indices, vals, user_feats, item_feats):
    """Sum of squared errors over the stored entries, CSR order
    (src/als.py:161-165 before the mean/sqrt)."""
    n = indptr.size - 1
    users = np.repeat(np.arange(n), np.diff(indptr))
    pred = predict_pairs(user_feats, item_feats, users, indices)
    return float(np.sum((vals - pred) ** 2))


# --------------------------------------------------------------------------
# the fit loop (src/als.py:143-171)
# --------------------------------------------------------------------------

def als_fit(users, items, y, shape, rank=10, alpha=0.1, tol=0.001, seed=None,
            max_sweeps=None, history=None):
    """Restates ``ALS.fit_transform``: ``while diff > tol`` with
    ``diff = rmse_prev - rmse_new`` starting from inf (so at least two
    sweeps), or exactly ``max_sweeps`` sweeps when given.

    Returns (user_feats, item_feats, rmse_per_sweep, csr)."""
    n, m = int(shape[0]), int(shape[1])
    indptr, indices, vals = build_csr(users, items, y, shape)
    cptr, cidx, cvals = csr_to_csc(n, m, indptr, indices, vals)
    uf, itf = init_factors(n, m, indptr, indices, vals, rank, seed)
    prev = float('inf')
    diff = prev
    hist = []
    sweeps = 0
    while (diff > tol) if max_sweeps is None else (sweeps < max_sweeps):
        uf = half_sweep(indptr, indices, vals, itf, rank, alpha)
        itf = half_sweep(cptr, cidx, cvals, uf, rank, alpha)
        nnz = int(indptr[n])
        new = np.sqrt(train_sse(indptr, indices, vals, uf, itf) / nnz)
        diff = prev - new
        prev = new
        hist.append(new)
        sweeps += 1
        if history is not None:
            history.append((uf.copy(), itf.copy()))
    return uf, itf, np.array(hist), (indptr, indices, vals)


def topk_items(user_feats, item_feats, users, k):
    """The k best items per user from the dense scores of src/als.py:361-364 (one row of
    U^T V per user), best first, ties by the smaller item id.  Returns (items, scores)."""
    scores = user_feats[:, users].T.dot(item_feats)
    order = np.lexsort((np.broadcast_to(np.arange(scores.shape[1]), scores.shape), -scores),
                       axis=1)[:, :k]
    return order, np.take_along_axis(scores, order, axis=1)
