"""Seeded synthetic rating problems shared by the parity tests."""
import numpy as np


def random_problem(seed, n, m, nnz, rank, levels=(1, 2, 3, 4, 5), empty_rows=(), empty_cols=(),
                   long_row=None):
    """CSR + CSC of nnz distinct entries and random factor tables.

    long_row = (row, length) plants one user who rated `length` items."""
    rng = np.random.default_rng(seed)
    flat = rng.choice(n * m, size=nnz, replace=False)
    users, items = flat // m, flat % m
    if long_row is not None:
        r, length = long_row
        extra = rng.choice(m, size=length, replace=False)
        keep_other = users != r
        users = np.concatenate([users[keep_other], np.full(length, r)])
        items = np.concatenate([items[keep_other], extra])
    keep = ~(np.isin(users, empty_rows) | np.isin(items, empty_cols))
    users, items = users[keep], items[keep]
    order = np.lexsort((items, users))
    users, items = users[order].astype(np.int32), items[order].astype(np.int32)
    vals = rng.choice(np.asarray(levels, dtype=np.float64), size=users.size)
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(users, minlength=n), out=indptr[1:])
    corder = np.lexsort((users, items))
    cptr = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(np.bincount(items, minlength=m), out=cptr[1:])
    U = rng.standard_normal((rank, n)) * 0.5 + 0.3
    V = rng.standard_normal((rank, m)) * 0.5 + 0.3
    return dict(n=n, m=m, rank=rank, users=users, items=items, vals=vals, indptr=indptr,
                indices=items, cptr=cptr, cidx=users[corder], cvals=vals[corder], U=U, V=V)
