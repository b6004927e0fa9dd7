"""CPU restatement of the reference IMC / DirtyIMC path (numpy, fp64).
TEST INFRASTRUCTURE — see ``oracle/als_oracle.py`` for who may import it.

Parity is pinned against vectors produced by running the upstream reference
(``tests/golden/imc_unit.npz``, ``imc_random.npz``, ``dimc.npz``) and the
known-answer values of the reference's ``tests/test_imc.py:53-80``.

Two equivalent forms are given: the per-sample form the reference computes in
(``users`` is n_s x p, ``items`` is n_s x q) and the index form the HIP path
computes in (attribute matrices + one (user, item) index pair per sample).
"""
import warnings

import numpy as np
import scipy.optimize as so

from . import als_oracle


# ---- per-sample form: literal restatement -------------------------------

def cost(arr, users, items, ratings, lam, shape):
    """src/imc.py:43-48.  Quirk kept: ``lam * ||Z||^2`` sits inside the sum
    over samples, so it is counted n_samples times."""
    z_arr = arr.reshape(shape)
    preds = (users.dot(z_arr) * items).sum(-1)
    return 0.5 * np.sum((ratings - preds) ** 2 +
                        lam * np.linalg.norm(z_arr) ** 2)


def cost_prime(arr, users, items, ratings, lam, shape):
    """src/imc.py:65-71: ``X^T diag(preds) Y - (X^T * r) Y + lam Z``."""
    z_arr = arr.reshape(shape)
    diag = (users.dot(z_arr) * items).sum(-1)
    grad = (users.T * diag).dot(items) - (users.T * ratings).dot(items) \
        + lam * z_arr
    return grad.ravel()


def cost_hess(_, s_vec, users, items, ratings, lam, shape):
    """src/imc.py:91-96.  Quirk kept: the scalar ``lam`` is added to every
    entry (not ``lam * S``); pinned by tests/test_imc.py:70-80."""
    s_mat = s_vec.reshape(shape)
    diag = (users.dot(s_mat) * items).sum(-1)
    hess = (users.T * diag).dot(items) + lam
    return hess.ravel()


# ---- index form ----------------------------------------------------------

def preds_indexed(Z, X_attrs, Y_attrs, users, items):
    return (X_attrs.dot(Z)[users] * Y_attrs[items]).sum(-1)


def cost_indexed(Z, X_attrs, Y_attrs, users, items, ratings, lam):
    preds = preds_indexed(Z, X_attrs, Y_attrs, users, items)
    return 0.5 * (np.sum((ratings - preds) ** 2) +
                  ratings.size * lam * np.sum(Z * Z))


def _xt_diag_y(X_attrs, Y_attrs, users, items, d):
    n, q = X_attrs.shape[0], Y_attrs.shape[1]
    W = np.zeros((n, q))
    np.add.at(W, users, d[:, None] * Y_attrs[items])
    return X_attrs.T.dot(W)


def grad_indexed(Z, X_attrs, Y_attrs, users, items, ratings, lam):
    preds = preds_indexed(Z, X_attrs, Y_attrs, users, items)
    return _xt_diag_y(X_attrs, Y_attrs, users, items, preds - ratings) \
        + lam * Z


def hessp_indexed(S, X_attrs, Y_attrs, users, items, lam):
    d = preds_indexed(S, X_attrs, Y_attrs, users, items)
    return _xt_diag_y(X_attrs, Y_attrs, users, items, d) + lam


def init_z(X_attrs, Y_attrs, users, items, ratings):
    """src/imc.py:157-159: ``X^T diag(R) Y / R.size`` pushed through an SVD
    and recomposed."""
    z0 = _xt_diag_y(X_attrs, Y_attrs, users, items,
                    np.asarray(ratings, dtype=np.float64)) / ratings.size
    u, s, v = np.linalg.svd(z0, False)
    return u.dot(np.diag(s)).dot(v)


# ---- fit (src/imc.py:99-170, :220-254) ------------------------------------

def fit_imc(X, Y, R, Z=None, method='Newton-CG', alpha=0.1):
    """Per-sample restatement of ``_fit_imc``; returns (Z, success, message)."""
    X = np.asarray(X)
    Y = np.asarray(Y)
    R = np.asarray(R)
    if Z is None:
        sum_x_r_y = (X.T * R).dot(Y) / R.size
        u, s, v = np.linalg.svd(sum_x_r_y, False)
        Z = u.dot(np.diag(s)).dot(v)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', category=RuntimeWarning)
        res = so.minimize(cost, Z.flatten(), method=method, jac=cost_prime,
                          hessp=cost_hess, args=(X, Y, R, alpha, Z.shape),
                          options={'disp': 0})
    return res['x'].reshape(Z.shape), res.success, res.message


def imc_score(Z, X, Y, R):
    """src/imc.py:350-356: POSITIVE rmse of per-sample predictions."""
    preds = (np.asarray(X).dot(Z) * np.asarray(Y)).sum(-1)
    return als_oracle.rmse(R, preds)


# ---- DirtyIMC (src/dimc.py:116-186) ---------------------------------------

def dimc_fit(x, y_, users, items, r_, shape, rank, als_alpha, imc_alpha, tol,
             seed, method='Newton-CG'):
    Z, _, _ = fit_imc(x, y_, r_, method=method, alpha=imc_alpha)
    xzy = (np.asarray(x).dot(Z) * np.asarray(y_)).sum(-1)
    uf, itf, _, _ = als_oracle.als_fit(users, items, r_ - xzy, shape,
                                       rank=rank, alpha=als_alpha, tol=tol,
                                       seed=seed)
    pred = als_oracle.predict_pairs(uf, itf, users, items) + xzy
    return Z, uf, itf, -als_oracle.rmse(r_, pred), pred


# ---- rank-constrained Z = U V^T (the HIP package's extension; NO reference counterpart:
# the reference validates n_components and ignores it, src/imc.py:148-154).  Parity of this
# part is UNPINNED: the restatement below is checked by finite differences only. ----------

def low_rank_cost_grad(U, V, X_attrs, Y_attrs, users, items, ratings, lam):
    """f = 0.5 (sum_e (r_e - x_u^T U V^T y_i)^2 + lam (||U||^2 + ||V||^2)) and its
    gradients (dU, dV)."""
    Z = U.dot(V.T)
    preds = preds_indexed(Z, X_attrs, Y_attrs, users, items)
    G = _xt_diag_y(X_attrs, Y_attrs, users, items, preds - ratings)
    f = 0.5 * (np.sum((ratings - preds) ** 2) + lam * (np.sum(U * U) + np.sum(V * V)))
    return f, G.dot(V) + lam * U, G.T.dot(U) + lam * V


def fit_low_rank(X_attrs, Y_attrs, users, items, ratings, r, lam, Z=None):
    """L-BFGS-B over (U, V) from the rank-r truncation of the initial Z."""
    p, q = X_attrs.shape[1], Y_attrs.shape[1]
    Z0 = init_z(X_attrs, Y_attrs, users, items, ratings) if Z is None else Z
    u, s, vt = np.linalg.svd(Z0, False)
    root = np.sqrt(s[:r])
    x0 = np.concatenate(((u[:, :r] * root).ravel(), (vt[:r].T * root).ravel()))

    def fun(a):
        U, V = a[:p * r].reshape(p, r), a[p * r:].reshape(q, r)
        f, gu, gv = low_rank_cost_grad(U, V, X_attrs, Y_attrs, users, items, ratings, lam)
        return f, np.concatenate((gu.ravel(), gv.ravel()))

    with warnings.catch_warnings():
        warnings.simplefilter('ignore', category=RuntimeWarning)
        res = so.minimize(fun, x0, jac=True, method='L-BFGS-B')
    U, V = res['x'][:p * r].reshape(p, r), res['x'][p * r:].reshape(q, r)
    return U.dot(V.T), U, V
