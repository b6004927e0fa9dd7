"""Dirty inductive matrix completion: IMC on the side features, then ALS on the
residual (src/dimc.py).  Same estimator surface as the reference's ``DirtyIMC``;
both stages run on the device through ``IMC`` and ``ALS``.
"""
import numpy as np
from scipy.sparse import issparse
from sklearn.base import BaseEstimator
from sklearn.utils.validation import check_is_fitted

from .als import ALS
from .imc import IMC, problem_from_samples
from .utils import _check_x, _check_y, root_mean_squared_error


class DirtyIMC(BaseEstimator):
    """R ~ X Z Y^T + U^T V (src/dimc.py:21-89).

    X is ``(x, y_, users, items)``: per-sample user features, per-sample item
    features and the (user, item) index of every sample, or a 4-field DataHolder.
    """

    def __init__(self, rank=10, n_components=None, method='Newton-CG',
                 als_alpha=0.1, imc_alpha=0.1, tol=0.1, random_state=None,
                 n_jobs=1, verbose=0):
        self.rank = rank
        self.n_components = n_components
        self.method = method
        self.als_alpha = als_alpha
        self.imc_alpha = imc_alpha
        self.tol = tol
        self.random_state = random_state
        self.n_jobs = n_jobs
        self.verbose = verbose

    def fit(self, X, y, Z=None, shape=None):
        """src/dimc.py:91-114."""
        _, _, _ = self.fit_transform(X, y, Z=Z, shape=shape)
        return self

    @staticmethod
    def _xzy(x, y_, Z):
        """x_e^T Z y_e per sample (src/dimc.py:156-157, :184) on the device."""
        prob = problem_from_samples(x, y_, np.zeros(np.atleast_2d(x).shape[0]))
        return prob.predict_samples(Z)

    def fit_transform(self, X, y, Z=None, shape=None):
        """Fit IMC, then ALS on what IMC leaves unexplained; returns
        ``(Z, user_feats, item_feats)`` (src/dimc.py:116-162)."""
        self.als = ALS(rank=self.rank, alpha=self.als_alpha, tol=self.tol,
                       random_state=self.random_state, n_jobs=self.n_jobs,
                       verbose=self.verbose)
        self.imc = IMC(n_components=self.n_components, method=self.method,
                       alpha=self.imc_alpha, verbose=self.verbose)
        x, y_, users, items = _check_x(X, indices=True)
        if (y.ndim < 2 or y.shape[0] == 1) and not shape:
            raise ValueError('When y is a scalar or 1-D array shape must be' +
                             'provided.')
        r_ = _check_y(y, users, items)
        self.imc.fit((x, y_), r_, Z=Z)
        xzy = self._xzy(x, y_, self.imc.Z)
        if not shape:
            shape = y.shape
        self.als.fit((users, items), r_ - xzy, shape=shape)
        self.reconstruction_err_ = self.score(X, y)
        return self.imc.Z, self.als.user_feats, self.als.item_feats

    def predict(self, X):
        """uv + xzy for the given samples (src/dimc.py:164-186)."""
        check_is_fitted(self, ['imc', 'als'])
        x, y_, users, items = _check_x(X, indices=True)
        uv = self.als._pair_predictions(users, items)
        xzy = self._xzy(x, y_, self.imc.Z)
        return uv + xzy

    def score(self, X, y):
        """NEGATIVE root mean squared error (src/dimc.py:188-213)."""
        if issparse(y):
            _, _, users, items = _check_x(X, indices=True)
            r_ = np.asarray(y.tocsr()[users, items]).ravel()
        else:
            r_ = y
        preds = self.predict(X)
        return -root_mean_squared_error(r_, preds)

    def update_user(self, user, item, value):
        """src/dimc.py:215-235."""
        self.als.update_user(user, item, value)

    def add_user(self):
        """src/dimc.py:237-247."""
        self.als.add_user()
