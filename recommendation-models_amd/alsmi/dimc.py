"""Dirty inductive matrix completion: IMC on the side features, then ALS on the
residual (src/dimc.py).  Same estimator surface as the reference's ``DirtyIMC``;
both stages run on the device through ``IMC`` and ``ALS``.
"""
import numpy as np
from scipy.sparse import issparse
from sklearn.base import BaseEstimator
from sklearn.utils.validation import check_is_fitted

from .als import ALS
from .engine import default_group
from .imc import IMC, ImcProblem, problem_from_samples
from .utils import _check_x, _check_y, root_mean_squared_error


class DirtyIMC(BaseEstimator):
    """R ~ X Z Y^T + U^T V (src/dimc.py:21-89).

    X is ``(x, y_, users, items)`` or a 4-field DataHolder.  The reference's form: ``x`` and
    ``y_`` hold one feature row per SAMPLE.  Also accepted (``feature_form``): the attribute
    form, ``x`` one row per USER and ``y_`` one row per ITEM, indexed by ``users`` /
    ``items`` — nothing is expanded to per-sample rows, which at 100 M samples x 256
    attributes would be 205 GB.  ``feature_form='auto'`` takes the attribute form when the
    row counts of ``x`` / ``y_`` differ from the sample count and equal the matrix shape.
    """

    def __init__(self, rank=10, n_components=None, method='Newton-CG',
                 als_alpha=0.1, imc_alpha=0.1, tol=0.1, random_state=None,
                 n_jobs=1, verbose=0, feature_form='auto'):
        self.rank = rank
        self.n_components = n_components
        self.method = method
        self.als_alpha = als_alpha
        self.imc_alpha = imc_alpha
        self.tol = tol
        self.random_state = random_state
        self.n_jobs = n_jobs
        self.verbose = verbose
        self.feature_form = feature_form

    def fit(self, X, y, Z=None, shape=None):
        """src/dimc.py:91-114."""
        _, _, _ = self.fit_transform(X, y, Z=Z, shape=shape)
        return self

    def _attribute_form(self, x, y_, users, items, shape=None):
        """True when x / y_ hold one row per user / item rather than one per sample."""
        if self.feature_form in ('attribute', 'sample'):
            return self.feature_form == 'attribute'
        if self.feature_form != 'auto':
            raise ValueError("feature_form must be 'auto', 'sample' or 'attribute'")
        n_s = np.asarray(users).size
        nx, ny = np.atleast_2d(x).shape[0], np.atleast_2d(y_).shape[0]
        if nx == n_s and ny == n_s:
            return False  # the reference's form (also when the counts happen to coincide)
        if shape is None:
            shape = getattr(getattr(self, 'als', None), 'shape_', None)
        if shape is not None and (nx, ny) != tuple(int(v) for v in shape):
            raise ValueError('x / y_ have %d / %d rows: neither one per sample (%d) nor one '
                             'per user / item %r' % (nx, ny, n_s, tuple(shape)))
        return True

    def _problem(self, x, y_, users, items, ratings, attribute, group=None):
        if attribute:
            return ImcProblem(x, y_, users, items, ratings, group=group)
        return problem_from_samples(x, y_, ratings)

    def _xzy(self, x, y_, users, items, Z, attribute):
        """x_e^T Z y_e per sample (src/dimc.py:156-157, :184) on the device."""
        n_s = np.asarray(users).size
        return self._problem(x, y_, users, items, np.zeros(n_s), attribute).predict_samples(Z)

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
        if not shape:
            shape = y.shape
        attribute = self._attribute_form(x, y_, users, items, shape)
        prob = self._problem(x, y_, users, items, r_, attribute, group=default_group())
        self.imc._fit_on(prob, Z)
        self.als.shape_ = tuple(int(v) for v in shape)
        if prob.group is None:
            # One rank: the residual r - x^T Z y (src/dimc.py:155-160) never leaves the device.
            if attribute:
                # The IMC problem already holds the samples there, grouped by user; that order
                # keeps the caller's order among the samples of one (user, item) pair, which is
                # all the reference's element-by-element fill (last write wins) depends on.
                du, di, dr = prob.grouped_triplets_dev()
            else:
                # per-sample feature rows: every sample is its own "user" of the IMC problem,
                # so its grouped order is the caller's order; the real index pairs go up once
                from . import device as dev
                du, di = dev.upload_pairs(users, items, self.als.shape_, prob.device)
                dr = prob.ratings.to(du.device, prob._torch.float64)
            resid = dr - prob.predict_samples_grouped_dev(self.imc.Z)
            self.als.fit_device_triplets(du, di, resid, self.als.shape_)
            # score(X, y) (src/dimc.py:161): r - (uv + xzy) = residual - uv, and the ALS stage
            # has just scored its own predictions against exactly that residual
            self.reconstruction_err_ = self.als.reconstruction_err_
            return self.imc.Z, self.als.user_feats, self.als.item_feats
        xzy = prob.predict_samples(self.imc.Z)
        self.als.fit((users, items), r_ - xzy, shape=shape)
        # score(X, y) (src/dimc.py:161) from the parts at hand: the same samples' x^T Z y are in
        # xzy already, so the IMC problem is not grouped and uploaded a second time
        uv = self.als._pair_predictions(users, items)
        self.reconstruction_err_ = -root_mean_squared_error(r_, uv + xzy)
        return self.imc.Z, self.als.user_feats, self.als.item_feats

    def fit_sharded(self, X, y, shape, Z=None):
        """``fit`` when every rank of the initialised ``torch.distributed`` group holds only
        its OWN samples (attribute form: X = (x one row per user, y_ one row per item, users,
        items), y this rank's 1-D ratings, the distinct (user, item) pairs of the whole
        problem split over the ranks in any way).  IMC on the union of the samples (each rank
        evaluates its share, the p x q results are all-reduced), then ALS on the residual
        r - x^T Z y (src/dimc.py:155-160), whose entries travel to the owners of their user and
        item rows (ALS.fit_sharded); residuals that are exactly zero are not stored, as in the
        reference's element-by-element fill.  ``reconstruction_err_`` is the score over all
        ranks' samples."""
        self.als = ALS(rank=self.rank, alpha=self.als_alpha, tol=self.tol,
                       random_state=self.random_state, n_jobs=self.n_jobs,
                       verbose=self.verbose)
        self.imc = IMC(n_components=self.n_components, method=self.method,
                       alpha=self.imc_alpha, verbose=self.verbose)
        x, y_, users, items = _check_x(X, indices=True)
        users, items = np.asarray(users).ravel(), np.asarray(items).ravel()
        r_ = np.asarray(y, dtype=np.float64).ravel()
        prob = self.imc.fit_sharded((x, y_), users, items, r_, Z=Z)
        resid = r_ - prob.predict_samples(self.imc.Z)
        keep = resid != 0
        self.als.fit_sharded(users[keep], items[keep], resid[keep], shape)
        # score(X, y) (src/dimc.py:161): r - (uv + xzy) = residual - uv over all ranks' samples
        import torch
        from . import device as dev
        uv = self.als._pair_predictions(users, items)
        tot = torch.tensor([float(((resid - uv) ** 2).sum()), float(r_.size)],
                           dtype=torch.float64, device=dev.require_gpu())
        if prob.group is not None:
            import torch.distributed as dist
            dist.all_reduce(tot, group=prob.group)
        tot = tot.cpu().numpy()
        self.reconstruction_err_ = -float(np.sqrt(tot[0] / tot[1])) if tot[1] else float('nan')
        return self

    def predict(self, X):
        """uv + xzy for the given samples (src/dimc.py:164-186)."""
        check_is_fitted(self, ['imc', 'als'])
        x, y_, users, items = _check_x(X, indices=True)
        uv = self.als._pair_predictions(users, items)
        xzy = self._xzy(x, y_, users, items, self.imc.Z,
                        self._attribute_form(x, y_, users, items))
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
