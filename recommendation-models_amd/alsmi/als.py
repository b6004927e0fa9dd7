"""ALS-WR matrix factorisation estimator on MI355X.

Same estimator surface as the reference's ``src/als.py`` ``ALS`` class
(constructor arguments, ``fit / fit_transform / _predict / predict_one /
predict_all / score / update_user / add_user``, attributes ``user_feats``,
``item_feats``, ``data``, ``reconstruction_err_``); the per-row normal equations,
their solves and every pair prediction run in libalsmi's HIP kernels.
"""
import os
import warnings

import numpy as np
import scipy.sparse as sps
from sklearn.base import BaseEstimator
from sklearn.utils.validation import check_is_fitted, check_random_state

from . import device as dev
from .engine import AlsEngine, default_group as _default_group
from .utils import _check_x, _check_y, root_mean_squared_error


class ALS(BaseEstimator):
    """Alternating least squares with weighted-lambda regularisation.

    Parameters
    ----------
    rank, alpha, tol, random_state, n_jobs, verbose
        As in the reference (src/als.py:75-85).  ``n_jobs`` is accepted for
        compatibility; the rows of a half-sweep are spread over the GPU (and over
        the ranks of an initialised ``torch.distributed`` group) instead of joblib
        workers.
    dtype : {'float64', 'float32'}
        Arithmetic type of the device path.  float64 reproduces the reference to
        ~1e-10; float32 is the throughput mode (factors within ~1e-4).
    max_iter : int or None
        When given, run exactly this many sweeps instead of the reference's
        ``while diff > tol`` rule.
    """

    def __init__(self, rank=10, alpha=0.1, tol=0.001, random_state=None,
                 n_jobs=1, verbose=0, dtype='float64', max_iter=None):
        self.rank = rank
        self.alpha = alpha
        self.tol = tol
        self.random_state = random_state
        if n_jobs == -1:
            n_jobs = os.cpu_count()
        self.n_jobs = n_jobs
        self.verbose = verbose
        self.dtype = dtype
        self.max_iter = max_iter

    # ------------------------------------------------------------------ fit
    def fit(self, X, y, shape=None):
        """Fit the model; see ``fit_transform`` (src/als.py:87-107)."""
        _, _ = self.fit_transform(X, y, shape=shape)
        return self

    # ``data`` (the ratings as a scipy CSR matrix, src/als.py:142) is built on the device; the
    # host copy is only materialised when something reads it.
    @property
    def data(self):
        if getattr(self, '_data', None) is None and getattr(self, '_data_dev', None) is not None:
            (indptr, indices, vals), shape = self._data_dev
            self._data = sps.csr_matrix((vals.cpu().numpy(), indices.cpu().numpy(),
                                         indptr.cpu().numpy()), shape=shape)
            self._data_dev = None
        return getattr(self, '_data', None)

    @data.setter
    def data(self, value):
        self._data = value
        self._data_dev = None

    def __getstate__(self):
        _ = self.data  # pickles carry the host matrix, never device tensors
        state = dict(self.__dict__)
        state.pop('_data_dev', None)
        state.pop('_tables', None)
        return state

    def _build_data(self, X, y, shape):
        """Ratings in canonical CSR form (src/als.py:132-142).  Returns
        ``(csr, shape, item_avg, dev_pairs)``: csr = (indptr, indices, vals) on the device for
        triplet input (then dev_pairs holds the uploaded triplets) or on the host for a sparse
        ``y``; item_avg = mean stored rating per item (0 for an unrated one)."""
        if (y.ndim < 2 or y.shape[0] == 1) and not shape:
            raise ValueError('When y is a scalar or 1-D array shape must be' +
                             'provided.')
        users, items = _check_x(X)
        if not sps.issparse(y):
            shape = tuple(int(v) for v in shape)
            pairs = dev.upload_triplets(users, items, y, shape, dev.require_gpu())
            return self._build_from_device_triplets(pairs, shape)
        data = y.tocsr().astype(np.float64)
        if data is y:
            data = data.copy()
        data.sum_duplicates()
        data.eliminate_zeros()
        data.sort_indices()
        m = data.shape[1]
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            counts = np.bincount(data.indices, minlength=m)
            sums = np.bincount(data.indices, weights=data.data, minlength=m)
            with np.errstate(invalid='ignore', divide='ignore'):
                item_avg = sums / counts
        item_avg[np.isnan(item_avg)] = 0
        self.data = data
        return (data.indptr, data.indices, data.data), data.shape, item_avg, None

    def _build_from_device_triplets(self, pairs, shape):
        """(users int32, items int32, y float64) device tensors -> the tuple `_build_data`
        returns: CSR on the device (last write wins, zeros dropped, sorted columns)."""
        import torch
        csr = dev.build_csr_device(*pairs, shape, np.float64)
        m = shape[1]
        # per-item sums from the item-major copy, one segment per item: the same from run to
        # run, and about a millisecond for 20 M ratings where adding into the m sums with
        # atomics (index_add_) took 128 ms on 26 744 items and changed in the last bits
        if int(csr[1].numel()):
            cptr, _, cvals = dev.csr_to_csc(csr[0], csr[1], csr[2], shape)
            counts = (cptr[1:] - cptr[:-1]).to(torch.float64)
            sums = torch.segment_reduce(cvals, 'sum', offsets=cptr, initial=0.0)
        else:
            counts = torch.zeros(m, dtype=torch.float64, device=csr[0].device)
            sums = torch.zeros_like(counts)
        item_avg = torch.where(counts > 0, sums / counts.clamp_min(1),
                               torch.zeros_like(sums)).cpu().numpy()
        self._data, self._data_dev = None, (csr, shape)
        return csr, shape, item_avg, pairs

    def fit_device_triplets(self, users, items, y, shape):
        """``fit`` for ratings that are already on the device: int32 user and item ids and
        float64 values, torch tensors on the current GPU, indices inside `shape` (DirtyIMC
        hands over the residual of its IMC stage this way, src/dimc.py:155-160, without a
        round trip through the host).  Same semantics as the 1-D ``y`` + ``shape`` form."""
        import torch
        shape = tuple(int(v) for v in shape)
        if users.dtype != torch.int32 or items.dtype != torch.int32 or y.dtype != torch.float64:
            raise ValueError('users/items must be int32 and y float64 device tensors')
        if not (users.numel() == items.numel() == y.numel()):
            raise ValueError('users, items and y must have the same length')
        self._fit_built(self._build_from_device_triplets((users, items, y), shape), None, None)
        return self

    def fit_transform(self, X, y, shape=None):
        """Factorise the ratings; returns ``(user_feats, item_feats)`` of shapes
        (rank, n_users) and (rank, n_items) (src/als.py:109-171)."""
        return self._fit_built(self._build_data(X, y, shape), X, y)

    def _fit_built(self, built, X, y):
        """The sweep loop on ratings in canonical form (src/als.py:143-171)."""
        csr, (n, m), item_avg, dev_pairs = built
        engine = AlsEngine.from_global(csr[0], csr[1], csr[2], (n, m), self.rank,
                                       self.alpha, dtype=self.dtype, group=_default_group())
        self._sweep_loop(engine, item_avg)
        if dev_pairs is not None and dev_pairs[0].numel() > 0:
            # score(X, y) (src/als.py:169) on the triplets and factor tables still on the device
            du, di, dy = dev_pairs
            _, sse = dev.sddmm(self.rank, du, di, engine.U, engine.V, dy.to(engine.U.dtype),
                               want_pred=False)
            self.reconstruction_err_ = -float(np.sqrt(float(sse.cpu()[0]) / du.numel()))
        elif X is None:
            self.reconstruction_err_ = float('nan')  # fit_device_triplets without a sample
        else:
            self.reconstruction_err_ = self.score(X, y)
        return self.user_feats, self.item_feats

    def fit_sharded(self, users, items, y, shape):
        """``fit`` when every rank of the initialised ``torch.distributed`` group holds only a
        PART of the ratings (config-4-sized problems never exist on one rank): `users`, `items`,
        `y` are this rank's share of the distinct stored (user, item, rating) triplets, host
        arrays or tensors on the rank's GPU, zero ratings not allowed.  The ranks exchange the
        entries so that each ends with the CSR slice of its user range and the CSC slice of its
        item range (AlsEngine.from_shards), run the reference's loop (src/als.py:143-169) on
        them and all end with the full ``user_feats`` / ``item_feats``.  ``data`` is not
        materialised (``update_user`` / ``add_user`` need a fit through ``fit``);
        ``reconstruction_err_`` is the score over all ranks' triplets."""
        import torch
        device = dev.require_gpu()
        shape = tuple(int(v) for v in shape)
        as_t = lambda a, dt: (a if torch.is_tensor(a) else
                              torch.from_numpy(np.ascontiguousarray(a))).to(device=device, dtype=dt)
        du, di, dy = as_t(users, torch.int64), as_t(items, torch.int64), as_t(y, torch.float64)
        if dy.numel() and bool((dy == 0).any()):
            raise ValueError('fit_sharded: zero ratings are not stored entries; drop them')
        group = _default_group()
        engine = AlsEngine.from_shards(du, di, dy, shape, self.rank, self.alpha, dtype=self.dtype,
                                       group=group)
        self._data, self._data_dev = None, None
        self._sweep_loop(engine, engine.item_means().cpu().numpy())
        _, sse = dev.sddmm(self.rank, du.to(torch.int32), di.to(torch.int32), engine.U, engine.V,
                           dy.to(engine.U.dtype), want_pred=False) if du.numel() else (None, None)
        tot = torch.zeros(2, dtype=torch.float64, device=device)
        if du.numel():
            tot[0], tot[1] = sse[0], float(du.numel())
        if engine.collective:
            import torch.distributed as dist
            dist.all_reduce(tot, group=group)
        tot = tot.cpu().numpy()
        self.reconstruction_err_ = -float(np.sqrt(tot[0] / tot[1])) if tot[1] else float('nan')
        return self

    def _sweep_loop(self, engine, item_avg):
        """Initial factors and the ``while diff > tol`` loop (src/als.py:143-169) on an engine
        that holds the ratings; leaves ``user_feats``, ``item_feats``, ``rmse_history_``,
        ``n_iter_`` and the device tables behind."""
        n, m = engine.n, engine.m
        random_state = check_random_state(self.random_state)

        # initial factors: uniform random items with row 0 = mean rating of the
        # item, zero users (src/als.py:143-152); the MT19937 stream stays on the host
        self.item_feats = random_state.rand(self.rank, m)
        self.item_feats[0] = item_avg
        self.user_feats = np.zeros((self.rank, n))
        engine.set_factors(self.user_feats, self.item_feats)
        # The error of sweep s is produced by the user half of sweep s+1, which reads the
        # same ratings and item factors anyway; its new user rows stay parked until the
        # stopping rule (src/als.py:154, 161-167) has seen that error, so stopping returns
        # exactly the factors of sweep s as the reference does.
        rmse = float('inf')
        sweeps = 0
        self.rmse_history_ = []
        nnz = engine.nnz

        sum_sq = None

        def to_rmse(sse, recheck=False):
            """sqrt(sse / nnz).  The fused error is a difference of three sums (sum r^2 -
            2 u.b + u^T G u); in float32 on a near-exact fit they cancel to rounding noise,
            possibly below zero.  When it falls under 1e-5 of sum r^2 (or is not finite) and the factors it
            belongs to are still in place (`recheck`), the standalone pass (fp64 accumulation
            of per-entry residuals) replaces it; otherwise it is clamped at zero."""
            nonlocal sum_sq
            if not nnz:
                return float('nan')
            value = float(sse.cpu()[0])
            if self.dtype == 'float32' or np.dtype(self.dtype) == np.float32:
                if sum_sq is None:
                    sum_sq = engine.sum_sq_ratings()
                if not np.isfinite(value) or value < 1e-5 * sum_sq:
                    if recheck:
                        return engine.train_rmse()
                    value = max(value, 0.0)
            return float(np.sqrt(max(value, 0.0) / nnz))

        def record(new_rmse):
            self.rmse_history_.append(new_rmse)
            if self.verbose:
                print('sweep %d: train rmse %.6f' % (sweeps, new_rmse))

        by_tol = self.max_iter is None
        go = by_tol or self.max_iter > 0  # the reference always runs the first sweep
        if go:
            engine.half_sweep_users()
            engine.half_sweep_items()
            sweeps = 1
        pending = go
        while pending:
            if not by_tol and sweeps >= self.max_iter:
                break
            new_rmse = to_rmse(engine.half_sweep_users_err(keep_old=by_tol), recheck=by_tol)
            record(new_rmse)
            diff = rmse - new_rmse
            rmse = new_rmse
            if by_tol and (not diff > self.tol or not np.isfinite(new_rmse)):
                engine.discard_users()
                pending = False
                break
            if not by_tol and not np.isfinite(new_rmse):
                # a fixed sweep count: the fused float32 error (a difference of three sums)
                # can overflow where the factors are fine, and the factors it belonged to are
                # overwritten already.  Keep the NaN in the history and go on; stop only when
                # the factors themselves are no longer finite.
                import torch
                if not bool(torch.isfinite(engine.U).all()) or \
                        not bool(torch.isfinite(engine.V).all()):
                    pending = False
                    break
            if by_tol:
                engine.commit_users()
            engine.half_sweep_items()
            sweeps += 1
        if pending:  # the last sweep's error was not needed to decide anything
            record(engine.train_rmse())
        self.n_iter_ = sweeps
        self.user_feats, self.item_feats = engine.get_factors()
        self.shape_ = (n, m)
        # the device tables stay for predictions (dropped when the host factors change)
        self._remember_tables(engine.U, engine.V)

    # -------------------------------------------------------------- predict
    def _pair_predictions(self, users, items):
        device = dev.require_gpu()
        import torch
        users = np.asarray(users).ravel()
        items = np.asarray(items).ravel()
        n, m = self.user_feats.shape[1], self.item_feats.shape[1]
        if users.size and (users.min() < -n or users.max() >= n or
                           items.min() < -m or items.max() >= m):
            raise IndexError('user/item index out of range')
        users = np.where(users < 0, users + n, users).astype(np.int32)
        items = np.where(items < 0, items + m, items).astype(np.int32)
        tables = self._cached_tables()
        if tables is None:
            tables = self._remember_tables(
                dev.factors_to_device(self.user_feats, self.dtype, device),
                dev.factors_to_device(self.item_feats, self.dtype, device))
        U, V = tables[:2]
        pred, _ = dev.sddmm(self.rank, torch.from_numpy(users).to(device),
                            torch.from_numpy(items).to(device), U, V)
        return pred.to(torch.float64).cpu().numpy()

    @staticmethod
    def _fingerprint(feats):
        """A cheap signature of a (rank, n) host array: its shape and the sums of its first
        and last row and column (O(rank + n) instead of a pass over the whole array)."""
        feats = np.asarray(feats)
        if feats.size == 0:
            return (feats.shape,)
        return (feats.shape, float(feats[0].sum()), float(feats[-1].sum()),
                float(feats[:, 0].sum()), float(feats[:, -1].sum()))

    def _remember_tables(self, U, V):
        """Keep the device copies of the factor tables together with the host arrays they
        were made from (strong references: a later array cannot reuse their identity) and a
        signature of their contents."""
        self._tables = (U, V, self.user_feats, self.item_feats,
                        self._fingerprint(self.user_feats), self._fingerprint(self.item_feats))
        return self._tables

    def invalidate(self):
        """Drop the device copies of the factors (they are rebuilt on the next prediction, and
        until then hold rank x (users + items) elements of device memory).  Call it after
        editing ``user_feats`` / ``item_feats`` in place in a way the signature check cannot
        see: the check compares the array objects, their shapes and the sums of the first and
        last row and column."""
        self._tables = None

    def _cached_tables(self):
        """The device copies of the factor tables if they still belong to the host arrays
        (the same array objects with an unchanged signature), else None."""
        t = getattr(self, '_tables', None)
        if t is None or t[2] is not self.user_feats or t[3] is not self.item_feats or \
                t[4] != self._fingerprint(self.user_feats) or \
                t[5] != self._fingerprint(self.item_feats):
            return None
        return t

    def _predict(self, X):
        """Predictions for arrays of (user, item) pairs (src/als.py:302-321)."""
        check_is_fitted(self, ['item_feats', 'user_feats'])
        users, items = _check_x(X)
        return self._pair_predictions(users, items)

    def predict_one(self, user, item):
        """Prediction for one pair, as a length-1 array (src/als.py:323-345)."""
        return self._predict((np.array([user]), np.array([item])))

    def predict_all(self, user):
        """Predictions of every item for one user (src/als.py:347-364)."""
        m = self.item_feats.shape[1]  # (not self.data.shape: that would fetch the ratings)
        users = np.repeat(user, m)
        items = np.arange(m)
        return self._predict((users, items))

    def recommend(self, users, k=10):
        """The k items with the highest predicted rating for each of `users`: ``(items,
        scores)`` arrays of shape (len(users), k), best first (ties: smaller item id).
        The ranking the reference leaves to the caller of ``predict_all``
        (src/als.py:347-364), done on the device for a batch of users."""
        check_is_fitted(self, ['item_feats', 'user_feats'])
        import torch
        device = dev.require_gpu()
        users = np.atleast_1d(np.asarray(users)).ravel()
        n = self.user_feats.shape[1]
        if users.size and (users.min() < -n or users.max() >= n):
            raise IndexError('user index out of range')
        users = np.where(users < 0, users + n, users).astype(np.int32)
        self._pair_predictions(users[:0], users[:0])  # makes sure the device tables exist
        U, V = self._cached_tables()[:2]
        items, scores = dev.topk_items(self.rank, torch.from_numpy(users).to(device), U, V, int(k))
        return items.cpu().numpy(), scores.to(torch.float64).cpu().numpy()

    def score(self, X, y):
        """NEGATIVE root mean squared error on the given pairs (src/als.py:366-388)."""
        check_is_fitted(self, ['item_feats', 'user_feats'])
        users, items = _check_x(X)
        r_ = _check_y(y, users, items)
        pred = self._pair_predictions(users, items)
        return -root_mean_squared_error(r_, pred)

    # ------------------------------------------------------ online updates
    def update_user(self, user, item, value):
        """Store a new rating and re-solve that user's factor vector against the
        current item factors.

        The reference delegates this to an external ``fit_als.py`` script that is
        not part of its repository (src/als.py:390-422); here it is the same
        single-row normal-equation solve the sweep uses, run on the device."""
        check_is_fitted(self, ['item_feats', 'user_feats'])
        device = dev.require_gpu()
        with warnings.catch_warnings():
            warnings.simplefilter('ignore', sps.SparseEfficiencyWarning)
            self.data[user, item] = value
        self.data.eliminate_zeros()
        self.data.sort_indices()
        row = self.data[user]
        side = dev.SparseSide(np.array([0, row.nnz], dtype=np.int64), row.indices, row.data, 1,
                              self.data.shape[1], self.dtype, device)
        plan = dev.SweepPlan(side, self.rank)
        tables = self._cached_tables()
        V = tables[1] if tables is not None else \
            dev.factors_to_device(self.item_feats, self.dtype, device)
        out = V.new_zeros((1, V.shape[1]))
        plan.half_sweep(V, out, self.alpha)
        self.user_feats[:, user] = dev.factors_to_host(out, self.rank)[:, 0]
        self._tables = None  # the host factors changed: rebuild the device copy on next use
        plan.close()

    def add_user(self):
        """Append an empty user: one more row of ``data`` and a zero column of
        ``user_feats`` (src/als.py:424-439)."""
        check_is_fitted(self, ['item_feats', 'user_feats'])
        shape = self.data.shape
        self.data = sps.vstack([self.data, sps.csr_matrix((1, shape[1]))], format='csr')
        new_col = np.zeros((self.rank, 1))
        self.user_feats = np.hstack((self.user_feats, new_col))
        self._tables = None
