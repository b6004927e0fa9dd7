"""Inductive matrix completion (R ~ X Z Y^T) on MI355X.

Same surface as the reference's ``src/imc.py``: module functions ``_cost``,
``_cost_prime``, ``_cost_hess`` with the scipy.optimize callback signature
``args = (users, items, ratings, lam, shape)``, ``_fit_imc`` and the ``IMC``
estimator.  The optimiser loop stays scipy's (so iterates follow the reference);
every function / gradient / Hessian-vector evaluation is one pass of libalsmi's
SDDMM + SpMM + GEMM kernels over the samples.

Two input forms are accepted wherever the reference takes ``(X, y)``:
  * per-sample (what the current reference accepts, src/imc.py:241-242):
    ``X = (x, y_)`` with one feature row per sample and ``y`` the 1-D ratings;
  * attribute form (what the reference's tests/test_imc.py:127-210 and
    tests/test_utils.py:80-94 specify): ``X = (X_attrs, Y_attrs)`` with one row
    per user / item and ``y`` a sparse ratings matrix.  Nothing is expanded to
    per-sample rows on the device.
"""
import numbers
import warnings

import numpy as np
import scipy.optimize as so
from scipy.sparse import issparse
from sklearn.base import BaseEstimator
from sklearn.exceptions import ConvergenceWarning
from sklearn.utils.validation import check_is_fitted

from . import _lib
from . import device as dev
from .engine import default_group, force_collectives
from .utils import _check_x, root_mean_squared_error

INTEGER_TYPES = (numbers.Integral, np.integer)


class ImcProblem(object):
    """Device-resident IMC data in index form and the three evaluations on it."""

    def __init__(self, X_attrs, Y_attrs, users, items, ratings, dtype=np.float64, group=None,
                 shard=None):
        """group: a torch.distributed process group; each rank then keeps the samples of one
        contiguous user range (balanced by samples) and every evaluation ends with an
        all-reduce of the p x q result and of the scalar, so all ranks see identical values
        and drive identical optimiser trajectories.  shard = (index, count) selects such a
        range without a group (the caller combines the partial results itself)."""
        self._init_device()
        self.group = group
        if group is not None:
            import torch.distributed as dist
            shard = (dist.get_rank(group), dist.get_world_size(group))
            if shard[1] == 1 and not force_collectives():
                self.group, shard = None, None
        self.dtype = np.dtype(dtype)
        X_attrs = np.atleast_2d(np.asarray(X_attrs, dtype=np.float64))
        Y_attrs = np.atleast_2d(np.asarray(Y_attrs, dtype=np.float64))
        users = np.asarray(users).ravel().astype(np.int64)
        items = np.asarray(items).ravel().astype(np.int64)
        ratings = np.asarray(ratings, dtype=np.float64).ravel()
        if not (users.size == items.size == ratings.size):
            raise ValueError('users, items and ratings must have the same length')
        self.n_users, self.p = X_attrs.shape
        self.n_items, self.q = Y_attrs.shape
        if users.size and (users.min() < 0 or users.max() >= self.n_users or
                           items.min() < 0 or items.max() >= self.n_items):
            raise IndexError('sample index out of range')
        self.ld = self._padded_width(self.q)
        self.n_samples = int(users.size)
        self.ratings_host = ratings  # all samples, caller order (score needs every rank's)
        # group the samples by user; `order` maps grouped position -> caller position
        indptr = np.zeros(self.n_users + 1, dtype=np.int64)
        np.cumsum(np.bincount(users, minlength=self.n_users), out=indptr[1:])
        if users.size < 2 or bool(np.all(users[1:] >= users[:-1])):
            # already grouped (samples listed user by user, e.g. taken from a CSR matrix):
            # no sort and no reordering copies
            self.order = None
            items_g, ratings_g = items.astype(np.int32), ratings
        else:
            self.order = np.argsort(users, kind='stable')
            items_g, ratings_g = items[self.order].astype(np.int32), ratings[self.order]
        # this rank's users [ulo, uhi) and their samples [slo, shi) of the grouped order
        self.ulo, self.uhi, self.slo, self.shi = 0, self.n_users, 0, self.n_samples
        self._sample_bounds = np.array([0, self.n_samples], dtype=np.int64)
        if shard is not None:
            from .engine import split_by_nnz
            bounds = split_by_nnz(indptr, int(shard[1]))
            self._sample_bounds = indptr[bounds]  # every rank's samples of the grouped order
            self.ulo, self.uhi = int(bounds[shard[0]]), int(bounds[shard[0] + 1])
            self.slo, self.shi = int(indptr[self.ulo]), int(indptr[self.uhi])
            indptr = indptr[self.ulo:self.uhi + 1] - self.slo
            items_g, ratings_g = items_g[self.slo:self.shi], ratings_g[self.slo:self.shi]
            X_attrs = X_attrs[self.ulo:self.uhi]
            # keep only the item rows this rank's samples refer to
            uniq, inv = np.unique(items_g, return_inverse=True)
            Y_attrs, items_g = Y_attrs[uniq], inv.astype(np.int32)
        self.n_local_users = self.uhi - self.ulo
        self.n_local_samples = self.shi - self.slo
        if self.q == self.ld:
            ypad = Y_attrs  # the row stride needs no padding: no second host copy of the table
        else:
            ypad = np.zeros((Y_attrs.shape[0], self.ld))
            ypad[:, :self.q] = Y_attrs
        self._place(indptr, items_g, ratings_g, X_attrs, ypad)

    @classmethod
    def from_local_samples(cls, X_attrs, Y_attrs, users, items, ratings, dtype=np.float64,
                           group=None):
        """Every rank of `group` brings its OWN samples (any split of the samples over the
        ranks; no rank sees the others') together with the attribute rows of the users and
        items they refer to (the whole tables are accepted, only the referenced rows are
        uploaded).  Cost, gradient and Hessian-vector product are sums over the samples, so
        each rank evaluates its share and the p x q result and the scalar are all-reduced
        (src/imc.py:43-96 on the union of the samples); predictions stay with the rank that
        owns the sample, ``n_samples`` is the count over all ranks."""
        self = cls.__new__(cls)
        self._init_device()
        self.group = group
        if group is not None:
            import torch.distributed as dist
            if dist.get_world_size(group) == 1 and not force_collectives():
                self.group = None
        self.local = True
        self.dtype = np.dtype(dtype)
        X_attrs = np.atleast_2d(np.asarray(X_attrs, dtype=np.float64))
        Y_attrs = np.atleast_2d(np.asarray(Y_attrs, dtype=np.float64))
        users = np.asarray(users).ravel().astype(np.int64)
        items = np.asarray(items).ravel().astype(np.int64)
        ratings = np.asarray(ratings, dtype=np.float64).ravel()
        if not (users.size == items.size == ratings.size):
            raise ValueError('users, items and ratings must have the same length')
        self.n_users, self.p = X_attrs.shape
        self.n_items, self.q = Y_attrs.shape
        if users.size and (users.min() < 0 or users.max() >= self.n_users or
                           items.min() < 0 or items.max() >= self.n_items):
            raise IndexError('sample index out of range')
        self.ld = self._padded_width(self.q)
        self.ratings_host = ratings
        # this rank's users and items renumbered 0..; samples grouped by (local) user
        uniq_u, lu = np.unique(users, return_inverse=True)
        uniq_i, li = np.unique(items, return_inverse=True)
        self.order = None if users.size < 2 or bool(np.all(lu[1:] >= lu[:-1])) else \
            np.argsort(lu, kind='stable')
        if self.order is not None:
            lu, li, ratings = lu[self.order], li[self.order], ratings[self.order]
        indptr = np.zeros(uniq_u.size + 1, dtype=np.int64)
        np.cumsum(np.bincount(lu, minlength=uniq_u.size), out=indptr[1:])
        self.n_local_users, self.n_local_samples = int(uniq_u.size), int(users.size)
        self.ulo, self.uhi, self.slo, self.shi = 0, self.n_local_users, 0, self.n_local_samples
        self._sample_bounds = np.array([0, self.n_local_samples], dtype=np.int64)
        total = self._torch.tensor([float(users.size)], dtype=self._torch.float64,
                                   device=self.device)
        if self.group is not None:
            self._all_reduce(total)
        self.n_samples = int(total.cpu()[0])
        Y_loc = Y_attrs[uniq_i]
        if self.q != self.ld:
            ypad = np.zeros((Y_loc.shape[0], self.ld))
            ypad[:, :self.q] = Y_loc
            Y_loc = ypad
        self._place(indptr, li.astype(np.int32), ratings, X_attrs[uniq_u], Y_loc)
        return self

    @classmethod
    def from_device_csr(cls, indptr, items, ratings, X, Y, dtype=np.float64):
        """The samples as the CSR structure of the ratings, already on the device (torch
        tensors: indptr int64 [n_users+1], items int32, ratings, X [n_users x p],
        Y [n_items x q]); predictions come back in that stored order.  For problems built
        on the device (synthetic benchmarks); no host copy of the samples is made."""
        import torch
        self = cls.__new__(cls)
        self._init_device()
        self.group = None
        self.dtype = np.dtype(dtype)
        self.n_users, self.p = (int(v) for v in X.shape)
        self.n_items, self.q = (int(v) for v in Y.shape)
        self.ld = self._padded_width(self.q)
        self.n_samples = int(items.numel())
        self.ratings_host = None
        self.order = None
        self.ulo, self.uhi, self.slo, self.shi = 0, self.n_users, 0, self.n_samples
        self._sample_bounds = np.array([0, self.n_samples], dtype=np.int64)
        self.n_local_users, self.n_local_samples = self.n_users, self.n_samples
        td = dev.torch_dtype(self.dtype)
        self.indptr = indptr.to(self.device, torch.int64).contiguous()
        self.items = items.to(self.device, torch.int32).contiguous()
        self.ratings = ratings.to(self.device, td).contiguous()
        self.X = X.to(self.device, td).contiguous()
        self.Y = torch.zeros((self.n_items, self.ld), dtype=td, device=self.device)
        self.Y[:, :self.q] = Y.to(self.device, td)
        self._alloc_workspace()
        return self

    # ---- device side (the test suite's CPU stand-in overrides these four) -----------------
    def _init_device(self):
        import torch
        self._torch = torch
        self.device = dev.require_gpu()

    def _padded_width(self, q):
        ld = _lib.load().alsmi_imc_padded_width(int(q))
        if ld == 0:
            raise ValueError('item attribute count must be between 1 and 256, got %d' % q)
        return ld

    def _place(self, indptr, items_g, ratings_g, X_attrs, ypad):
        torch = self._torch
        td = dev.torch_dtype(self.dtype)
        to = lambda a, t: torch.from_numpy(np.ascontiguousarray(a)).to(self.device, t)
        self.indptr = to(indptr, torch.int64)
        self.items = to(items_g, torch.int32)
        self.ratings = to(ratings_g, td)
        self.X = to(X_attrs, td)
        self.Y = to(ypad, td)
        self._alloc_workspace()

    def _alloc_workspace(self):
        torch = self._torch
        self._td = dev.torch_dtype(self.dtype)
        nbytes = int(_lib.load().alsmi_imc_workspace_bytes(
            _lib.dtype_code(self.dtype), max(self.n_local_users, 1), self.n_local_samples, self.p,
            self.q))
        self.workspace = torch.empty(nbytes // 8 + 1, dtype=torch.float64, device=self.device)

    def _run(self, mode, Md, r, out, local_pred, sse):
        """One pass of the kernels over this rank's samples."""
        _lib.check(_lib.load().alsmi_imc_eval(
            _lib.dtype_code(self.dtype), int(mode), self.n_local_users, self.n_local_samples,
            self.p, self.q, dev._ptr(self.indptr), dev._ptr(self.items), dev._ptr(r), dev._ptr(self.X),
            dev._ptr(self.Y), dev._ptr(Md), dev._ptr(out), dev._ptr(local_pred),
            dev._ptr(sse), dev._ptr(self.workspace), dev._stream()), 'alsmi_imc_eval')

    def _all_reduce(self, t):
        """Sum of `t` over the ranks of the group, in place.  Several ranks rehearsing on one
        GPU under gloo go through a host copy (gloo hangs on device buffers that are reduced
        again and again, as the evaluation buffers are); RCCL reduces the device tensor."""
        import torch.distributed as dist
        if t.is_cuda and dist.get_backend(self.group) == 'gloo':
            host = t.cpu()
            dist.all_reduce(host, group=self.group)
            t.copy_(host)
        else:
            dist.all_reduce(t, group=self.group)

    def _eval_dev(self, mode, M, want_out=True, want_sse=False, want_pred=False, ratings=None):
        """One evaluation, results left on the device: (out [p x q] or None, sse [1] fp64 or
        None, predictions of ALL samples in the grouped order or None); None for everything
        when there are no samples."""
        torch = self._torch
        M = np.ascontiguousarray(np.asarray(M, dtype=np.float64).reshape(self.p, self.q))
        # the p x q operand and result and the scalar live in buffers kept across evaluations
        # (an optimiser calls this hundreds of times); only the values cross the host boundary
        bufs = getattr(self, '_eval_bufs', None)
        if bufs is None:
            bufs = self._eval_bufs = (
                torch.empty((self.p, self.q), dtype=self._td, device=self.device),
                torch.empty((self.p, self.q), dtype=self._td, device=self.device),
                torch.empty(1, dtype=torch.float64, device=self.device))
        if self.n_users == 0 or self.n_samples == 0:
            return None, None, None
        Md = bufs[0]
        Md.copy_(torch.from_numpy(M))
        out = bufs[1].zero_() if want_out else None
        sse = bufs[2].zero_() if want_sse else None
        # predictions: this rank fills its samples' slots of the full (grouped) vector, the
        # other ranks' slots arrive by the gather below
        # (a shard without a group leaves them zero: its caller adds the shards up)
        local = getattr(self, 'local', False)  # predictions of this rank's own samples only
        alloc = torch.zeros if self.group is None and self.n_local_samples != self.n_samples \
            else torch.empty
        pred = alloc(self.n_local_samples if local else self.n_samples, dtype=self._td,
                     device=self.device) if want_pred else None
        r = self.ratings if ratings is None else ratings
        if self.n_local_users > 0 and self.n_local_samples > 0:
            local_pred = pred[self.slo:self.shi] if want_pred else None
            self._run(mode, Md, r, out, local_pred, sse)
        if self.group is not None:
            import torch.distributed as dist
            for t in (out, sse):
                if t is not None:
                    self._all_reduce(t)
            if pred is not None and not local:
                # all-gather-v of the ranks' slices: one broadcast per owner straight into its
                # slots (1/N of the vector each, where an all-reduce of a zero-filled vector
                # moved all of it through every rank)
                sb = self._sample_bounds
                bounce = pred.is_cuda and dist.get_backend(self.group) == 'gloo'
                for k in range(sb.size - 1):
                    if sb[k + 1] > sb[k]:
                        part = pred[int(sb[k]):int(sb[k + 1])]
                        src = dist.get_global_rank(self.group, k)
                        if bounce:   # (ranks rehearsing on one GPU under gloo: see _all_reduce)
                            host = part.cpu()
                            dist.broadcast(host, src=src, group=self.group)
                            part.copy_(host)
                        else:
                            dist.broadcast(part, src=src, group=self.group)
        return out, sse, pred

    def _eval(self, mode, M, want_out=True, want_sse=False, want_pred=False, ratings=None):
        """`_eval_dev` with the results brought to the host (float64; predictions in the
        caller's sample order)."""
        torch = self._torch
        out, sse, pred = self._eval_dev(mode, M, want_out, want_sse, want_pred, ratings)
        if self.n_users == 0 or self.n_samples == 0:
            zero = np.zeros((self.p, self.q))
            return (zero if want_out else None, 0.0 if want_sse else None,
                    np.zeros(0) if want_pred else None)
        out_h = out.to(torch.float64).cpu().numpy() if want_out else None
        sse_h = float(sse.cpu()[0]) if want_sse else None
        pred_h = None
        if want_pred:
            grouped = pred.to(torch.float64).cpu().numpy()
            if self.order is None:
                pred_h = grouped
            else:
                pred_h = np.empty_like(grouped)
                pred_h[self.order] = grouped
        return out_h, sse_h, pred_h

    # ---- rank-constrained form Z = U V^T (extension; see _fit_low_rank) --------------------
    def _low_rank_setup(self, r):
        """Item-major copy of the samples and the workspaces of the r-wide passes."""
        if getattr(self, '_lr', None) is not None and self._lr['r'] == r:
            return self._lr
        torch = self._torch
        lib = _lib.load()
        n_items_local = int(self.Y.shape[0])
        cptr, cidx, cvals = dev.csr_to_csc(self.indptr, self.items, self.ratings,
                                           (self.n_local_users, n_items_local))
        ld = lib.alsmi_imc_padded_width(int(r))
        code = _lib.dtype_code(self.dtype)
        ws = lambda rows, p: torch.empty(
            int(lib.alsmi_imc_workspace_bytes(code, max(rows, 1), self.n_local_samples, p, r))
            // 8 + 1,
            dtype=torch.float64, device=self.device)
        self._lr = dict(r=r, ld=ld, cptr=cptr, cidx=cidx, cvals=cvals,
                        A=torch.zeros((max(self.n_local_users, 1), ld), dtype=self._td,
                                      device=self.device),
                        B=torch.zeros((max(n_items_local, 1), ld), dtype=self._td,
                                      device=self.device),
                        Yq=self.Y[:, :self.q].contiguous(),
                        ws_u=ws(self.n_local_users, self.p), ws_i=ws(n_items_local, self.q))
        return self._lr

    def low_rank_eval(self, U, V):
        """sum_e (r_e - x_u^T U V^T y_i)^2 and the data parts of its gradients, G V and G^T U
        with G = X^T diag(pred - r) Y, from two passes over the samples that are r = U.shape[1]
        wide instead of q wide: user-major against the projected item table Y V, item-major
        against the projected user table X U."""
        torch = self._torch
        lib = _lib.load()
        U = np.ascontiguousarray(U, dtype=np.float64)
        V = np.ascontiguousarray(V, dtype=np.float64)
        r = U.shape[1]
        if self.n_samples == 0:
            return 0.0, np.zeros_like(U), np.zeros_like(V)
        L = self._low_rank_setup(r)
        code = _lib.dtype_code(self.dtype)
        Ud = torch.from_numpy(U).to(self.device, self._td)
        Vd = torch.from_numpy(V).to(self.device, self._td)
        gU = torch.zeros((self.p, r), dtype=self._td, device=self.device)
        gV = torch.zeros((self.q, r), dtype=self._td, device=self.device)
        sse = torch.zeros(1, dtype=torch.float64, device=self.device)
        if self.n_local_users > 0 and self.n_local_samples > 0:
            n_items_local = int(self.Y.shape[0])
            st = dev._stream()
            _lib.check(lib.alsmi_imc_project(code, self.n_local_users, self.p, r, dev._ptr(self.X),
                                             dev._ptr(Ud), dev._ptr(L['A']), st),
                       'alsmi_imc_project')
            _lib.check(lib.alsmi_imc_project(code, n_items_local, self.q, r, dev._ptr(L['Yq']),
                                             dev._ptr(Vd), dev._ptr(L['B']), st),
                       'alsmi_imc_project')
            _lib.check(lib.alsmi_imc_eval(
                code, 0, self.n_local_users, self.n_local_samples, self.p, r,
                dev._ptr(self.indptr),
                dev._ptr(self.items), dev._ptr(self.ratings), dev._ptr(self.X), dev._ptr(L['B']),
                dev._ptr(Ud), dev._ptr(gU), None, dev._ptr(sse), dev._ptr(L['ws_u']), st),
                'alsmi_imc_eval')
            _lib.check(lib.alsmi_imc_eval(
                code, 0, n_items_local, self.n_local_samples, self.q, r, dev._ptr(L['cptr']),
                dev._ptr(L['cidx']),
                dev._ptr(L['cvals']), dev._ptr(L['Yq']), dev._ptr(L['A']), dev._ptr(Vd),
                dev._ptr(gV), None, None, dev._ptr(L['ws_i']), st), 'alsmi_imc_eval')
        if self.group is not None:
            import torch.distributed as dist
            for t in (gU, gV, sse):
                self._all_reduce(t)
        return (float(sse.cpu()[0]), gU.to(torch.float64).cpu().numpy(),
                gV.to(torch.float64).cpu().numpy())

    # the three callbacks, with the reference's regulariser terms added on the host
    def cost(self, Z, lam):
        """src/imc.py:43-48, including its quirk: lam*||Z||^2 is inside the sum
        over samples, i.e. counted n_samples times."""
        Z = np.asarray(Z, dtype=np.float64)
        _, sse, _ = self._eval(0, Z, want_out=False, want_sse=True)
        return 0.5 * (sse + self.n_samples * lam * np.linalg.norm(Z) ** 2)

    def grad(self, Z, lam):
        """src/imc.py:65-71: X^T diag(preds - r) Y + lam Z."""
        Z = np.asarray(Z, dtype=np.float64).reshape(self.p, self.q)
        out, _, _ = self._eval(0, Z)
        return out + lam * Z

    def hessp(self, S, lam):
        """src/imc.py:91-96, including its quirk: the SCALAR lam is added to every
        entry (pinned by the reference's tests/test_imc.py:70-80)."""
        out, _, _ = self._eval(1, S)
        return out + lam

    def init_z(self):
        """src/imc.py:157-159: X^T diag(R) Y / R.size pushed through an SVD."""
        neg = -self.ratings  # (the local samples; the evaluation sums over the ranks)
        out, _, _ = self._eval(0, np.zeros((self.p, self.q)), ratings=neg)
        z0 = out / self.n_samples
        u, s, v = np.linalg.svd(z0, False)
        return u.dot(np.diag(s)).dot(v)

    def predict_samples(self, Z):
        """x_e^T Z y_e for every sample, in the caller's sample order."""
        _, _, pred = self._eval(1, Z, want_out=False, want_pred=True)
        return pred

    def predict_samples_grouped_dev(self, Z):
        """x_e^T Z y_e for every sample as a float64 device tensor in the GROUPED order (the
        samples user by user, each user's in the caller's order: the order of
        ``grouped_triplets_dev``)."""
        _, _, pred = self._eval_dev(1, Z, want_out=False, want_pred=True)
        if pred is None:
            return self._torch.zeros(0, dtype=self._torch.float64, device=self.device)
        return pred.to(self._torch.float64)

    def grouped_triplets_dev(self):
        """(users int32, items int32, ratings float64) of this rank's samples on the device, in
        the grouped order; only for an unsharded problem (item ids are then global)."""
        if self.n_local_samples != self.n_samples or int(self.Y.shape[0]) != self.n_items:
            raise RuntimeError('the samples of a sharded problem are not all on this rank')
        torch = self._torch
        users = torch.repeat_interleave(
            torch.arange(self.n_users, dtype=torch.int32, device=self.device),
            self.indptr[1:] - self.indptr[:-1], output_size=self.n_samples)
        return users, self.items, self.ratings.to(torch.float64)

    def sse(self, Z):
        _, sse, _ = self._eval(0, Z, want_out=False, want_sse=True)
        return sse


def problem_from_samples(users, items, ratings, dtype=np.float64):
    """Per-sample feature matrices (the reference's form) as an index-form problem
    in which every sample is its own 'user' and its own 'item'.  With an initialised
    torch.distributed group of several ranks the samples are sharded over the ranks."""
    users = np.atleast_2d(np.asarray(users))
    items = np.atleast_2d(np.asarray(items))
    n_s = users.shape[0]
    idx = np.arange(n_s)
    return ImcProblem(users, items, idx, idx, ratings, dtype=dtype, group=default_group())


_problem_cache = {}


def _cached_problem(users, items, ratings):
    """The scipy callbacks receive the same (users, items, ratings) objects on
    every call; keep their device copy instead of re-uploading per evaluation."""
    key = (id(users), id(items), id(ratings), np.shape(users), np.shape(items))
    hit = _problem_cache.get(key)
    if hit is not None and hit[0] is users and hit[1] is items and hit[2] is ratings:
        return hit[3]
    if isinstance(users, ImcProblem):
        prob = users
    else:
        prob = problem_from_samples(users, items, ratings)
    _problem_cache.clear()
    _problem_cache[key] = (users, items, ratings, prob)
    return prob


def _cost(arr, *args):
    """Regularised sum of squared errors (src/imc.py:30-48)."""
    users, items, ratings, lam, shape = args
    return _cached_problem(users, items, ratings).cost(np.asarray(arr).reshape(shape), lam)


def _cost_prime(arr, *args):
    """Gradient of ``_cost`` w.r.t. Z, flattened (src/imc.py:51-71)."""
    users, items, ratings, lam, shape = args
    prob = _cached_problem(users, items, ratings)
    return prob.grad(np.asarray(arr).reshape(shape), lam).ravel()


def _cost_hess(_, s_vec, *args):
    """Hessian-vector product, flattened (src/imc.py:74-96)."""
    users, items, ratings, lam, shape = args
    prob = _cached_problem(users, items, ratings)
    return prob.hessp(np.asarray(s_vec).reshape(shape), lam).ravel()


def _fit_problem(prob, Z, method, alpha, verbose):
    if Z is None:
        Z = prob.init_z()
    Z = np.asarray(Z, dtype=np.float64)
    shape = Z.shape
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', category=RuntimeWarning)
        res = so.minimize(
            lambda a: prob.cost(a.reshape(shape), alpha), Z.flatten(), method=method,
            jac=lambda a: prob.grad(a.reshape(shape), alpha).ravel(),
            hessp=lambda a, s: prob.hessp(s.reshape(shape), alpha).ravel(),
            options={'disp': verbose})
    return res['x'].reshape(shape), res.success, res.message


def _fit_imc(X, Y, R, Z=None, method='Newton-CG', n_components=None,
             alpha=0.1, verbose=0):
    """Minimise the IMC objective over Z for per-sample features X, Y and ratings
    R; returns ``(Z, success, message)`` (src/imc.py:99-170)."""
    _, n_features = np.atleast_2d(X).shape
    if n_components is None:
        n_components = n_features
    if not isinstance(n_components, INTEGER_TYPES) or n_components <= 0:
        raise ValueError("Number of components must be a positive integer;"
                         " got (n_components=%r)." % n_components)
    prob = problem_from_samples(X, Y, R)
    return _fit_problem(prob, Z, method, alpha, verbose)


def _fit_low_rank(prob, r, Z, method, alpha, verbose):
    """Extension with no counterpart in the reference (which validates ``n_components`` and
    then ignores it, src/imc.py:148-154): minimise over Z = U V^T of rank r,

        f(U, V) = 0.5 * (sum_e (r_e - x_u^T U V^T y_i)^2 + alpha (||U||^2 + ||V||^2)),
        df/dU = G V + alpha U,   df/dV = G^T U + alpha V,   G = X^T diag(pred - r) Y,

    evaluated by ``ImcProblem.low_rank_eval`` in passes that are r wide, not q wide.  Starts from the
    best rank-r approximation of the reference's initial Z (src/imc.py:157-159) or of the
    given Z.  Returns ``(U V^T, U, V, success, message)``."""
    Z0 = prob.init_z() if Z is None else np.asarray(Z, dtype=np.float64)
    u, s, vt = np.linalg.svd(Z0, False)
    root = np.sqrt(s[:r])
    U0, V0 = u[:, :r] * root, vt[:r].T * root
    p, q = prob.p, prob.q

    def unpack(a):
        return a[:p * r].reshape(p, r), a[p * r:].reshape(q, r)

    def fun(a):
        U, V = unpack(a)
        sse, g_u, g_v = prob.low_rank_eval(U, V)
        f = 0.5 * (sse + alpha * (np.sum(U * U) + np.sum(V * V)))
        g = np.concatenate(((g_u + alpha * U).ravel(), (g_v + alpha * V).ravel()))
        return f, g

    with warnings.catch_warnings():
        warnings.simplefilter('ignore', category=RuntimeWarning)
        res = so.minimize(fun, np.concatenate((U0.ravel(), V0.ravel())), jac=True,
                          method='L-BFGS-B' if method == 'Newton-CG' else method,
                          options={'disp': verbose} if verbose else {})
    U, V = unpack(res['x'])
    return U.dot(V.T), U, V, res.success, res.message


class IMC(BaseEstimator):
    """Inductive matrix completion estimator (src/imc.py:173-356).

    ``low_rank`` (extension, default off = the reference's behaviour, in which
    ``n_components`` is validated and recorded but Z stays a full p x q matrix): when true
    and ``n_components`` is smaller than both attribute counts, Z is constrained to
    U V^T with ``n_components`` columns (attributes ``U_``, ``V_``); see ``_fit_low_rank``."""

    def __init__(self, n_components=None, method='Newton-CG', alpha=0.1,
                 verbose=0, low_rank=False):
        self.n_components = n_components
        self.method = method
        self.alpha = alpha
        self.verbose = verbose
        self.low_rank = low_rank

    @staticmethod
    def _problem(X, y):
        x, y_ = _check_x(X)
        if issparse(y):
            y = y.tocsr()
            rows, cols = y.nonzero()
            if y.has_canonical_format:
                # sorted, no duplicates: the values at (rows, cols) are the stored non-zero
                # values in storage order; no lookup per entry
                vals = y.data[y.data != 0]
            else:
                vals = np.asarray(y[rows, cols]).ravel()
            return ImcProblem(x, y_, rows, cols, vals, group=default_group())
        return problem_from_samples(x, y_, y)

    def _fit_on(self, prob, Z=None):
        """Fit on an index-form problem (the shared tail of fit_transform; DirtyIMC calls it
        directly with attribute matrices and sample indices)."""
        n_features = prob.p
        if self.n_components and self.n_components < n_features:
            self.n_components_ = self.n_components
        else:
            self.n_components_ = n_features
        if not isinstance(self.n_components_, INTEGER_TYPES) or self.n_components_ <= 0:
            raise ValueError("Number of components must be a positive integer;"
                             " got (n_components=%r)." % self.n_components_)
        if self.low_rank and self.n_components_ < min(prob.p, prob.q):
            Z, self.U_, self.V_, success, msg = _fit_low_rank(
                prob, self.n_components_, Z, self.method, self.alpha, self.verbose)
        else:
            Z, success, msg = _fit_problem(prob, Z, self.method, self.alpha, self.verbose)
        if not success:
            warnings.warn(msg, ConvergenceWarning, stacklevel=1)
        self.Z = Z
        sse = prob.sse(Z)
        self.reconstruction_err_ = float(np.sqrt(sse / prob.n_samples))
        return Z

    def fit_sharded(self, X, users, items, ratings, Z=None):
        """``fit`` when every rank of the initialised ``torch.distributed`` group holds only
        its OWN samples: X = (X_attrs, Y_attrs) one row per user / item, `users`, `items`,
        `ratings` this rank's samples in index form (ImcProblem.from_local_samples).  Every
        rank ends with the same Z; ``reconstruction_err_`` is the RMSE over all ranks'
        samples.  Returns the problem (its ``predict_samples(Z)`` gives this rank's x^T Z y)."""
        x, y_ = _check_x(X)
        prob = ImcProblem.from_local_samples(x, y_, users, items, ratings, group=default_group())
        self._fit_on(prob, Z)
        return prob

    def fit_transform(self, X, y, Z=None):
        """Learn Z for the given data and return it (src/imc.py:220-254)."""
        return self._fit_on(self._problem(X, y), Z)

    def fit(self, X, y, Z=None):
        """src/imc.py:256-276."""
        _ = self.fit_transform(X, y, Z=Z)
        return self

    def _predict(self, X):
        """X Z Y^T for attribute matrices X (n x p), Y (m x q): the full n x m
        matrix; 1-D rows give a scalar (src/imc.py:278-295)."""
        x, y_ = _check_x(X)
        x_arr, y_arr = np.asarray(x), np.asarray(y_)
        x2, y2 = np.atleast_2d(x_arr), np.atleast_2d(y_arr)
        n, m = x2.shape[0], y2.shape[0]
        # every (row of X, row of Y) pair as one sample, user by user; the n*m index entries are
        # generated on the device instead of being built on the host, grouped and uploaded
        import torch
        device = dev.require_gpu()
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(device)
        indptr = torch.arange(n + 1, dtype=torch.int64, device=device) * m
        cols = torch.arange(m, dtype=torch.int32, device=device).repeat(n)
        zeros = torch.zeros(n * m, dtype=torch.float64, device=device)
        prob = ImcProblem.from_device_csr(indptr, cols, zeros, to(x2), to(y2))
        pred = prob.predict_samples(self.Z).reshape(n, m)
        if x_arr.ndim == 1 and y_arr.ndim == 1:
            return pred[0, 0]
        if x_arr.ndim == 1:
            return pred[0]
        if y_arr.ndim == 1:
            return pred[:, 0]
        return pred

    def predict_one(self, X):
        """src/imc.py:297-313."""
        check_is_fitted(self, 'n_components_')
        return self._predict(X)

    def predict_all(self, X):
        """src/imc.py:315-331."""
        check_is_fitted(self, 'n_components_')
        return self._predict(X)

    def score(self, X, y):
        """POSITIVE root mean squared error of the per-sample predictions
        (src/imc.py:333-356; note the sign differs from ALS and DirtyIMC)."""
        check_is_fitted(self, 'n_components_')
        prob = self._problem(X, y)
        # the predictions are all-reduced over the ranks of a group; the truth is the full
        # host vector every rank was given
        preds = prob.predict_samples(self.Z)
        return root_mean_squared_error(prob.ratings_host, preds)
