"""Device state and sweep driver of the ALS update loop (src/als.py:154-169).

One ``AlsEngine`` lives in one process and drives one GPU.  With a
``torch.distributed`` process group of size G the users and items are cut into G
contiguous row ranges balanced by stored entries (the role of the reference's
``np.array_split`` chunks, src/als.py:155-159); every rank updates its own ranges
and each half-sweep ends with an all-gather of the updated factor rows (RCCL over
xGMI when the backend is "nccl"), the training error with an all-reduce of the
squared-error sum.
"""
import os

import numpy as np
import torch

from . import device as dev


def default_group():
    """The world group when torch.distributed runs with more than one rank, else None."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist.group.WORLD
    return None


def split_by_nnz(indptr, parts):
    """Boundaries of ``parts`` contiguous row ranges with (almost) equal numbers of
    stored entries: int64 array of parts+1 row ids."""
    indptr = np.asarray(indptr)
    n = indptr.size - 1
    total = int(indptr[-1])
    targets = (np.arange(1, parts, dtype=np.int64) * total) // parts
    cuts = np.searchsorted(indptr, targets, side='left')
    bounds = np.concatenate(([0], cuts, [n])).astype(np.int64)
    return np.maximum.accumulate(np.minimum(bounds, n))


class LocalBackend(object):
    """The HIP kernels on this process's GPU (the only product backend)."""

    def __init__(self, device=None):
        self.device = device if device is not None else dev.require_gpu()

    def make_side(self, indptr, indices, vals, n_rows, n_cols, dtype):
        return dev.SparseSide(indptr, indices, vals, n_rows, n_cols, dtype, self.device)

    def make_plan(self, side, rank, chunk, row_begin=0, row_end=None):
        return dev.SweepPlan(side, rank, row_begin=row_begin, row_end=row_end, chunk=chunk)

    def zeros(self, rows, cols, dtype):
        return torch.zeros((rows, cols), dtype=dev.torch_dtype(dtype), device=self.device)

    def sddmm(self, rank, users, items, row_table, col_table, vals, want_pred):
        return dev.sddmm(rank, users, items, row_table, col_table, vals, want_pred)

    def padded_rank(self, rank):
        return dev.padded_rank(rank)

    def csr_to_csc(self, indptr, indices, vals, shape, dtype):
        """CSR (host or device arrays) -> (csr, csc) triples of device tensors."""
        td = dev.torch_dtype(dtype)
        d_ptr = dev._as_device(indptr, torch.int64, self.device)
        d_idx = dev._as_device(indices, torch.int32, self.device)
        d_val = dev._as_device(vals, td, self.device)
        return (d_ptr, d_idx, d_val), dev.csr_to_csc(d_ptr, d_idx, d_val, shape)


class AlsEngine(object):
    """csr / csc are this rank's LOCAL slices: (indptr, indices, vals) of its user
    rows (columns = global item ids) and of its item rows (columns = global user
    ids), numpy arrays or device tensors.  user_bounds / item_bounds give every
    rank's row range (world+1 entries each); None means a single rank."""

    def __init__(self, csr, csc, shape, rank, alpha, dtype=np.float64, user_bounds=None,
                 item_bounds=None, group=None, backend=None, chunk=0, total_nnz=None,
                 user_slabs=None, item_slabs=None):
        self.backend = backend if backend is not None else LocalBackend()
        self.n, self.m = int(shape[0]), int(shape[1])
        self.rank, self.alpha = int(rank), float(alpha)
        self.dtype = np.dtype(dtype)
        self.group = group
        if group is not None:
            import torch.distributed as dist
            self.world = dist.get_world_size(group)
            self.me = dist.get_rank(group)
        else:
            self.world, self.me = 1, 0
        self.user_bounds = np.asarray([0, self.n] if user_bounds is None else user_bounds,
                                      dtype=np.int64)
        self.item_bounds = np.asarray([0, self.m] if item_bounds is None else item_bounds,
                                      dtype=np.int64)
        if self.user_bounds.size != self.world + 1 or self.item_bounds.size != self.world + 1:
            raise ValueError('row bounds must have world_size + 1 entries')
        self.ulo, self.uhi = (int(v) for v in self.user_bounds[self.me:self.me + 2])
        self.ilo, self.ihi = (int(v) for v in self.item_bounds[self.me:self.me + 2])
        b = self.backend
        self.csr = b.make_side(csr[0], csr[1], csr[2], self.uhi - self.ulo, self.m, self.dtype)
        self.csc = b.make_side(csc[0], csc[1], csc[2], self.ihi - self.ilo, self.n, self.dtype)
        # With several ranks each half runs as a few row slabs, balanced by entries, and the
        # all-gather of one slab travels while the next is computed (the user table is the
        # large one: N x 35 MB at 138 493 users per rank and rank 64, 5 GB at 10 M users and
        # rank 128).
        if user_slabs is None:
            user_slabs = int(os.environ.get('ALSMI_USER_SLABS', '0')) or \
                (1 if self.world == 1 else 4)
        if item_slabs is None:
            item_slabs = int(os.environ.get('ALSMI_ITEM_SLABS', '0')) or \
                (1 if self.world == 1 else 2)
        self.u_sub = split_by_nnz(self.csr.indptr_host, max(1, int(user_slabs)))
        self.i_sub = split_by_nnz(self.csc.indptr_host, max(1, int(item_slabs)))
        self.plans_u = [b.make_plan(self.csr, self.rank, chunk, int(lo), int(hi))
                        for lo, hi in zip(self.u_sub[:-1], self.u_sub[1:])]
        self.plans_i = [b.make_plan(self.csc, self.rank, chunk, int(lo), int(hi))
                        for lo, hi in zip(self.i_sub[:-1], self.i_sub[1:])]
        self.plan_i = self.plans_i[0]
        self.kp = b.padded_rank(self.rank)
        self.U = b.zeros(self.n, self.kp, self.dtype)
        self.V = b.zeros(self.m, self.kp, self.dtype)
        self._U_next, self._pending = None, False
        self._pair_rows = None
        self.local_nnz = self.csr.nnz
        if total_nnz is None:
            total_nnz = self.local_nnz
            if self.world > 1:
                import torch.distributed as dist
                t = torch.tensor([float(self.local_nnz)], dtype=torch.float64,
                                 device=self.U.device)
                dist.all_reduce(t, group=group)
                total_nnz = int(t.cpu()[0])
        self.nnz = int(total_nnz)
        # every rank's slab boundaries (global row ids), [world x slabs+1], and the staging
        # buffers of the slab gathers (allocated once: nothing is allocated inside a sweep)
        self.u_sub_all = self._exchange_bounds(self.u_sub + self.ulo)
        self.i_sub_all = self._exchange_bounds(self.i_sub + self.ilo)
        self._stage_u = self._make_stages(self.u_sub_all, self.U)
        self._stage_i = self._make_stages(self.i_sub_all, self.V)

    def _exchange_bounds(self, mine_np):
        mine = torch.from_numpy(np.ascontiguousarray(mine_np)).to(self.U.device)
        if self.world > 1:
            import torch.distributed as dist
            every = torch.empty(self.world * mine.numel(), dtype=mine.dtype, device=mine.device)
            dist.all_gather_into_tensor(every, mine, group=self.group)
            return every.cpu().numpy().reshape(self.world, -1)
        return mine.cpu().numpy().reshape(1, -1)

    def _make_stages(self, sub_all, table):
        """Per slab: (send [width x ld], recv [world*width x ld]) with width = the longest
        rank's slab; None on a single rank."""
        if self.world == 1:
            return None
        out = []
        for s in range(sub_all.shape[1] - 1):
            width = int((sub_all[:, s + 1] - sub_all[:, s]).max())
            out.append((torch.zeros((width, table.shape[1]), dtype=table.dtype,
                                    device=table.device),
                        torch.empty((self.world * width, table.shape[1]), dtype=table.dtype,
                                    device=table.device)))
        return out

    @classmethod
    def from_global(cls, indptr, indices, vals, shape, rank, alpha, dtype=np.float64,
                    group=None, backend=None, chunk=0, user_slabs=None, item_slabs=None):
        """Every rank is given the whole CSR (host arrays or device tensors).  The CSC is
        built on the device (the reference re-runs scipy's csr_tocsc per item,
        src/als.py:268-280), both are cut into per-rank row ranges balanced by stored
        entries and only the local slices are kept."""
        backend = backend if backend is not None else LocalBackend()
        n, m = int(shape[0]), int(shape[1])
        if group is not None:
            import torch.distributed as dist
            world, me = dist.get_world_size(group), dist.get_rank(group)
        else:
            world, me = 1, 0
        (ptr, idx, val), (cptr, cidx, cval) = backend.csr_to_csc(indptr, indices, vals, (n, m),
                                                                 dtype)
        ptr_h = ptr.cpu().numpy() if torch.is_tensor(ptr) else np.asarray(ptr, dtype=np.int64)
        cptr_h = cptr.cpu().numpy() if torch.is_tensor(cptr) else np.asarray(cptr, dtype=np.int64)
        ub, ib = split_by_nnz(ptr_h, world), split_by_nnz(cptr_h, world)
        a, z = int(ptr_h[ub[me]]), int(ptr_h[ub[me + 1]])
        csr = (ptr_h[ub[me]:ub[me + 1] + 1] - a, idx[a:z], val[a:z])
        a, z = int(cptr_h[ib[me]]), int(cptr_h[ib[me + 1]])
        csc = (cptr_h[ib[me]:ib[me + 1] + 1] - a, cidx[a:z], cval[a:z])
        return cls(csr, csc, (n, m), rank, alpha, dtype=dtype, user_bounds=ub, item_bounds=ib,
                   group=group, backend=backend, chunk=chunk, total_nnz=int(ptr_h[-1]),
                   user_slabs=user_slabs, item_slabs=item_slabs)

    # ---- factors ------------------------------------------------------------
    def set_factors(self, user_feats, item_feats):
        """Host (rank, n) / (rank, m) arrays -> device tables."""
        for table, feats in ((self.U, user_feats), (self.V, item_feats)):
            host = np.zeros((feats.shape[1], self.kp), dtype=self.dtype)
            host[:, :self.rank] = np.asarray(feats).T
            table.copy_(torch.from_numpy(host))

    def get_factors(self):
        return dev.factors_to_host(self.U, self.rank), dev.factors_to_host(self.V, self.rank)

    # ---- collectives --------------------------------------------------------
    def _all_gather_rows(self, table, bounds):
        """Every rank contributes rows [bounds[r], bounds[r+1]) of `table`; afterwards all
        ranks hold the whole table.  The send buffer never aliases the receive buffer."""
        if self.world == 1:
            return
        import torch.distributed as dist
        sizes = np.diff(bounds)
        lo, hi = int(bounds[self.me]), int(bounds[self.me + 1])
        if sizes.min() == sizes.max():
            send = table[lo:hi].clone()
            dist.all_gather_into_tensor(table, send, group=self.group)
            return
        width = int(sizes.max())
        send = torch.zeros((width, table.shape[1]), dtype=table.dtype, device=table.device)
        send[:hi - lo].copy_(table[lo:hi])
        recv = torch.empty((self.world * width, table.shape[1]), dtype=table.dtype,
                           device=table.device)
        dist.all_gather_into_tensor(recv, send, group=self.group)
        for r in range(self.world):
            if r != self.me and sizes[r]:
                table[int(bounds[r]):int(bounds[r + 1])].copy_(
                    recv[r * width:r * width + int(sizes[r])])

    def _start_slab_gather(self, table, sub_all, stages, s, src=None):
        """Begin the all-gather of slab s (rows sub_all[r, s:s+2] of every rank r) into
        `table` and return what `_finish_slab_gather` needs.  This rank's rows are taken
        from `src` (a table indexed like `table`; default `table` itself).  The collective
        is ordered after the kernels already enqueued on the current stream and runs beside
        the ones enqueued later."""
        import torch.distributed as dist
        lo = sub_all[:, s]
        sizes = sub_all[:, s + 1] - lo
        width = int(sizes.max())
        send, recv = stages[s]
        mlo, mine = int(lo[self.me]), int(sizes[self.me])
        send[:mine].copy_((table if src is None else src)[mlo:mlo + mine])
        # asynchronous where that is the backend's normal mode (RCCL; gloo on host tensors).
        # gloo with device tensors (several ranks rehearsing on one GPU) is unreliable:
        # that case goes through host copies, blocking.
        overlap = dist.get_backend(self.group) == 'nccl' or not table.is_cuda
        if overlap:
            work = dist.all_gather_into_tensor(recv, send, group=self.group, async_op=True)
        else:
            work = None
            host = torch.empty(recv.shape, dtype=recv.dtype)
            dist.all_gather_into_tensor(host, send.cpu(), group=self.group)
            recv.copy_(host)
        return work, recv, lo, sizes, width

    def _finish_slab_gather(self, table, pending, include_me=False):
        work, recv, lo, sizes, width = pending
        if work is not None:
            work.wait()
        for r in range(self.world):
            if (include_me or r != self.me) and sizes[r]:
                table[int(lo[r]):int(lo[r] + sizes[r])].copy_(
                    recv[r * width:r * width + int(sizes[r])])

    # ---- the loop body ------------------------------------------------------
    def half_sweep_users(self):
        self._user_half(None, self.U, False)

    def _user_half(self, cur, dst_table, want_err):
        mine = slice(self.ulo, self.uhi)
        return self._half_local(self.plans_u, self.V, cur, dst_table, mine, self.u_sub_all,
                                self._stage_u, want_err)

    def _half_local(self, plans, src, cur, dst_table, mine, sub_all, stages, want_err):
        """The plans index this rank's rows from 0, the gathers index whole tables."""
        sse = None
        pending = []
        dst_local = dst_table[mine]
        for s, plan in enumerate(plans):
            if want_err:
                part = plan.half_sweep_err(src, cur, dst_local, self.alpha)
                sse = part if sse is None else sse + part
            else:
                plan.half_sweep(src, dst_local, self.alpha)
            if self.world > 1:
                pending.append(self._start_slab_gather(dst_table, sub_all, stages, s))
        for p in pending:
            self._finish_slab_gather(dst_table, p)
        return sse

    def half_sweep_users_err(self, keep_old=False):
        """The user half-sweep, which also gives the squared training error of the
        factors as they are BEFORE it (the error the reference computes at the end of the
        previous sweep, src/als.py:161-165) at no extra pass over the ratings.  Returns the
        all-reduced 1-element fp64 device tensor.  With keep_old the new rows go to a second
        full-size table (their slab gathers included) until ``commit_users()`` swaps it in
        (or ``discard_users()`` drops it), so a caller that decides to stop on that error
        still holds the factors it belongs to."""
        cur = self.U[self.ulo:self.uhi]
        if keep_old:
            if self._U_next is None:
                self._U_next = torch.zeros_like(self.U)
            sse = self._user_half(cur, self._U_next, True)
            self._pending = True
        else:
            sse = self._user_half(cur, self.U, True)
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(sse, group=self.group)
        return sse

    def commit_users(self):
        if not self._pending:
            raise RuntimeError('no parked user half-sweep to commit')
        # the parked table is complete on every rank (its slabs were gathered as they were
        # computed): it becomes the user table, the old one the next parking space
        self.U, self._U_next = self._U_next, self.U
        self._pending = False

    def discard_users(self):
        self._pending = False

    def half_sweep_items(self):
        self._half_local(self.plans_i, self.U, None, self.V, slice(self.ilo, self.ihi),
                         self.i_sub_all, self._stage_i, False)

    def train_sse_async(self):
        """Sum of squared errors over all stored entries as a 1-element fp64 device
        tensor (src/als.py:161-165 before the mean and sqrt)."""
        if self._pair_rows is None:  # local row id of every stored entry, built on first use
            self._pair_rows = self.csr.expand_rows()
        _, sse = self.backend.sddmm(self.rank, self._pair_rows, self.csr.indices,
                                    self.U[self.ulo:self.uhi], self.V, self.csr.vals, False)
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(sse, group=self.group)
        return sse

    def sum_sq_ratings(self):
        """Sum of the squared stored ratings over all ranks (a float)."""
        t = (self.csr.vals.to(torch.float64) ** 2).sum().reshape(1)
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(t, group=self.group)
        return float(t.cpu()[0])

    def train_rmse(self):
        sse = float(self.train_sse_async().cpu()[0])
        return float(np.sqrt(sse / self.nnz)) if self.nnz else float('nan')

    def sweep(self):
        self.half_sweep_users()
        self.half_sweep_items()
