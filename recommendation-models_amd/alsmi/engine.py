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
import warnings

import numpy as np
import torch

from . import device as dev


def gather_mode():
    """How the updated factor rows travel between ranks: 'p2p' (one batch of pairwise
    transfers per gather, the default) or 'bcast' (one broadcast per owning rank)."""
    mode = os.environ.get('ALSMI_GATHER', 'p2p')
    if mode not in ('p2p', 'bcast'):
        raise ValueError("ALSMI_GATHER must be 'p2p' or 'bcast'")
    return mode


def force_collectives():
    """ALSMI_FORCE_COLLECTIVES=1: a process group of ONE rank still goes through every
    collective of the multi-rank path (the broadcasts of the row gathers, the all-reduces),
    which lets a one-GPU machine run that path on RCCL."""
    return os.environ.get('ALSMI_FORCE_COLLECTIVES', '0') not in ('', '0')


def default_group():
    """The world group when torch.distributed runs with more than one rank (or with one rank
    and the collectives forced on), else None."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and \
            (dist.get_world_size() > 1 or force_collectives()):
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
        # whether the half-sweeps end in collectives: always with several ranks; with a group
        # of one rank only on request (the calls are then in-place broadcasts to itself)
        self.collective = group is not None and (self.world > 1 or force_collectives())
        self.time_gathers, self._gather_stamps = False, []
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
        # gather of one slab travels while the next is computed.  A slab is a launch of its
        # own (138 493 users, 20 M entries, rank 64 on one GPU: 1.08 / 1.18 / 1.51 ms for 1 / 2 /
        # 4 user slabs; 1.25 M users, 125 M entries, rank 128: 17.0 / 17.7 / 17.6 ms), so the
        # count follows the bytes this rank contributes to the gather: 4 slabs from 256 MB
        # (640 MB per rank at 10 M users over 8 GPUs and rank 128), 2 from 16 MB (35 MB at
        # 138 493 users per rank and rank 64), else 1.
        self.kp = b.padded_rank(self.rank)

        def slabs_for(rows, env):
            forced = int(os.environ.get(env, '0'))
            if forced:
                return forced
            if not self.collective:
                return 1
            nbytes = rows * self.kp * self.dtype.itemsize
            return 4 if nbytes >= (256 << 20) else 2 if nbytes >= (16 << 20) else 1

        if user_slabs is None:
            # (the largest rank's rows: every rank must arrive at the same count)
            user_slabs = slabs_for(int(np.diff(self.user_bounds).max()), 'ALSMI_USER_SLABS')
        if item_slabs is None:
            item_slabs = slabs_for(int(np.diff(self.item_bounds).max()), 'ALSMI_ITEM_SLABS')
        self.u_sub = split_by_nnz(self.csr.indptr_host, max(1, int(user_slabs)))
        self.i_sub = split_by_nnz(self.csc.indptr_host, max(1, int(item_slabs)))
        self.plans_u = [b.make_plan(self.csr, self.rank, chunk, int(lo), int(hi))
                        for lo, hi in zip(self.u_sub[:-1], self.u_sub[1:])]
        self.plans_i = [b.make_plan(self.csc, self.rank, chunk, int(lo), int(hi))
                        for lo, hi in zip(self.i_sub[:-1], self.i_sub[1:])]
        self.plan_i = self.plans_i[0]
        self.U = b.zeros(self.n, self.kp, self.dtype)
        self.V = b.zeros(self.m, self.kp, self.dtype)
        self._U_next, self._pending = None, False
        self._pair_rows = None
        self.local_nnz = self.csr.nnz
        if total_nnz is None:
            total_nnz = self.local_nnz
            if self.collective:
                import torch.distributed as dist
                t = torch.tensor([float(self.local_nnz)], dtype=torch.float64,
                                 device=self.U.device)
                dist.all_reduce(t, group=group)
                total_nnz = int(t.cpu()[0])
        self.nnz = int(total_nnz)
        self._host_bounce = False
        self._p2p_refused = False
        if self.collective:
            import torch.distributed as dist
            self._host_bounce = self.U.is_cuda and dist.get_backend(group) == 'gloo'
        # every rank's slab boundaries (global row ids), [world x slabs+1]
        self.u_sub_all = self._exchange_bounds(self.u_sub + self.ulo)
        self.i_sub_all = self._exchange_bounds(self.i_sub + self.ilo)

    def _exchange_bounds(self, mine_np):
        mine = torch.from_numpy(np.ascontiguousarray(mine_np)).to(self.U.device)
        if self.collective:
            import torch.distributed as dist
            every = torch.empty(self.world * mine.numel(), dtype=mine.dtype, device=mine.device)
            dist.all_gather_into_tensor(every, mine, group=self.group)
            return every.cpu().numpy().reshape(self.world, -1)
        return mine.cpu().numpy().reshape(1, -1)

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

    @classmethod
    def from_shards(cls, users, items, vals, shape, rank, alpha, dtype=np.float64, group=None,
                    backend=None, chunk=0, user_slabs=None, item_slabs=None):
        """Every rank is given only a PART of the ratings: (users, items, vals) torch tensors
        (int ids, any float type) on the device its collectives run on, the distinct stored
        (user, item) pairs of the whole matrix split over the ranks in any way.  No rank ever
        holds the whole matrix: the per-user and per-item entry counts are all-reduced, the row
        ranges cut from them (balanced by entries, as from_global cuts them), every entry
        travels once to the owner of its user and once to the owner of its item
        (all-to-all), and each rank orders what it received into its CSR and CSC slices —
        the sharded counterpart of the reference building one scipy matrix (src/als.py:137-142)
        and converting it per item (src/als.py:268-280).  A pair that occurs twice (on one rank
        or on two) is an error here: "the last write wins" has no meaning across ranks."""
        backend = backend if backend is not None else LocalBackend()
        n, m = int(shape[0]), int(shape[1])
        users, items = users.to(torch.int64), items.to(torch.int64)
        if not (users.numel() == items.numel() == vals.numel()):
            raise ValueError('users, items and vals must have the same length')
        if users.numel() and (int(users.min()) < 0 or int(users.max()) >= n or
                              int(items.min()) < 0 or int(items.max()) >= m):
            raise IndexError('index out of bounds for shape {}'.format((n, m)))
        collective = bounce = False
        world, me = 1, 0
        if group is not None:
            import torch.distributed as dist
            world, me = dist.get_world_size(group), dist.get_rank(group)
            collective = world > 1 or force_collectives()
            # several ranks rehearsing on one GPU under gloo: the exchanges go through the host
            bounce = users.is_cuda and dist.get_backend(group) == 'gloo'
        work = users.device if not (collective and bounce) else torch.device('cpu')

        def all_reduce(t):
            if collective:
                import torch.distributed as dist
                t = t.to(work)
                dist.all_reduce(t, group=group)
            return t

        def exchange(dest, cols):
            """Rows with destination rank dest[k] (sorted by it) -> the rows this rank receives
            from every rank, in sender order."""
            if not collective:
                return cols
            import torch.distributed as dist
            order = torch.argsort(dest, stable=True)
            send = torch.bincount(dest, minlength=world).to(work)
            recv = torch.empty_like(send)
            dist.all_to_all_single(recv, send, group=group)
            send_l, recv_l = send.cpu().tolist(), recv.cpu().tolist()
            out = []
            for c in cols:
                src = c[order].to(work).contiguous()
                dst = torch.empty(int(sum(recv_l)), dtype=c.dtype, device=work)
                dist.all_to_all_single(dst, src, output_split_sizes=recv_l,
                                       input_split_sizes=send_l, group=group)
                out.append(dst.to(users.device))
            return out

        def cut(counts):
            ptr = np.zeros(counts.numel() + 1, dtype=np.int64)
            np.cumsum(counts.cpu().numpy(), out=ptr[1:])
            return split_by_nnz(ptr, world)

        ub = cut(all_reduce(torch.bincount(users, minlength=n)))
        ib = cut(all_reduce(torch.bincount(items, minlength=m)))

        def side(rows, cols, bounds, n_cols):
            """This rank's slice of the row-major structure: the entries whose row lies in its
            range, ordered by (row, column)."""
            inner = torch.from_numpy(bounds[1:-1]).to(rows.device)
            r, c, v = exchange(torch.bucketize(rows, inner, right=True), [rows, cols, vals])
            lo, hi = int(bounds[me]), int(bounds[me + 1])
            key, order = torch.sort((r - lo) * n_cols + c)
            if key.numel() > 1 and bool((key[1:] == key[:-1]).any()):
                raise ValueError('from_shards: a (user, item) pair is stored more than once')
            ptr = torch.zeros(hi - lo + 1, dtype=torch.int64, device=rows.device)
            torch.cumsum(torch.bincount(r - lo, minlength=hi - lo), 0, out=ptr[1:])
            return ptr, (key % n_cols).to(torch.int32), v[order]

        csr = side(users, items, ub, m)
        csc = side(items, users, ib, n)
        total = all_reduce(torch.tensor([users.numel()], dtype=torch.int64, device=users.device))
        return cls(csr, csc, (n, m), rank, alpha, dtype=dtype, user_bounds=ub, item_bounds=ib,
                   group=group, backend=backend, chunk=chunk, total_nnz=int(total.cpu()[0]),
                   user_slabs=user_slabs, item_slabs=item_slabs)

    def item_means(self):
        """Mean stored rating of every item (0 for an unrated one) as a float64 [m] tensor on
        every rank (src/als.py:144-148): segment sums over this rank's item-major slice, one
        per item and in stored order, then the row gather of the half-sweeps."""
        ptr = torch.from_numpy(self.csc.indptr_host).to(self.V.device)
        table = torch.zeros((self.m, 1), dtype=torch.float64, device=self.V.device)
        if self.csc.nnz:
            sums = torch.segment_reduce(self.csc.vals.to(torch.float64), 'sum', offsets=ptr,
                                        initial=0.0)
            cnt = (ptr[1:] - ptr[:-1]).to(torch.float64)
            table[self.ilo:self.ihi, 0] = torch.where(cnt > 0, sums / cnt.clamp_min(1),
                                                       torch.zeros_like(sums))
        self._all_gather_rows(table, self.item_bounds)
        return table[:, 0]

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
    # The all-gather of updated factor rows is an all-gather-v straight from and into the rows
    # of the factor table itself: the row ranges differ from rank to rank (they are balanced by
    # stored entries), so nothing is padded, staged or copied afterwards.  Every rank sends its
    # rows to every other rank and receives theirs as ONE batch of point-to-point transfers
    # (`batch_isend_irecv`: a single grouped launch on RCCL): xGMI links GPU pairs directly, so
    # the 2 (N-1) transfers of a rank run side by side, each on the link of its pair, where a
    # broadcast per owner would run one after the other.  ALSMI_GATHER=bcast selects those
    # broadcasts instead (they are also what several ranks sharing one GPU under gloo use).
    def _bcast_rows(self, table, los, his, async_op):
        """Rank r sends rows [los[r], his[r]) of its `table` to the same rows of every other
        rank's table.  Returns the pending work handles (empty when blocking)."""
        import torch.distributed as dist
        if not self._host_bounce and gather_mode() == 'p2p' and not self._p2p_refused:
            mine = table[int(los[self.me]):int(his[self.me])]
            # (a group of one rank whose collectives are forced on exchanges with itself)
            peers = [r for r in range(self.world) if r != self.me] or [self.me]
            ops = []
            for r in peers:
                peer = dist.get_global_rank(self.group, r)
                rows = table[int(los[r]):int(his[r])]
                if mine.shape[0] > 0:
                    ops.append(dist.P2POp(dist.isend, mine, peer, self.group))
                if rows.shape[0] > 0:
                    ops.append(dist.P2POp(dist.irecv, rows, peer, self.group))
            try:
                works = dist.batch_isend_irecv(ops) if ops else []
            except (RuntimeError, ValueError, NotImplementedError) as err:
                # a backend that refuses batched point-to-point transfers outright (every rank
                # gets the same refusal before anything is sent): the broadcasts below serve
                warnings.warn('alsmi: batched point-to-point gather refused (%s); using one '
                              'broadcast per owning rank' % err)
                self._p2p_refused = True
                return self._bcast_rows(table, los, his, async_op)
            if async_op:
                return works
            for w in works:
                w.wait()
            return []
        works = []
        for r in range(self.world):
            lo, hi = int(los[r]), int(his[r])
            if hi <= lo:
                continue
            rows = table[lo:hi]
            src = dist.get_global_rank(self.group, r)
            if self._host_bounce:
                # several ranks rehearsing on one GPU under gloo: gloo is unreliable on device
                # tensors, so the rows travel through a host copy, blocking
                host = rows.cpu()
                dist.broadcast(host, src=src, group=self.group)
                if r != self.me:
                    rows.copy_(host)
            elif async_op:
                works.append(dist.broadcast(rows, src=src, group=self.group, async_op=True))
            else:
                dist.broadcast(rows, src=src, group=self.group)
        return works

    def _all_gather_rows(self, table, bounds):
        """Every rank contributes rows [bounds[r], bounds[r+1]) of `table`; afterwards all
        ranks hold the whole table."""
        if self.collective:
            self._bcast_rows(table, bounds[:-1], bounds[1:], False)

    def _start_slab_gather(self, table, sub_all, s):
        """Begin the gather of slab s (rows sub_all[r, s:s+2] of every rank r) into `table`.
        The broadcasts are ordered after the kernels already enqueued on the current stream
        and run beside the ones enqueued later; `_finish_gathers` makes the stream wait."""
        return self._bcast_rows(table, sub_all[:, s], sub_all[:, s + 1], True)

    def _finish_gathers(self, pending):
        """Wait for the given gathers.  With `time_gathers` set the time the compute stream
        spends in these waits (the part of the communication that no kernel hides) is added
        to `gather_exposed_ms`, measured with events on that stream."""
        works = [w for p in pending for w in p]
        timed = self.time_gathers and works
        if timed:
            t0 = self._stamp()
        for w in works:
            w.wait()
        if timed:
            self._gather_stamps.append((t0, self._stamp()))

    def _stamp(self):
        if self.U.is_cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            return ev
        import time
        return time.perf_counter()

    def gather_exposed_ms(self, reset=True):
        """Sum over the half-sweeps since the last reset of the time the compute stream waited
        for gathers (synchronises the device)."""
        total = 0.0
        for a, b in self._gather_stamps:
            if self.U.is_cuda:
                b.synchronize()
                total += a.elapsed_time(b)
            else:
                total += (b - a) * 1e3
        if reset:
            self._gather_stamps = []
        return total

    # ---- the loop body ------------------------------------------------------
    def half_sweep_users(self):
        self._user_half(None, self.U, False)

    def _user_half(self, cur, dst_table, want_err):
        mine = slice(self.ulo, self.uhi)
        return self._half_local(self.plans_u, self.V, cur, dst_table, mine, self.u_sub_all,
                                want_err)

    def _half_local(self, plans, src, cur, dst_table, mine, sub_all, want_err):
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
            if self.collective:
                pending.append(self._start_slab_gather(dst_table, sub_all, s))
        self._finish_gathers(pending)
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
        if self.collective:
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
                         self.i_sub_all, False)

    def train_sse_async(self):
        """Sum of squared errors over all stored entries as a 1-element fp64 device
        tensor (src/als.py:161-165 before the mean and sqrt)."""
        if self._pair_rows is None:  # local row id of every stored entry, built on first use
            self._pair_rows = self.csr.expand_rows()
        _, sse = self.backend.sddmm(self.rank, self._pair_rows, self.csr.indices,
                                    self.U[self.ulo:self.uhi], self.V, self.csr.vals, False)
        if self.collective:
            import torch.distributed as dist
            dist.all_reduce(sse, group=self.group)
        return sse

    def sum_sq_ratings(self):
        """Sum of the squared stored ratings over all ranks (a float)."""
        t = (self.csr.vals.to(torch.float64) ** 2).sum().reshape(1)
        if self.collective:
            import torch.distributed as dist
            dist.all_reduce(t, group=self.group)
        return float(t.cpu()[0])

    def train_rmse(self):
        sse = float(self.train_sse_async().cpu()[0])
        return float(np.sqrt(sse / self.nnz)) if self.nnz else float('nan')

    def sweep(self):
        self.half_sweep_users()
        self.half_sweep_items()
