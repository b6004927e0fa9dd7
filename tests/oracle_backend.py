"""CPU stand-in for the device backend of alsmi.engine.AlsEngine, built on the
oracle.  TEST INFRASTRUCTURE: it lets the sharding / all-gather / all-reduce logic
of the engine run under the gloo backend on machines without a GPU.  The product
never constructs it."""
import numpy as np
import torch

from oracle import als_oracle as ao
from oracle import c_oracle as co


class OracleSide(object):
    def __init__(self, indptr, indices, vals, n_rows, n_cols, dtype):
        as_np = lambda a: a.cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
        self.indptr_host = np.ascontiguousarray(as_np(indptr), dtype=np.int64)
        self.indices_np = np.ascontiguousarray(as_np(indices), dtype=np.int32)
        self.vals_np = np.ascontiguousarray(as_np(vals), dtype=np.float64)
        self.n_rows, self.n_cols = int(n_rows), int(n_cols)
        self.nnz = int(self.indptr_host[-1])
        self.indices = torch.from_numpy(self.indices_np)
        self.vals = torch.from_numpy(self.vals_np)
        self.dtype = np.dtype(np.float64)

    def expand_rows(self):
        rows = np.repeat(np.arange(self.n_rows, dtype=np.int32), np.diff(self.indptr_host))
        return torch.from_numpy(rows)


class OraclePlan(object):
    def __init__(self, side, rank, row_begin=0, row_end=None):
        self.side, self.rank = side, rank
        self.row_begin = int(row_begin)
        self.row_end = side.n_rows if row_end is None else int(row_end)

    def half_sweep(self, src, dst, alpha):
        s = self.side
        if self.row_end <= self.row_begin:
            return
        out = co.half_sweep(s.indptr_host, s.indices_np, s.vals_np,
                            src[:, :self.rank].numpy(), alpha, row_begin=self.row_begin,
                            row_end=self.row_end, threads=1)
        dst[self.row_begin:self.row_end, :self.rank] = torch.from_numpy(out)


    def half_sweep_err(self, src, cur, dst, alpha):
        s = self.side
        sse = torch.zeros(1, dtype=torch.float64)
        if self.row_end <= self.row_begin:
            return sse
        a, z = int(s.indptr_host[self.row_begin]), int(s.indptr_host[self.row_end])
        rows = np.repeat(np.arange(self.row_begin, self.row_end),
                         np.diff(s.indptr_host[self.row_begin:self.row_end + 1]))
        pred = (cur[:s.n_rows, :self.rank].numpy()[rows] *
                src[:, :self.rank].numpy()[s.indices_np[a:z]]).sum(-1)
        sse[0] = float(((s.vals_np[a:z] - pred) ** 2).sum())
        self.half_sweep(src, dst, alpha)
        return sse


class OracleBackend(object):
    def make_side(self, indptr, indices, vals, n_rows, n_cols, dtype):
        return OracleSide(indptr, indices, vals, n_rows, n_cols, dtype)

    def make_plan(self, side, rank, chunk, row_begin=0, row_end=None):
        return OraclePlan(side, rank, row_begin, row_end)

    def zeros(self, rows, cols, dtype):
        return torch.zeros((rows, cols), dtype=torch.float64)

    def padded_rank(self, rank):
        return 16 if rank <= 16 else 32 if rank <= 32 else 64 if rank <= 64 else 128

    def csr_to_csc(self, indptr, indices, vals, shape, dtype):
        indptr = np.asarray(indptr, dtype=np.int64)
        indices = np.asarray(indices, dtype=np.int32)
        vals = np.asarray(vals, dtype=np.float64)
        cptr, cidx, cvals = ao.csr_to_csc(int(shape[0]), int(shape[1]), indptr, indices, vals)
        return (indptr, indices, vals), (cptr, cidx, cvals)

    def sddmm(self, rank, users, items, row_table, col_table, vals, want_pred):
        u = row_table[users.long(), :rank]
        v = col_table[items.long(), :rank]
        pred = (u * v).sum(-1)
        sse = ((vals - pred) ** 2).sum().reshape(1) if vals is not None else None
        return (pred if want_pred else None), sse


def make_oracle_imc_problem():
    """CPU stand-in for alsmi.imc.ImcProblem (same class, device side replaced by the numpy /
    C oracle), so that the sample sharding and the all-reduces of the IMC evaluations can run
    under gloo without a GPU."""
    from alsmi.imc import ImcProblem

    class OracleImcProblem(ImcProblem):
        def _init_device(self):
            self._torch = torch
            self.device = torch.device('cpu')

        def _padded_width(self, q):
            if not 1 <= q <= 256:
                raise ValueError('item attribute count must be between 1 and 256, got %d' % q)
            ld = 16
            while ld < q:
                ld *= 2
            return ld

        def _place(self, indptr, items_g, ratings_g, X_attrs, ypad):
            self.indptr = torch.from_numpy(np.ascontiguousarray(indptr, dtype=np.int64))
            self.items = torch.from_numpy(np.ascontiguousarray(items_g, dtype=np.int32))
            self.ratings = torch.from_numpy(np.ascontiguousarray(ratings_g, dtype=np.float64))
            self.X = torch.from_numpy(np.ascontiguousarray(X_attrs, dtype=np.float64))
            self.Y = torch.from_numpy(np.ascontiguousarray(ypad, dtype=np.float64))
            self._td = torch.float64

        def _run(self, mode, Md, r, out, local_pred, sse):
            Y = self.Y[:, :self.q].numpy()
            M = Md.numpy().astype(np.float64)
            got, s = co.imc_eval(int(mode), self.indptr.numpy(), self.items.numpy(), r.numpy(),
                                 self.X.numpy(), Y, M, threads=1)
            if out is not None:
                out.copy_(torch.from_numpy(got))
            if sse is not None:
                sse[0] = s
            if local_pred is not None:
                A = self.X.numpy().dot(M)
                rows = np.repeat(np.arange(self.n_local_users), np.diff(self.indptr.numpy()))
                local_pred.copy_(torch.from_numpy((A[rows] * Y[self.items.numpy()]).sum(-1)))

    return OracleImcProblem
