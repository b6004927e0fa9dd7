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
