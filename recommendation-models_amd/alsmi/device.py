"""Device-side containers and thin wrappers over the libalsmi entry points.

torch is used for device memory, streams and (in ``dist.py``) collectives only;
every arithmetic step is a libalsmi kernel.
"""
import ctypes

import numpy as np
import torch

from . import _lib


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError('alsmi needs a HIP device (torch.cuda.is_available() is False); '
                           'there is no CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def torch_dtype(np_dtype):
    return torch.float32 if np.dtype(np_dtype) == np.float32 else torch.float64


def padded_rank(rank):
    kp = _lib.load().alsmi_padded_rank(int(rank))
    if kp == 0:
        raise ValueError('rank must be between 1 and 128, got %r' % (rank,))
    return kp


def factors_to_device(feats, dtype, device):
    """(rank, n) host array, the reference's layout (src/als.py:149-151) ->
    row-major [n x ld] device table with zero pad columns."""
    feats = np.asarray(feats)
    rank, n = feats.shape
    kp = padded_rank(rank)
    host = np.zeros((n, kp), dtype=dtype)
    host[:, :rank] = feats.T
    return torch.from_numpy(host).to(device)


def factors_to_host(table, rank):
    """[n x ld] device table -> (rank, n) float64 C-order host array."""
    return table[:, :rank].t().contiguous().to(torch.float64).cpu().numpy()  # transposed on the device


def _as_device(a, tdtype, device):
    if torch.is_tensor(a):
        return a.to(device=device, dtype=tdtype).contiguous()
    return torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=tdtype)


class SparseSide(object):
    """One orientation of the ratings on the device: CSR for the user side,
    CSC (as the CSR of the transpose) for the item side."""

    def __init__(self, indptr, indices, vals, n_rows, n_cols, dtype, device):
        """indptr/indices/vals may be numpy arrays (uploaded) or torch tensors."""
        if torch.is_tensor(indptr):
            indptr = indptr.cpu().numpy()
        indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        if indptr.shape != (n_rows + 1,):
            raise ValueError('indptr must have n_rows + 1 entries')
        nnz = int(indptr[-1])
        if len(indices) != nnz or len(vals) != nnz:
            raise ValueError('indices/vals do not match indptr')
        self.n_rows, self.n_cols, self.nnz = int(n_rows), int(n_cols), nnz
        self.dtype = np.dtype(dtype)
        self.indptr_host = indptr
        self.indptr = torch.from_numpy(indptr).to(device)
        self.indices = _as_device(indices, torch.int32, device)
        self.vals = _as_device(vals, torch_dtype(self.dtype), device)
        if nnz and (int(self.indices.min()) < 0 or int(self.indices.max()) >= n_cols):
            raise ValueError('column index out of range')
        self.device = device

    def expand_rows(self, row_begin=0, row_end=None):
        """Row id of every stored entry of the row range (int32, device)."""
        if row_end is None:
            row_end = self.n_rows
        lo, hi = int(self.indptr_host[row_begin]), int(self.indptr_host[row_end])
        out = torch.empty(hi - lo, dtype=torch.int32, device=self.device)
        _lib.check(_lib.load().alsmi_expand_rows(_ptr(self.indptr), row_begin, row_end, _ptr(out),
                                                 _stream()), 'alsmi_expand_rows')
        return out


class SweepPlan(object):
    """Schedule of one side's half-sweep over rows [row_begin, row_end)."""

    def __init__(self, side, rank, row_begin=0, row_end=None, chunk=0):
        lib = _lib.load()
        if row_end is None:
            row_end = side.n_rows
        self.side, self.rank = side, int(rank)
        self.row_begin, self.row_end = int(row_begin), int(row_end)
        self.kp = padded_rank(rank)
        handle = ctypes.c_void_p(0)
        with torch.cuda.device(side.device):
            _lib.check(lib.alsmi_plan_create(
                ctypes.byref(handle), side.indptr_host.ctypes.data_as(ctypes.c_void_p),
                side.n_rows, self.row_begin, self.row_end, self.rank,
                _lib.dtype_code(side.dtype), int(chunk)), 'alsmi_plan_create')
        self._handle = handle

    def info(self):
        vals = [ctypes.c_int64(0) for _ in range(4)]
        _lib.check(_lib.load().alsmi_plan_info(self._handle, *[ctypes.byref(v) for v in vals]),
                   'alsmi_plan_info')
        return dict(zip(('n_rows', 'n_heavy_rows', 'n_chunks', 'workspace_bytes'),
                        (v.value for v in vals)))

    def _check_tables(self, src, dst):
        td = torch_dtype(self.side.dtype)
        if src.dtype != td or src.dim() != 2 or src.shape[1] != self.kp or \
                src.shape[0] < self.side.n_cols or not src.is_contiguous():
            raise ValueError('source factor table must be contiguous [>=%d x %d] %s'
                             % (self.side.n_cols, self.kp, td))
        if dst is not None and (dst.dtype != td or dst.dim() != 2 or dst.shape[1] != self.kp or
                                dst.shape[0] < self.side.n_rows or not dst.is_contiguous()):
            raise ValueError('destination factor table must be contiguous [>=%d x %d] %s'
                             % (self.side.n_rows, self.kp, td))

    def half_sweep(self, src, dst, alpha):
        """dst[r] = (sum f f^T + alpha n_r I)^-1 sum val f for the plan's rows."""
        self._check_tables(src, dst)
        s = self.side
        if s.nnz == 0:
            dst[self.row_begin:self.row_end].zero_()
            return
        _lib.check(_lib.load().alsmi_half_sweep(
            self._handle, _ptr(s.indptr), _ptr(s.indices), _ptr(s.vals), _ptr(src),
            max(int(src.shape[0]), 1), _ptr(dst), float(alpha), _stream()), 'alsmi_half_sweep')

    def half_sweep_err(self, src, cur, dst, alpha):
        """half_sweep that also returns the squared training error of the vectors it
        replaces: sum over the plan's stored entries of (val - <cur[r], src[col]>)^2, a
        1-element fp64 device tensor (src/als.py:161-165, fused).  cur may be dst."""
        self._check_tables(src, dst)
        self._check_tables(src, cur)
        s = self.side
        sse = torch.zeros(1, dtype=torch.float64, device=s.device)
        if s.nnz == 0:
            dst[self.row_begin:self.row_end].zero_()
            return sse
        _lib.check(_lib.load().alsmi_half_sweep_err(
            self._handle, _ptr(s.indptr), _ptr(s.indices), _ptr(s.vals), _ptr(src),
            max(int(src.shape[0]), 1), _ptr(cur), _ptr(dst), float(alpha), _ptr(sse), _stream()),
            'alsmi_half_sweep_err')
        return sse

    def normal_equations(self, src, alpha):
        """Test aid: (A [rows x ld x ld], b [rows x ld]) before the solve."""
        self._check_tables(src, None)
        s = self.side
        n = self.row_end - self.row_begin
        td = torch_dtype(s.dtype)
        A = torch.zeros((n, self.kp, self.kp), dtype=td, device=s.device)
        b = torch.zeros((n, self.kp), dtype=td, device=s.device)
        _lib.check(_lib.load().alsmi_normal_equations(
            self._handle, _ptr(s.indptr), _ptr(s.indices), _ptr(s.vals), _ptr(src),
            int(src.shape[0]), _ptr(A), _ptr(b), float(alpha), _stream()),
            'alsmi_normal_equations')
        return A, b

    def close(self):
        if getattr(self, '_handle', None) is not None and self._handle.value:
            _lib.load().alsmi_plan_destroy(self._handle)
            self._handle = ctypes.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_workspaces = {}


def _sddmm_workspace(device):
    key = (device.type, device.index)
    ws = _workspaces.get(key)
    if ws is None:
        nbytes = int(_lib.load().alsmi_sddmm_workspace_bytes())
        ws = torch.empty(nbytes // 8 + 1, dtype=torch.float64, device=device)
        _workspaces[key] = ws
    return ws


def sddmm(rank, users, items, row_table, col_table, vals=None, want_pred=True):
    """pred_p = <row_table[users[p]], col_table[items[p]]>; with ``vals`` also the
    sum of squared errors (a 1-element fp64 device tensor).  All device tensors."""
    lib = _lib.load()
    device = row_table.device
    n_pairs = int(users.numel())
    if users.dtype != torch.int32 or items.dtype != torch.int32 or items.numel() != n_pairs:
        raise ValueError('users/items must be int32 device tensors of equal length')
    if row_table.dtype != col_table.dtype:
        raise ValueError('factor tables must share a dtype')
    np_dtype = np.float32 if row_table.dtype == torch.float32 else np.float64
    pred = torch.empty(n_pairs, dtype=row_table.dtype, device=device) if want_pred else None
    sse = None
    if vals is not None:
        if vals.dtype != row_table.dtype or vals.numel() != n_pairs:
            raise ValueError('vals must match the tables dtype and the pair count')
        sse = torch.zeros(1, dtype=torch.float64, device=device)
    _lib.check(lib.alsmi_sddmm(_lib.dtype_code(np_dtype), int(rank), n_pairs, _ptr(users),
                               _ptr(items), _ptr(row_table), _ptr(col_table), _ptr(vals),
                               _ptr(pred), _ptr(sse), _ptr(_sddmm_workspace(device)), _stream()),
               'alsmi_sddmm')
    return pred, sse


_sparse_ws = {}


def _sparse_workspace(n_entries, device):
    """Grow-only workspace for the structure builders (one per device)."""
    lib = _lib.load()
    need = int(lib.alsmi_sparse_workspace_bytes(int(n_entries)))
    if need == 0:
        raise RuntimeError('alsmi_sparse_workspace_bytes failed (no HIP device, or too many '
                           'entries: %d)' % n_entries)
    key = (device.type, device.index)
    ws = _sparse_ws.get(key)
    if ws is None or ws.numel() * 8 < need:
        ws = torch.empty(need // 8 + 1, dtype=torch.float64, device=device)
        _sparse_ws[key] = ws
    return ws


def upload_pairs(users, items, shape, device):
    """Host (users, items) -> device int32 tensors with the reference's index rules (negative
    indices wrap, out-of-range raises IndexError)."""
    n, m = int(shape[0]), int(shape[1])
    users = np.asarray(users).ravel()
    items = np.asarray(items).ravel()
    if users.size != items.size:
        raise ValueError('users and items must have the same length')
    if users.size and (users.min() < -n or users.max() >= n or
                       items.min() < -m or items.max() >= m):
        raise IndexError('index out of bounds for shape {}'.format((n, m)))
    d_users = torch.from_numpy(np.ascontiguousarray(users)).to(device)
    d_items = torch.from_numpy(np.ascontiguousarray(items)).to(device)
    d_users = torch.where(d_users < 0, d_users + n, d_users).to(torch.int32)
    d_items = torch.where(d_items < 0, d_items + m, d_items).to(torch.int32)
    return d_users, d_items


def upload_triplets(users, items, y, shape, device):
    """Host (users, items, y) -> device (int32, int32, float64); see upload_pairs."""
    y = np.asarray(y, dtype=np.float64).ravel()
    if not (np.size(users) == np.size(items) == y.size):
        raise ValueError('users, items and y must have the same length')
    d_users, d_items = upload_pairs(users, items, shape, device)
    return d_users, d_items, torch.from_numpy(y).to(device)


def build_csr_device(d_users, d_items, d_y, shape, dtype):
    """Device triplets -> CSR on the device, with the reference's lil-fill semantics
    (src/als.py:137-140).  Returns device tensors (indptr int64 [n+1], indices int32 [nnz],
    vals dtype [nnz])."""
    n, m = int(shape[0]), int(shape[1])
    device = d_users.device
    n_in = int(d_users.numel())
    td = torch_dtype(dtype)
    indptr = torch.empty(n + 1, dtype=torch.int64, device=device)
    indices = torch.empty(max(n_in, 1), dtype=torch.int32, device=device)
    vals = torch.empty(max(n_in, 1), dtype=td, device=device)
    nnz = torch.zeros(1, dtype=torch.int64, device=device)
    ws = _sparse_workspace(n_in, device)
    _lib.check(_lib.load().alsmi_build_csr(
        _lib.dtype_code(dtype), n, m, n_in, _ptr(d_users), _ptr(d_items), _ptr(d_y), _ptr(indptr),
        _ptr(indices), _ptr(vals), _ptr(nnz), _ptr(ws), _stream()), 'alsmi_build_csr')
    count = int(nnz.cpu()[0])
    return indptr, indices[:count].contiguous(), vals[:count].contiguous()


def build_csr(users, items, y, shape, dtype, device):
    """Ratings triplets (host arrays) -> CSR on the device; see build_csr_device."""
    return build_csr_device(*upload_triplets(users, items, y, shape, device), shape, dtype)


def csr_to_csc(indptr, indices, vals, shape):
    """Device CSR -> device CSC (the CSR of the transpose), bit-identical to scipy's
    csr_tocsc (src/als.py:275-279)."""
    n, m = int(shape[0]), int(shape[1])
    device = indptr.device
    nnz = int(indices.numel())
    dtype = np.float32 if vals.dtype == torch.float32 else np.float64
    cptr = torch.empty(m + 1, dtype=torch.int64, device=device)
    cidx = torch.empty(nnz, dtype=torch.int32, device=device)
    cvals = torch.empty(nnz, dtype=vals.dtype, device=device)
    ws = _sparse_workspace(nnz, device)
    _lib.check(_lib.load().alsmi_csr_to_csc(
        _lib.dtype_code(dtype), n, m, nnz, _ptr(indptr), _ptr(indices), _ptr(vals), _ptr(cptr),
        _ptr(cidx), _ptr(cvals), _ptr(ws), _stream()), 'alsmi_csr_to_csc')
    return cptr, cidx, cvals


def topk_items(rank, users, row_table, col_table, k, max_workspace_bytes=1 << 30):
    """The k best items of each user in `users` (int32 device tensor): (items int32 [n x k],
    scores [n x k]) device tensors, best first, ties by the smaller item id.  The users are
    processed in batches whose dense score block stays under `max_workspace_bytes`."""
    lib = _lib.load()
    device = row_table.device
    n_sel, n_items = int(users.numel()), int(col_table.shape[0])
    if users.dtype != torch.int32:
        raise ValueError('users must be an int32 device tensor')
    if not 1 <= k <= min(1024, n_items):
        raise ValueError('k must be between 1 and min(1024, number of items), got %r' % (k,))
    np_dtype = np.float32 if row_table.dtype == torch.float32 else np.float64
    code = _lib.dtype_code(np_dtype)
    items = torch.empty((n_sel, k), dtype=torch.int32, device=device)
    scores = torch.empty((n_sel, k), dtype=row_table.dtype, device=device)
    per_user = max(int(lib.alsmi_topk_workspace_bytes(code, 1, n_items)), 1)
    batch = max(1, min(n_sel, max_workspace_bytes // per_user))
    ws = torch.empty(int(lib.alsmi_topk_workspace_bytes(code, batch, n_items)) // 8 + 1,
                     dtype=torch.float64, device=device)
    for lo in range(0, n_sel, batch):
        hi = min(n_sel, lo + batch)
        _lib.check(lib.alsmi_topk_items(code, int(rank), hi - lo, _ptr(users[lo:hi]),
                                        _ptr(row_table), _ptr(col_table), n_items, int(k),
                                        _ptr(items[lo:hi]), _ptr(scores[lo:hi]), _ptr(ws),
                                        _stream()), 'alsmi_topk_items')
    return items, scores
