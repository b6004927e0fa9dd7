"""alsmi_sddmm_noalloc (the entry point that allocates nothing: its partial sums live in a
device array of the library image) against alsmi_sddmm with a caller workspace, through the
C ABI, bit for bit, on two calls in a row of which the second is smaller — the second call
must not see anything the first left in the shared array."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _call(lib, noalloc, dtype_code, rank, users, items, U, V, vals, ws, torch):
    n = int(users.numel())
    pred = torch.full((n,), -7.0, dtype=U.dtype, device=U.device)
    sse = torch.full((1,), -1.0, dtype=torch.float64, device=U.device)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    if noalloc:
        rc = lib.alsmi_sddmm_noalloc(dtype_code, rank, n, vp(users), vp(items), vp(U), vp(V),
                                     vp(vals), vp(pred), vp(sse), st)
    else:
        rc = lib.alsmi_sddmm(dtype_code, rank, n, vp(users), vp(items), vp(U), vp(V), vp(vals),
                             vp(pred), vp(sse), vp(ws), st)
    assert rc == 0, lib.alsmi_last_error()
    torch.cuda.synchronize()
    return pred, sse


@pytest.mark.parametrize('dtype,rank', [('float32', 64), ('float64', 10), ('float32', 128)])
def test_noalloc_sddmm_equals_workspace_sddmm(dtype, rank):
    import torch
    from alsmi import _lib, device as dev
    device = dev.require_gpu()
    lib = _lib.load()
    code = _lib.dtype_code(dtype)
    td = dev.torch_dtype(dtype)
    kp = dev.padded_rank(rank)
    g = torch.Generator(device=device)
    g.manual_seed(5 + rank)
    n, m = 5000, 3000
    U = torch.zeros((n, kp), dtype=td, device=device)
    V = torch.zeros((m, kp), dtype=td, device=device)
    U[:, :rank] = torch.randn((n, rank), generator=g, device=device, dtype=td)
    V[:, :rank] = torch.randn((m, rank), generator=g, device=device, dtype=td)
    ws = torch.zeros(int(lib.alsmi_sddmm_workspace_bytes()) // 8 + 1, dtype=torch.float64,
                     device=device)
    # first call: enough pairs that every block of the capped grid writes a partial
    # (2048 blocks x 256 pairs); second call: a handful of pairs, far fewer blocks
    free_before = None
    for n_pairs in (2048 * 256 * 2 + 77, 1000, 3):
        users = torch.randint(0, n, (n_pairs,), generator=g, device=device).to(torch.int32)
        items = torch.randint(0, m, (n_pairs,), generator=g, device=device).to(torch.int32)
        vals = torch.randn(n_pairs, generator=g, device=device, dtype=td)
        want_pred, want_sse = _call(lib, False, code, rank, users, items, U, V, vals, ws, torch)
        pred, sse = _call(lib, True, code, rank, users, items, U, V, vals, None, torch)
        if free_before is not None:
            # after the first (warm-up) call on this device a call takes no device memory:
            # the driver's free-byte count is the same before and after it (the tensors of
            # this test come from torch's cache, which holds its blocks across the calls)
            torch.cuda.synchronize()
            before = torch.cuda.mem_get_info()[0]
            vp = lambda t: ctypes.c_void_p(t.data_ptr())
            for _ in range(3):                      # (into the outputs at hand: no torch allocation)
                rc = lib.alsmi_sddmm_noalloc(code, rank, n_pairs, vp(users), vp(items), vp(U),
                                             vp(V), vp(vals), vp(pred), vp(sse), None)
                assert rc == 0
            torch.cuda.synchronize()
            assert torch.cuda.mem_get_info()[0] == before
        free_before = True
        assert torch.equal(pred, want_pred)
        assert float(sse.cpu()[0]) == float(want_sse.cpu()[0])
        assert float(sse.cpu()[0]) > 0


def test_noalloc_sddmm_limits():
    import torch
    from alsmi import _lib, device as dev
    dev.require_gpu()
    lib = _lib.load()
    rc = lib.alsmi_sddmm_noalloc(0, 8, (1 << 28) + 1, None, None, None, None, None, None, None,
                                 None)
    assert rc == -1 and b'limit' in lib.alsmi_last_error()
    # the empty call writes a zero sum
    sse = torch.full((1,), 5.0, dtype=torch.float64, device='cuda')
    rc = lib.alsmi_sddmm_noalloc(0, 8, 0, None, None, None, None, None, None,
                                 ctypes.c_void_p(sse.data_ptr()), None)
    assert rc == 0 and float(sse.cpu()[0]) == 0.0
