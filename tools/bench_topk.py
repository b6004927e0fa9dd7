#!/usr/bin/env python3
"""Top-k recommendation throughput: the k best items of every user at the ML-20M shape
(alsmi_topk_items, scores = predict_all's).  python tools/bench_topk.py [--dtype D] [--k K]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'recommendation-models_amd')):
    sys.path.insert(0, p)
import numpy as np
import torch
from alsmi import device as dev

ap = argparse.ArgumentParser()
ap.add_argument('--dtype', default='float32')
ap.add_argument('--k', type=int, default=10)
ap.add_argument('--rank', type=int, default=64)
ap.add_argument('--users', type=int, default=138493)
ap.add_argument('--items', type=int, default=26744)
ap.add_argument('--check', type=int, default=0, help='compare this many sampled users with numpy')
ap.add_argument('--ws-mb', type=int, default=0, help='score workspace per batch in MiB (0: the default)')
a = ap.parse_args()
d = dev.require_gpu()
td = torch.float64 if a.dtype == 'float64' else torch.float32
kp = dev.padded_rank(a.rank)
g = torch.Generator(device=d); g.manual_seed(0)
U = torch.zeros((a.users, kp), dtype=td, device=d); U[:, :a.rank] = torch.rand((a.users, a.rank), generator=g, device=d, dtype=td)
V = torch.zeros((a.items, kp), dtype=td, device=d); V[:, :a.rank] = torch.rand((a.items, a.rank), generator=g, device=d, dtype=td)
users = torch.arange(a.users, dtype=torch.int32, device=d)
dev.topk_items(a.rank, users[:1000], U, V, a.k); torch.cuda.synchronize()
t0 = time.perf_counter()
kw = dict(max_workspace_bytes=a.ws_mb << 20) if a.ws_mb else {}
items, scores = dev.topk_items(a.rank, users, U, V, a.k, **kw)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
s = a.dtype == 'float64' and 8 or 4
print(json.dumps({'workload': 'top-%d of %d items for %d users, rank %d, %s' % (a.k, a.items, a.users, a.rank, a.dtype),
                  'seconds': dt, 'users_per_s': a.users / dt,
                  'score_GFLOPs': 2.0 * a.users * a.items * a.rank / dt / 1e9,
                  'score_block_GBps': 2.0 * a.users * a.items * s / dt / 1e9}))
if a.check:
    rng = np.random.default_rng(0)
    pick = rng.choice(a.users, size=min(a.check, a.users), replace=False)
    Uh = U[:, :a.rank].to(torch.float64).cpu().numpy()
    Vh = V[:, :a.rank].to(torch.float64).cpu().numpy()
    S = Uh[pick].dot(Vh.T)
    got_i = items[torch.from_numpy(pick).to(d)].cpu().numpy().astype(np.int64)
    got_s = scores[torch.from_numpy(pick).to(d)].to(torch.float64).cpu().numpy()
    eps = 2.0 ** -52 if a.dtype == 'float64' else 2.0 ** -23
    tol = 8 * np.sqrt(a.rank) * eps * np.abs(S).max()
    worst = 0.0
    for r in range(pick.size):
        assert len(set(got_i[r])) == a.k
        worst = max(worst, np.abs(S[r, got_i[r]] - got_s[r]).max())
        assert (np.diff(got_s[r]) <= 0).all()
        rest = np.delete(S[r], got_i[r])
        assert rest.max() <= got_s[r, -1] + 2 * tol, (r, rest.max(), got_s[r, -1])
    assert worst <= tol, (worst, tol)
    print('checked %d users against numpy: score difference at most %.2e (allowed %.2e)' % (pick.size, worst, tol))
