"""Worker of tests/test_gpu_shard_rehearsal.py: one of several ranks that share the one GPU of
the box under gloo (the exchanges then travel through host copies).  Every rank holds a
shuffled part of the ratings / samples, runs ALS.fit_sharded and DirtyIMC.fit_sharded and
rank 0 writes what it ended with."""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, 'recommendation-models_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)


def case():
    rng = np.random.default_rng(33)
    n, m, p, q, ns = 500, 260, 5, 7, 15000
    flat = rng.permutation(n * m)[:ns]
    users, items = flat // m, flat % m
    X, Y = rng.standard_normal((n, p)), rng.standard_normal((m, q))
    Zs = rng.standard_normal((p, q)) * 0.3
    P3, Q3 = rng.standard_normal((n, 3)) * 0.5, rng.standard_normal((m, 3)) * 0.5
    vals = np.einsum('ij,ij->i', X[users].dot(Zs), Y[items]) + \
        np.einsum('ij,ij->i', P3[users], Q3[items]) + rng.standard_normal(ns) * 0.05
    stars = np.clip(np.round(2 * (3 + vals)) / 2, 0.5, 5.0)
    return n, m, X, Y, users, items, vals, stars


def main():
    import torch
    import torch.distributed as dist
    from alsmi import ALS, DirtyIMC
    out = sys.argv[1]
    if os.environ.get('ALSMI_WATCHDOG'):   # debugging aid: traceback + exit after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ['ALSMI_WATCHDOG']), exit=True)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo')
    world, me = dist.get_world_size(), dist.get_rank()
    n, m, X, Y, users, items, vals, stars = case()
    owner = np.random.default_rng(1).permutation(users.size) % world
    mine = owner == me
    als = ALS(rank=6, random_state=2, max_iter=4, dtype='float64').fit_sharded(
        users[mine], items[mine], stars[mine], (n, m))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        d = DirtyIMC(rank=4, tol=1e-4, random_state=3, imc_alpha=1e-6, als_alpha=0.05)
        d.fit_sharded((X, Y, users[mine], items[mine]), vals[mine], (n, m))
    if me == 0:
        np.savez(out, uf=als.user_feats, itf=als.item_feats, hist=als.rmse_history_,
                 err=als.reconstruction_err_, Z=d.imc.Z, duf=d.als.user_feats,
                 dif=d.als.item_feats, derr=d.reconstruction_err_, n_iter=d.als.n_iter_)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
