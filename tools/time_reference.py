#!/usr/bin/env python3
"""Time the upstream Python reference itself (imported from its checkout, never copied) on
this machine's CPU: full ALS sweeps at its own README-sized case (1000 x 500, 25 000 entries,
rank 10) and at a mid-size case (20 000 x 5 000, 1 M entries, rank 64), one process, the
way its loop body runs a sweep (src/als.py:155-165).  Prints one JSON line per case.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python3 /root/repo/tools/time_reference.py [--mid]

Only usable where the reference checkout exists (``ALSMI_REFERENCE``, default
/root/reference); nothing on the GPU box calls this.
"""
import argparse
import json
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

import numpy as np  # noqa: E402
import scipy.sparse as sps  # noqa: E402

import make_golden as mg  # noqa: E402  (applies the scipy.sparse.sputils patch, imports src)


def time_case(name, n, m, nnz, rank, sweeps, item_sample=None):
    users, items, vals = mg.random_ratings(11, n, m, nnz)
    from sklearn.utils.validation import check_random_state
    als = mg.ALS(rank=rank, alpha=0.1, random_state=0)
    data = sps.csr_matrix((vals, (users, items)), shape=(n, m))
    rs = check_random_state(0)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        item_avg = data.sum(0) / (data != 0).sum(0)
    item_avg[np.isnan(item_avg)] = 0
    als.item_feats = rs.rand(rank, m)
    als.item_feats[0] = item_avg
    als.user_feats = np.zeros((rank, n))
    als.data = data
    t_user = t_item = 0.0
    item_ids = np.arange(m) if item_sample is None else np.arange(min(m, item_sample))
    for _ in range(sweeps):
        t0 = time.perf_counter()
        als._update_parallel(np.array_split(np.arange(n), 1))
        t1 = time.perf_counter()
        als._update_parallel(np.array_split(item_ids, 1), user=False)
        t2 = time.perf_counter()
        t_user += t1 - t0
        t_item += (t2 - t1) * m / len(item_ids)
    per = (t_user + t_item) / sweeps
    print(json.dumps({
        'case': name, 'n_users': n, 'n_items': m, 'nnz': nnz, 'rank': rank, 'sweeps': sweeps,
        'user_half_s': t_user / sweeps, 'item_half_s': t_item / sweeps,
        'item_half': 'all columns' if item_sample is None else
                     'first %d columns, extrapolated to %d' % (len(item_ids), m),
        'seconds_per_sweep': per, 'sweeps_per_sec': 1 / per, 'nnz_per_sec': nnz / per,
        'cores': 1, 'kind': 'reference'}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mid', action='store_true', help='also the 20 000 x 5 000 rank-64 case')
    args = ap.parse_args()
    time_case('config1', 1000, 500, 25000, 10, 3)
    if args.mid:
        time_case('mid', 20000, 5000, 1000000, 64, 1, item_sample=500)


if __name__ == '__main__':
    main()
