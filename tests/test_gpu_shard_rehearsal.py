"""ALS.fit_sharded and DirtyIMC.fit_sharded with THREE ranks that share the box's one GPU
under gloo (every rank brings a shuffled third of the ratings; the entry exchange, the row
gathers and the all-reduces really cross process boundaries) against the single-process fits
of the same data."""
import os
import socket
import subprocess
import sys
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_three_ranks_on_one_gpu_match_the_single_process_fits(tmp_path):
    sys.path.insert(0, HERE)
    from shard_rehearsal_worker import case
    from alsmi import ALS, DirtyIMC
    out = str(tmp_path / 'rank0.npz')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '3',
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()),
           os.path.join(HERE, 'shard_rehearsal_worker.py'), out]
    env = dict(os.environ, OMP_NUM_THREADS='2')
    run = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert run.returncode == 0, run.stderr[-3000:]
    got = np.load(out)
    n, m, X, Y, users, items, vals, stars = case()
    ref = ALS(rank=6, random_state=2, max_iter=4, dtype='float64').fit((users, items), stars,
                                                                       shape=(n, m))
    # a row's arithmetic does not depend on which rank owns it or on how the entries arrived
    assert np.array_equal(got['uf'], ref.user_feats)
    assert np.array_equal(got['itf'], ref.item_feats)
    np.testing.assert_allclose(got['hist'], ref.rmse_history_, rtol=1e-12)
    np.testing.assert_allclose(got['err'], ref.reconstruction_err_, rtol=1e-12)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        d = DirtyIMC(rank=4, tol=1e-4, random_state=3, imc_alpha=1e-6, als_alpha=0.05)
        d.fit((X, Y, users, items), vals, shape=(n, m))
    np.testing.assert_allclose(got['Z'], d.imc.Z, rtol=1e-5, atol=1e-7)
    assert int(got['n_iter']) == d.als.n_iter_
    np.testing.assert_allclose(got['duf'], d.als.user_feats, rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(got['dif'], d.als.item_feats, rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(got['derr'], d.reconstruction_err_, rtol=1e-5)
