"""bench.py's multi-rank path rehearsed on CPU: two ranks under torch.distributed.run with the
gloo backend and the oracle-backed engine backend (ALSMI_BENCH_BACKEND), against a
single-process run of the same two-block problem."""
import json
import os
import socket
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def run_bench(world, extra=()):
    env = dict(os.environ)
    env['ALSMI_BENCH_BACKEND'] = 'oracle_backend:OracleBackend'
    env['PYTHONPATH'] = os.pathsep.join([HERE, ROOT, env.get('PYTHONPATH', '')])
    env['OMP_NUM_THREADS'] = '1'
    cmd = [sys.executable]
    if world > 1:
        cmd += ['-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world),
                '--master-addr', '127.0.0.1', '--master-port', str(free_port())]
    cmd += [os.path.join(ROOT, 'bench.py'), '--gpus', str(world), '--steps', '2', '--warmup', '1',
            '--config', 'tiny', '--cpu-frac', '0'] + list(extra)
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_two_rank_bench_line_matches_single_process_sweeps():
    import torch
    sys.path[:0] = [HERE, ROOT, os.path.join(ROOT, 'recommendation-models_amd')]
    import bench
    from alsmi import synth
    from alsmi.engine import AlsEngine
    from oracle_backend import OracleBackend
    line = run_bench(2)
    cfg = bench.CONFIGS['tiny']
    assert line['n_gpus'] == 2 and line['steps'] == 2 and line['scaling'] == 'weak'
    assert line['unit'] == 'nnz/s' and line['config']['nnz'] == 2 * cfg['nnz']
    assert line['config']['users'] == 2 * cfg['n'] and line['value'] > 0
    assert abs(line['value'] - line['config']['nnz'] * 1e3 / line['ms_per_step']) < 1e-6 * line['value']
    # the same two blocks in one process: 3 sweeps, the error of the last one
    P = synth.sharded_problem(20242, cfg['n'], cfg['m'], cfg['nnz'], 1, 0, torch.device('cpu'),
                              n_blocks=2)
    eng = AlsEngine(P['csr'], P['csc'], P['shape'], cfg['rank'], 0.1, backend=OracleBackend(),
                    total_nnz=P['total_nnz'])
    bench.init_factors(eng, cfg['rank'], torch.device('cpu'))
    for _ in range(3):
        eng.sweep()
    np.testing.assert_allclose(line['train_rmse']['last'], eng.train_rmse(), rtol=1e-9)
    assert line['train_rmse']['first'] > line['train_rmse']['last']


def test_single_rank_bench_line_fields():
    line = run_bench(1, ['--separate-error'])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step',
                'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline'):
        assert key in line, key
    # a stand-in backend identifies itself and reports no roofline (it times no kernel)
    assert line['backend'].startswith('stand-in') and line['roofline'] is None
    assert 'workload' in line['config'] and line['vs_baseline'] is None
