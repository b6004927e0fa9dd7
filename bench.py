#!/usr/bin/env python3
"""ALS sweep throughput on synthetic MovieLens-20M-shaped ratings (BASELINE config 2).

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one ALS sweep: user half-sweep, item half-sweep, training squared error over
all stored entries (what one iteration of the reference's loop does, src/als.py:154-167).
The squared error of a sweep is produced by the user half of the NEXT step, which reads
the same ratings and item factors anyway (the schedule ALS.fit uses); the error of the last
timed sweep is a standalone pass inside the timed region, so K timed steps contain K user
halves, K item halves and K+1 error sums.
With N ranks the problem is N user blocks of the single-GPU size (weak scaling): every
rank updates its own users and its slice of the items, each half-sweep ends with an
all-gather of the updated factor rows.  Inputs are resident in HBM before the timed
region.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'recommendation-models_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

CONFIGS = {
    # name: users per GPU, items, stored entries per GPU, rank
    'c2': dict(n=138493, m=26744, nnz=20000000, rank=64,
               workload='ALS rank 64, synthetic ML-20M shape 138493x26744, 20M nnz per GPU'),
    'r128': dict(n=138493, m=26744, nnz=20000000, rank=128,
                 workload='ALS rank 128, synthetic ML-20M shape 138493x26744, 20M nnz per GPU '
                          '(single-GPU slice of the rank-128 configuration)'),
    'c4slice': dict(n=1250000, m=1000000, nnz=125000000, rank=128,
                    workload='ALS rank 128, one GPU\'s share of the 10M x 1M, 1B-entry configuration '
                             'at 8 GPUs: 1.25M users x 1M items, 125M nnz'),
    'small': dict(n=20000, m=5000, nnz=1000000, rank=64,
                  workload='ALS rank 64, synthetic 20000x5000, 1M nnz per GPU (smoke size)'),
}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec


def half_sweep_bytes(nnz, rows, kp, s):
    """Algorithmic bytes of one half-sweep: per stored entry one factor row, its int32
    index and its rating; per updated row the written vector and its two row pointers."""
    return nnz * (kp * s + 4 + s) + rows * (kp * s + 8)


def sddmm_bytes(nnz, rows, kp, s):
    """Training-error pass with row reuse: per entry one item factor row, two int32
    indices and the rating; per user row its factor vector once."""
    return nnz * (kp * s + 8 + s) + rows * kp * s


def cpu_baseline(engine, frac, threads):
    """Time the C restatement of the reference (oracle/als_oracle.c) on the host cores
    for a bounded sample of the same workload and extrapolate to entries per second."""
    from oracle import c_oracle as co
    k = engine.rank
    U = engine.U[:, :k].to(torch.float64).cpu().numpy()
    V = engine.V[:, :k].to(torch.float64).cpu().numpy()
    out = {}
    times = {}
    for name, side, src in (('user', engine.csr, V), ('item', engine.csc, U)):
        rows = max(1, int(side.n_rows * frac))
        indptr = side.indptr_host[:rows + 1]
        cnt = int(indptr[-1])
        idx = side.indices[:cnt].cpu().numpy()
        val = side.vals[:cnt].to(torch.float64).cpu().numpy()
        t0 = time.perf_counter()
        co.half_sweep(indptr, idx, val, src, engine.alpha, threads=threads)
        times[name] = (time.perf_counter() - t0) * side.nnz / max(cnt, 1)
        out[name + '_rows'] = rows
        out[name + '_entries'] = cnt
        if name == 'user':
            t0 = time.perf_counter()
            co.sse(indptr, idx, val, U, V, threads=threads)
            times['sse'] = (time.perf_counter() - t0) * side.nnz / max(cnt, 1)
    sweep_s = times['user'] + times['item'] + times['sse']
    return dict(value=engine.local_nnz / sweep_s, unit='nnz/s', cores=threads, kind='port',
                sweeps_per_sec=1.0 / sweep_s,
                sample='first %d user rows (%d entries) + first %d item rows (%d entries) of the '
                       'same matrix, fp64 C port with OpenMP, extrapolated by entries'
                       % (out['user_rows'], out['user_entries'], out['item_rows'],
                          out['item_entries']))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--config', default='c2', choices=sorted(CONFIGS))
    ap.add_argument('--dtype', default='float32', choices=['float32', 'float64'])
    ap.add_argument('--chunk', type=int, default=0)
    ap.add_argument('--separate-error', action='store_true',
                    help='compute the training error in its own pass after every sweep '
                         'instead of inside the next user half-sweep (A/B aid)')
    ap.add_argument('--cpu-frac', type=float, default=0.25,
                    help='fraction of rows the CPU baseline is timed on (0 disables it)')
    args = ap.parse_args()

    if os.environ.get('ALSMI_BENCH_WATCHDOG'):  # debugging aid: traceback + exit after N s
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ['ALSMI_BENCH_WATCHDOG']), exit=True)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank_id = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run --nproc-per-node %d' % args.gpus)
        args.gpus = world
    # ALSMI_DIST_BACKEND=gloo lets several ranks share one GPU (a rehearsal of the multi-rank
    # path on a one-GPU box; RCCL wants one device per rank)
    backend = os.environ.get('ALSMI_DIST_BACKEND', 'nccl')
    if backend != 'nccl':
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    group = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(backend)
        group = dist.group.WORLD

    from alsmi import synth
    from alsmi.engine import AlsEngine
    cfg = CONFIGS[args.config]
    t_gen = time.perf_counter()
    prob = synth.sharded_problem(20242, cfg['n'], cfg['m'], cfg['nnz'], world, rank_id, device)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen
    engine = AlsEngine(prob['csr'], prob['csc'], prob['shape'], cfg['rank'], 0.1,
                       dtype=args.dtype, user_bounds=prob['user_bounds'],
                       item_bounds=prob['item_bounds'], group=group, chunk=args.chunk,
                       total_nnz=prob['total_nnz'])
    del prob
    # initial factors as in src/als.py:143-152 (row 0 of the item factors = mean rating)
    m = engine.m
    csc = engine.csc
    cnt = torch.diff(csc.indptr).to(torch.float64)
    # per-item sums as differences of a running sum over the item-ordered ratings (an
    # index_add_ on the few popular items serialises on atomics: 15 s at 125 M entries)
    run = torch.zeros(csc.nnz + 1, dtype=torch.float64, device=device)
    torch.cumsum(csc.vals.to(torch.float64), 0, out=run[1:])
    sums = run[csc.indptr[1:]] - run[csc.indptr[:-1]]
    del run
    avg = torch.where(cnt > 0, sums / cnt.clamp_min(1), torch.zeros_like(sums))
    init = torch.from_numpy(np.random.RandomState(42).rand(cfg['rank'], m).T.copy()).to(device)
    engine.V.zero_()
    engine.V[:, :cfg['rank']] = init.to(engine.V.dtype)
    engine.V[engine.ilo:engine.ihi, 0] = avg.to(engine.V.dtype)
    engine._all_gather_rows(engine.V, engine.item_bounds)
    engine.U.zero_()

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    fused = not args.separate_error
    sse_log = []

    def step(events=None):
        if events:
            events[0].record()
        if fused:
            sse_log.append(engine.half_sweep_users_err())  # error of the state before
        else:
            engine.half_sweep_users()
        if events:
            events[1].record()
        engine.half_sweep_items()
        if events:
            events[2].record()
        if not fused:
            sse_log.append(engine.train_sse_async())

    for _ in range(args.warmup):
        step()
    engine.train_sse_async()  # untimed: first-use set-up of the standalone error pass
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    ev_tail = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        step(ev[s])
    ev_tail[0].record()
    if fused:
        sse_log.append(engine.train_sse_async())  # error of the last timed sweep
    ev_tail[1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.cpu()[0])

    rmse = [float(np.sqrt(float(x.cpu()[0]) / engine.nnz)) for x in sse_log]
    t_user = float(np.mean([ev[s][0].elapsed_time(ev[s][1]) for s in range(args.steps)])) * 1e-3
    t_item = float(np.mean([ev[s][1].elapsed_time(ev[s][2]) for s in range(args.steps)])) * 1e-3
    t_sse = ev_tail[0].elapsed_time(ev_tail[1]) * 1e-3 if fused else None

    if rank_id == 0:
        s = 4 if args.dtype == 'float32' else 8
        kp = engine.kp
        b_user = half_sweep_bytes(engine.csr.nnz, engine.csr.n_rows, kp, s)
        if fused:
            b_user += engine.csr.n_rows * kp * s  # the replaced vectors are read for the error
        b_item = half_sweep_bytes(engine.csc.nnz, engine.csc.n_rows, kp, s)
        b_sse = sddmm_bytes(engine.csr.nnz, engine.csr.n_rows, kp, s)
        achieved = (b_user + b_item) / (t_user + t_item) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tpath) and world == 1 and args.config == 'c2':
            try:
                traffic = json.load(open(tpath)).get('half_sweep_hbm_bytes_per_launch')
            except Exception:
                traffic = None
        sweeps_per_sec = args.steps / elapsed
        line = {
            'metric': 'ALS nnz/sec (stored entries x sweeps/sec), ML-20M shape rank %d' % cfg['rank'],
            'value': engine.nnz * sweeps_per_sec,
            'unit': 'nnz/s',
            'sweeps_per_sec': sweeps_per_sec,
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32' if args.dtype == 'float32' else 'f64', 'data': 'synthetic',
            'config': {'workload': cfg['workload'], 'users': engine.n, 'items': engine.m,
                       'nnz': engine.nnz, 'rank': cfg['rank'], 'alpha': 0.1,
                       'parallelism': 'row-sharded x%d' % world},
            'roofline': {
                'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                'kernel': 'half-sweep launch = gram_chunks_kernel + solve_rows_kernel, '
                          'mean of user side and item side',
                'bytes_per_launch': (b_user + b_item) / 2.0,
                'ms_per_launch': (t_user + t_item) / 2.0 * 1e3,
                'user_side': {'ms': t_user * 1e3, 'GBps': b_user / t_user / 1e9},
                'item_side': {'ms': t_item * 1e3, 'GBps': b_item / t_item / 1e9},
            },
            'error_pass': ({'schedule': 'fused into the next user half-sweep; one standalone '
                                        'pass after the last timed sweep',
                            'standalone_ms': t_sse * 1e3,
                            'standalone_GBps': b_sse / t_sse / 1e9} if fused else
                           {'schedule': 'standalone pass after every sweep'}),
            'train_rmse': {'first': rmse[0] if rmse else None, 'last': rmse[-1] if rmse else None},
            'setup_s': {'generate': t_gen},
        }
        if world == 1 and args.cpu_frac > 0:
            threads = int(os.environ.get('OMP_NUM_THREADS', '0')) or min(16, os.cpu_count() or 1)
            try:
                line['cpu_baseline'] = cpu_baseline(engine, args.cpu_frac, threads)
            except Exception as exc:  # the checker is optional for the number itself
                line['cpu_baseline'] = {'value': None, 'unit': 'nnz/s', 'cores': threads,
                                        'kind': 'port', 'sample': 'failed: %r' % (exc,)}
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
