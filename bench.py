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
With N ranks the default problem is N user blocks of the single-GPU size (weak scaling);
`--config c4` is the fixed 10 M x 1 M, 1 B-entry rank-128 matrix cut over the ranks (strong
scaling).  Every rank updates its own users and its slice of the items, each half-sweep ends
with an all-gather of the updated factor rows, overlapped slab by slab with the computation.
Inputs are resident in HBM before the timed region.  Rank 0 prints one JSON line.
"""
import argparse
import glob
import hashlib
import importlib
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
    # n, nnz: users and stored entries per BLOCK; blocks: None = one block per rank (weak
    # scaling), a number = that many blocks in total, shared out over the ranks (strong)
    'c2': dict(n=138493, m=26744, nnz=20000000, rank=64, blocks=None,
               workload='ALS rank 64, synthetic ML-20M shape 138493x26744, 20M nnz per GPU'),
    'r128': dict(n=138493, m=26744, nnz=20000000, rank=128, blocks=None,
                 workload='ALS rank 128, synthetic ML-20M shape 138493x26744, 20M nnz per GPU'),
    'c4slice': dict(n=1250000, m=1000000, nnz=125000000, rank=128, blocks=None,
                    workload='ALS rank 128, one GPU\'s share of the 10M x 1M, 1B-entry '
                             'configuration at 8 GPUs: 1.25M users x 1M items, 125M nnz per GPU'),
    'c4': dict(n=1250000, m=1000000, nnz=125000000, rank=128, blocks=8,
               workload='ALS rank 128, 10M users x 1M items, 1B nnz in total (8 user blocks of '
                        '1.25M users / 125M nnz shared out over the ranks)'),
    'small': dict(n=20000, m=5000, nnz=1000000, rank=64, blocks=None,
                  workload='ALS rank 64, synthetic 20000x5000, 1M nnz per GPU (smoke size)'),
    'tiny': dict(n=300, m=200, nnz=6000, rank=8, blocks=None,
                 workload='ALS rank 8, synthetic 300x200, 6000 nnz per rank (rehearsal size)'),
}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec
# Dense matrix-pipe peaks (FLOP/s): 16-bit matrix cores, fp32 (matrix = vector rate), fp64 matrix
PEAK_F16_MATRIX, PEAK_F32, PEAK_F64 = 2.5e15, 157.3e12, 78.6e12


def half_sweep_bytes(nnz, rows, kp, s):
    """Algorithmic bytes of one half-sweep: per stored entry one factor row, its int32
    index and its rating; per updated row the written vector and its two row pointers."""
    return nnz * (kp * s + 4 + s) + rows * (kp * s + 8)


def sddmm_bytes(nnz, rows, kp, s):
    """Training-error pass with row reuse: per entry one item factor row, two int32
    indices and the rating; per user row its factor vector once."""
    return nnz * (kp * s + 8 + s) + rows * kp * s


def side_flops(indptr_host, rank, kp, dtype):
    """Arithmetic one half-sweep of a side issues, from its row lengths.

    Gram build: only the upper tiles of a row's kp x kp matrix are formed,
    2 * entries * kp^2 * (NV+1)/(2 NV) with NV = kp/16; a short row (padded rank 64 or
    128, at most min(rank, 48 or 96) entries) forms the n x n matrix of its neighbours
    instead, n = its entries padded to 16.  Solve: 2 (s^3/3 + 2 s^2) per row of an s x s system.
    Returns (gram_flops, solve_flops, matrix products issued per Gram product): in fp32 every
    Gram product is 3 products of the two-way fp16 split on the 16-bit matrix cores (4 at
    padded rank 128, whose concatenated operands form all four)."""
    deg = np.diff(np.asarray(indptr_host)).astype(np.float64)
    nv = kp // 16
    tri = (nv + 1) / (2.0 * nv)
    f32 = np.dtype(dtype) == np.float32
    dual_max = min(rank, 48 if kp == 64 else 96) if kp in (64, 128) else 0
    short = (deg > 0) & (deg <= dual_max)
    n_pad = 16.0 * np.ceil(deg[short] / 16.0)
    nd = n_pad / 16.0
    gram = 2.0 * deg[~short].sum() * kp * kp * tri + (2.0 * n_pad * n_pad * kp * (nd + 1) / (2 * nd)).sum()
    s_long = float(kp)
    solve = 2.0 * ((deg > 0) & ~short).sum() * (s_long ** 3 / 3 + 2 * s_long ** 2) + \
        (2.0 * (n_pad ** 3 / 3 + 2 * n_pad ** 2)).sum()
    products = (4 if kp == 128 else 3) if f32 else 1
    return float(gram), float(solve), products


def matrix_roof_seconds(gram, solve, products, dtype):
    """Time the matrix pipes need for that arithmetic at their dense peaks: fp32 Gram products
    on the 16-bit matrix cores, the fp32 solve at the fp32 matrix/vector rate; fp64 everything
    on the fp64 matrix pipe."""
    if np.dtype(dtype) == np.float32:
        return gram * products / PEAK_F16_MATRIX + solve / PEAK_F32
    return (gram + solve) / PEAK_F64


def kernel_source_sha():
    """Hash of the half-sweep kernels' sources (and of their build recipe): counters measured
    on another build of them are not reported."""
    h = hashlib.sha256()
    for name in ('als_sweep.hip', 'alsmi_common.h', 'alsmi_common.hip', 'Makefile'):
        h.update(open(os.path.join(ROOT, 'recommendation-models_amd', 'csrc', name), 'rb').read())
    return h.hexdigest()[:16]


def measured_counters(config, dtype):
    """HBM traffic and MFMA utilisation of this configuration's half-sweep kernels from the
    newest committed counter summary (profiles/rNN_summary.json, written by
    tools/profile_collect.py from rocprofv3 --pmc passes of this very command) — only when it
    was measured on the kernel sources that are running now."""
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_summary.json')), reverse=True):
        try:
            doc = json.load(open(path))
        except Exception:
            continue
        entry = doc.get('configs', {}).get('%s/%s' % (config, dtype))
        if entry is None:
            continue
        if doc.get('kernel_source_sha') != kernel_source_sha():
            return {'stale': 'counters in %s were measured on other kernel sources (%s)'
                             % (os.path.basename(path), doc.get('commit'))}
        entry = dict(entry)
        entry['measured_at_commit'] = doc.get('commit')
        entry['file'] = 'profiles/' + os.path.basename(path)
        return entry
    return None


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


class Stamp(object):
    """A point in time on the device's current stream (a wall-clock stamp without a GPU)."""

    def __init__(self, cuda):
        self.cuda = cuda
        self.ev = torch.cuda.Event(enable_timing=True) if cuda else None
        self.t = None

    def record(self):
        if self.cuda:
            self.ev.record()
        else:
            self.t = time.perf_counter()

    def ms_until(self, other):
        return self.ev.elapsed_time(other.ev) if self.cuda else (other.t - self.t) * 1e3


def init_factors(engine, rank, device):
    """Initial factors as in src/als.py:143-152: uniform item factors with row 0 = mean
    rating of the item, zero user factors."""
    m = engine.m
    csc = engine.csc
    indptr = csc.indptr_host
    cnt = torch.from_numpy(np.diff(indptr)).to(device=device, dtype=torch.float64)
    # per-item sums as differences of a running sum over the item-ordered ratings (an
    # index_add_ on the few popular items serialises on atomics: 15 s at 125 M entries)
    run = torch.zeros(csc.nnz + 1, dtype=torch.float64, device=device)
    torch.cumsum(csc.vals.to(torch.float64), 0, out=run[1:])
    ptr = torch.from_numpy(indptr).to(device)
    sums = run[ptr[1:]] - run[ptr[:-1]]
    del run
    avg = torch.where(cnt > 0, sums / cnt.clamp_min(1), torch.zeros_like(sums))
    init = torch.from_numpy(np.random.RandomState(42).rand(rank, m).T.copy()).to(device)
    engine.V.zero_()
    engine.V[:, :rank] = init.to(engine.V.dtype)
    engine.V[engine.ilo:engine.ihi, 0] = avg.to(engine.V.dtype)
    engine._all_gather_rows(engine.V, engine.item_bounds)
    engine.U.zero_()


def timed_sweeps(engine, steps, warmup, fused, barrier, cuda):
    """`warmup` untimed and `steps` timed sweeps; returns (elapsed seconds between the two
    barriers, per-side mean seconds, seconds of the closing error pass or None, rmse list)."""
    sse_log = []

    def step(stamps=None):
        if stamps:
            stamps[0].record()
        if fused:
            sse_log.append(engine.half_sweep_users_err())  # error of the state before
        else:
            engine.half_sweep_users()
        if stamps:
            stamps[1].record()
        engine.half_sweep_items()
        if stamps:
            stamps[2].record()
        if not fused:
            sse_log.append(engine.train_sse_async())

    for _ in range(warmup):
        step()
    engine.train_sse_async()  # untimed: first-use set-up of the standalone error pass
    st = [[Stamp(cuda) for _ in range(3)] for _ in range(steps)]
    tail = [Stamp(cuda), Stamp(cuda)]
    barrier()
    t0 = time.perf_counter()
    for s in range(steps):
        step(st[s])
    tail[0].record()
    if fused:
        sse_log.append(engine.train_sse_async())  # error of the last timed sweep
    tail[1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    t_user = float(np.mean([st[s][0].ms_until(st[s][1]) for s in range(steps)])) * 1e-3
    t_item = float(np.mean([st[s][1].ms_until(st[s][2]) for s in range(steps)])) * 1e-3
    t_sse = tail[0].ms_until(tail[1]) * 1e-3 if fused else None
    rmse = [float(np.sqrt(float(x.cpu()[0]) / engine.nnz)) for x in sse_log]
    return elapsed, t_user, t_item, t_sse, rmse


def side_report(sec, algo_bytes, flops, dtype, counters, side):
    """Per-side roofline entry.  Two roofs from the algorithmic model: the bytes at the HBM
    peak and the arithmetic at the matrix pipes' dense peaks; `bound` names the larger one
    and `frac` is that roof's time over the measured time.  `algorithmic_GBps` is the
    uncached-model rate (every stored entry fetches one factor row) and can exceed what HBM
    delivers when the gathered table is cache resident; the measured HBM bytes of the committed
    counter passes (when they belong to this build) stand beside it."""
    gram, solve, products = flops
    hbm_s = algo_bytes / (HBM_PEAK_GBS * 1e9)
    mat_s = matrix_roof_seconds(gram, solve, products, dtype)
    out = {'ms': sec * 1e3, 'algorithmic_bytes': algo_bytes, 'algorithmic_GBps': algo_bytes / sec / 1e9,
           'gram_flops': gram, 'solve_flops': solve, 'matrix_products_per_gram_product': products,
           'hbm_roof_ms': hbm_s * 1e3, 'matrix_roof_ms': mat_s * 1e3,
           'bound': 'hbm' if hbm_s >= mat_s else 'mfma', 'frac': max(hbm_s, mat_s) / sec}
    c = (counters or {}).get(side)
    if c:
        out['hbm_measured_bytes'] = c['hbm_bytes']
        out['hbm_measured_GBps'] = c['hbm_bytes'] / sec / 1e9
        out['hbm_measured_frac_of_peak'] = c['hbm_bytes'] / sec / 1e9 / HBM_PEAK_GBS
        out['mfma_util'] = c.get('mfma_util')
        # traffic well below the algorithmic bytes: the gathered table is served from L2 /
        # Infinity Cache and neither roof binds; what does is instruction issue
        out['limited_by'] = 'hbm' if c['hbm_bytes'] >= 0.5 * algo_bytes else \
            'instruction issue / cache gather (the gathered table is cache resident)'
    return out


def roofline_report(engine, dtype, t_user, t_item, fused, counters):
    """The `roofline` object of the JSON line for one engine's measured half-sweep times."""
    s = np.dtype(dtype).itemsize
    kp = engine.kp
    b_user = half_sweep_bytes(engine.csr.nnz, engine.csr.n_rows, kp, s)
    if fused:
        b_user += engine.csr.n_rows * kp * s  # the replaced vectors are read for the error
    b_item = half_sweep_bytes(engine.csc.nnz, engine.csc.n_rows, kp, s)
    f_user = side_flops(engine.csr.indptr_host, engine.rank, kp, dtype)
    f_item = side_flops(engine.csc.indptr_host, engine.rank, kp, dtype)
    user = side_report(t_user, b_user, f_user, dtype, counters, 'user_side')
    item = side_report(t_item, b_item, f_item, dtype, counters, 'item_side')
    t = t_user + t_item
    hbm_s = (b_user + b_item) / (HBM_PEAK_GBS * 1e9)
    mat_s = (user['matrix_roof_ms'] + item['matrix_roof_ms']) * 1e-3
    traffic = None
    if counters and 'user_side' in counters:
        traffic = (counters['user_side']['hbm_bytes'] + counters['item_side']['hbm_bytes']) / 2
    out = {'traffic': traffic,
           'hbm_roof_ms_per_sweep': hbm_s * 1e3, 'matrix_roof_ms_per_sweep': mat_s * 1e3,
           'algorithmic_GBps': (b_user + b_item) / t / 1e9}
    if hbm_s >= mat_s:
        achieved = (b_user + b_item) / t / 1e9
        out.update(bound='hbm', achieved=achieved, peak=HBM_PEAK_GBS, unit='GB/s',
                   frac=achieved / HBM_PEAK_GBS)
    else:
        # credited arithmetic = what the model prices (upper Gram tiles + solves); the peak is
        # the rate at which the matrix pipes would finish exactly that work
        work = sum(f[0] + f[1] for f in (f_user, f_item))
        achieved, peak = work / t / 1e12, work / mat_s / 1e12
        out.update(bound='mfma', achieved=achieved, peak=peak, unit='TFLOP/s', frac=achieved / peak)
    out.update({
        'note': 'two roofs from the algorithmic model: bytes (uncached: every stored entry fetches '
                'one factor row; per half-sweep nnz*(ld*s+4+s) + rows*(ld*s+8)) at the 8 TB/s HBM '
                'peak, and the issued arithmetic at the dense matrix-pipe peaks (2.5 PFLOP/s 16-bit, '
                '157.3 TFLOP/s fp32, 78.6 TFLOP/s fp64); bound = the larger, frac = its time / the '
                'HIP-event time.  Where the measured HBM traffic (counters) is far below the '
                'algorithmic bytes the gathered table is cache resident and what limits the side is '
                'instruction issue and the cache-to-CU gather rate (see limited_by per side)',
        'kernel': 'half-sweep launch = max_abs + gram_chunks + sum_slabs + solve_rows '
                  '(+ solve_rows_dual for short rows at padded rank 64 and 128), mean of user side and item side',
        'bytes_per_launch': (b_user + b_item) / 2.0,
        'ms_per_launch': t / 2.0 * 1e3,
        'user_side': user, 'item_side': item,
        'counters': ({k: v for k, v in counters.items()
                      if k not in ('user_side', 'item_side')} if counters else None),
    })
    return out


def rmse_gap(a, b):
    """Largest relative difference between two per-sweep RMSE histories over their common
    sweeps (both start from the same factors; entry j = the error after sweep j)."""
    n = min(len(a), len(b))
    pairs = [(x, y) for x, y in zip(a[:n], b[:n]) if x is not None and y is not None and y > 0]
    return (max(abs(x - y) / y for x, y in pairs) if pairs else None), n


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
    ap.add_argument('--no-f64', action='store_true',
                    help='skip the float64 companion measurement of the default run')
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
    # path on a one-GPU box; RCCL wants one device per rank).  ALSMI_BENCH_BACKEND=module:Class
    # replaces the engine's device backend by another implementation on host tensors (the test
    # suite rehearses this script's multi-rank logic that way on machines without a GPU).
    dist_backend = os.environ.get('ALSMI_DIST_BACKEND', 'nccl')
    plug = os.environ.get('ALSMI_BENCH_BACKEND')
    cuda = plug is None
    engine_backend = None
    if plug:
        mod, cls = plug.split(':')
        engine_backend = getattr(importlib.import_module(mod), cls)()
        device = torch.device('cpu')
        dist_backend = 'gloo'
    else:
        if dist_backend != 'nccl':
            local = local % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
        device = torch.device('cuda', local)
    group = None
    # ALSMI_FORCE_COLLECTIVES=1 with one rank: a one-rank group whose gathers and all-reduces
    # are really issued (the multi-rank code path on RCCL with the one GPU at hand)
    forced = world == 1 and os.environ.get('ALSMI_FORCE_COLLECTIVES', '0') not in ('', '0')
    if forced:
        os.environ.setdefault('MASTER_PORT', '29577')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
    if world > 1 or forced:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if dist_backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(dist_backend)
        group = dist.group.WORLD

    from alsmi import synth
    from alsmi.engine import AlsEngine
    cfg = CONFIGS[args.config]
    if cfg['blocks'] and cfg['blocks'] % world:
        raise SystemExit('--config %s needs a rank count that divides %d'
                         % (args.config, cfg['blocks']))
    t_gen = time.perf_counter()
    prob = synth.sharded_problem(20242, cfg['n'], cfg['m'], cfg['nnz'], world, rank_id, device,
                                 n_blocks=cfg['blocks'])
    if cuda:
        torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen

    def make_engine(dtype):
        eng = AlsEngine(prob['csr'], prob['csc'], prob['shape'], cfg['rank'], 0.1, dtype=dtype,
                        user_bounds=prob['user_bounds'], item_bounds=prob['item_bounds'],
                        group=group, chunk=args.chunk, total_nnz=prob['total_nnz'],
                        backend=engine_backend)
        init_factors(eng, cfg['rank'], device)
        return eng

    def barrier():
        if group is not None:
            import torch.distributed as dist
            dist.barrier()
        if cuda:
            torch.cuda.synchronize()

    fused = not args.separate_error
    engine = make_engine(args.dtype)
    engine.time_gathers = engine.collective
    elapsed, t_user, t_item, t_sse, rmse = timed_sweeps(engine, args.steps, args.warmup, fused,
                                                        barrier, cuda)
    # time the compute stream spent waiting for row gathers, per half-sweep (this rank; the
    # warm-up sweeps are in the sum as well, hence the division by all sweeps run)
    exposed = engine.gather_exposed_ms() / (2.0 * (args.steps + args.warmup)) \
        if engine.collective else None
    if group is not None:
        import torch.distributed as dist
        t = torch.tensor([elapsed, exposed or 0.0], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.cpu()[0])
        exposed = float(t.cpu()[1])

    # float64 companion (the reference's arithmetic, the 1e-10 parity mode) for the default
    # single-GPU run: same problem, same schedule, a few sweeps after the timed region
    f64 = None
    if world == 1 and args.dtype == 'float32' and args.config == 'c2' and not args.no_f64 and cuda:
        eng64 = make_engine('float64')
        e64, u64, i64, _, r64 = timed_sweeps(eng64, 5, 2, fused, barrier, cuda)
        roof64 = roofline_report(eng64, 'float64', u64, i64, fused,
                                 measured_counters(args.config, 'float64'))
        gap, n_cmp = rmse_gap(rmse, r64)
        f64 = {'dtype': 'f64 (v_mfma_f64_16x16x4 Gram, block LDL^T solve on the accumulator tiles)',
               'ms_per_step': e64 / 5 * 1e3, 'sweeps_per_sec': 5 / e64,
               'value': eng64.nnz * 5 / e64, 'unit': 'nnz/s',
               'user_ms': u64 * 1e3, 'item_ms': i64 * 1e3,
               'roofline_frac': roof64['frac'], 'roofline_bound': roof64['bound'],
               'roofline': {k: roof64[k] for k in ('bound', 'achieved', 'peak', 'unit', 'frac',
                                                   'traffic', 'hbm_roof_ms_per_sweep',
                                                   'matrix_roof_ms_per_sweep')},
               'train_rmse': r64,
               'rmse_fp32_vs_fp64_max_rel': gap, 'rmse_sweeps_compared': n_cmp}
        del eng64

    if rank_id == 0:
        s = 4 if args.dtype == 'float32' else 8
        kp = engine.kp
        b_sse = sddmm_bytes(engine.csr.nnz, engine.csr.n_rows, kp, s)
        counters = measured_counters(args.config, args.dtype) if world == 1 else None
        # a stand-in backend (ALSMI_BENCH_BACKEND, CPU rehearsals of the multi-rank logic) times no
        # kernel: its line carries no roofline
        roofline = roofline_report(engine, args.dtype, t_user, t_item, fused, counters) \
            if plug is None else None
        sweeps_per_sec = args.steps / elapsed
        strong = bool(cfg['blocks'])
        line = {
            'metric': ('ALS nnz/sec (stored entries x sweeps/sec), ML-20M shape rank %d'
                       if args.config in ('c2', 'r128') else
                       'ALS nnz/sec (stored entries x sweeps/sec), rank %d') % cfg['rank'],
            'value': engine.nnz * sweeps_per_sec,
            'unit': 'nnz/s',
            'sweeps_per_sec': sweeps_per_sec,
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'strong' if strong else 'weak',
            'vs_baseline': None,
            'dtype': ('f32 (products on the fp16 matrix cores from a scaled two-way fp16 split '
                      'of the fp32 factors, fp32 accumulate and solve; fp32-level accuracy)'
                      if args.dtype == 'float32' else 'f64'),
            'data': 'synthetic',
            'config': {'workload': cfg['workload'], 'users': engine.n, 'items': engine.m,
                       'nnz': engine.nnz, 'rank': cfg['rank'], 'alpha': 0.1,
                       'parallelism': 'row-sharded x%d' % world},
            'roofline': roofline,
            'error_pass': ({'schedule': 'fused into the next user half-sweep; one standalone '
                                        'pass after the last timed sweep',
                            'standalone_ms': t_sse * 1e3,
                            'standalone_GBps': b_sse / t_sse / 1e9} if fused else
                           {'schedule': 'standalone pass after every sweep'}),
            'train_rmse': {'first': rmse[0] if rmse else None, 'last': rmse[-1] if rmse else None,
                           'per_sweep': rmse,
                           'note': 'entry j = training RMSE after j sweeps (entry 0: the initial '
                                   'factors), each taken from the next user half-sweep\'s normal '
                                   'equations, the last one from the standalone pass'},
            'setup_s': {'generate': t_gen},
            'backend': 'hip' if plug is None else 'stand-in ' + plug,
        }
        if exposed is not None:
            line['allgather_exposed_ms'] = exposed
            line['allgather'] = ('all-gather-v of the updated rows as one broadcast per owning '
                                 'rank, straight into the factor table, %d user / %d item slabs '
                                 'per rank in flight beside the next slab\'s kernels; exposed = '
                                 'mean time per half-sweep the compute stream waited for them '
                                 '(max over ranks)' % (len(engine.plans_u), len(engine.plans_i)))
        if f64:
            line['f64'] = f64
            line['rmse_fp32_vs_fp64_max_rel'] = f64['rmse_fp32_vs_fp64_max_rel']
        if world == 1 and args.cpu_frac > 0 and cuda:
            threads = int(os.environ.get('OMP_NUM_THREADS', '0')) or min(16, os.cpu_count() or 1)
            try:
                line['cpu_baseline'] = cpu_baseline(engine, args.cpu_frac, threads)
            except Exception as exc:  # the checker is optional for the number itself
                line['cpu_baseline'] = {'value': None, 'unit': 'nnz/s', 'cores': threads,
                                        'kind': 'port', 'sample': 'failed: %r' % (exc,)}
        print(json.dumps(line))
    if group is not None:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
