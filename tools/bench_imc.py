#!/usr/bin/env python3
"""IMC evaluation throughput at BASELINE config 3 shape (138 493 x 26 744, 20 M entries,
users x 128 and items x 256 side features): one cost+gradient evaluation and one
Hessian-vector product per step, fp64.  Prints one JSON line.  Not the headline bench (that is
bench.py); gives the per-kernel IMC numbers.

    python tools/bench_imc.py [--small] [--steps 5] [--cpu-frac 0.02]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, 'recommendation-models_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--small', action='store_true')
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--cpu-frac', type=float, default=0.02)
    ap.add_argument('--dtype', default='float64', choices=['float32', 'float64'])
    ap.add_argument('--low-rank', type=int, default=0,
                    help='also time the rank-constrained evaluation (Z = U V^T) of this rank')
    args = ap.parse_args()
    from alsmi import _lib, synth
    from alsmi import device as dev
    device = dev.require_gpu()
    n, m, nnz, p, q = (20000, 5000, 1000000, 32, 64) if args.small else \
        (138493, 26744, 20000000, 128, 256)
    prob = synth.sharded_problem(20243, n, m, nnz, 1, 0, device)
    indptr, items, vals = prob['csr']
    g = torch.Generator(device=device)
    g.manual_seed(3)
    td = torch.float64 if args.dtype == 'float64' else torch.float32
    code = _lib.F64 if args.dtype == 'float64' else _lib.F32
    X = (torch.randn((n, p), generator=g, device=device, dtype=torch.float64) / np.sqrt(p)).to(td)
    ld = _lib.load().alsmi_imc_padded_width(q)
    Y = torch.zeros((m, ld), device=device, dtype=td)
    Y[:, :q] = (torch.randn((m, q), generator=g, device=device, dtype=torch.float64)
                / np.sqrt(q)).to(td)
    Z = (torch.randn((p, q), generator=g, device=device, dtype=torch.float64) * 0.1).to(td)
    ratings = vals.to(td)
    out = torch.empty((p, q), device=device, dtype=td)
    sse = torch.zeros(1, device=device, dtype=torch.float64)
    nbytes = int(_lib.load().alsmi_imc_workspace_bytes(code, n, nnz, p, q))
    ws = torch.empty(nbytes // 8 + 1, device=device, dtype=torch.float64)

    def evaluate(mode):
        _lib.check(_lib.load().alsmi_imc_eval(
            code, mode, n, nnz, p, q, dev._ptr(indptr), dev._ptr(items), dev._ptr(ratings),
            dev._ptr(X), dev._ptr(Y), dev._ptr(Z), dev._ptr(out), None,
            dev._ptr(sse) if mode == 0 else None, dev._ptr(ws), dev._stream()), 'alsmi_imc_eval')

    for mode in (0, 1):
        evaluate(mode)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t_grad = t_hess = 0.0
    for _ in range(args.steps):
        ev[0].record()
        evaluate(0)
        ev[1].record()
        evaluate(1)
        ev[2].record()
        torch.cuda.synchronize()
        t_grad += ev[0].elapsed_time(ev[1]) * 1e-3
        t_hess += ev[1].elapsed_time(ev[2]) * 1e-3
    t_grad /= args.steps
    t_hess /= args.steps
    s = 8 if args.dtype == 'float64' else 4
    algo = nnz * (ld * s + 4 + s) + 2 * n * ld * s   # per evaluation (SDDMM+SpMM pass)
    flops = 4.0 * n * p * q + 4.0 * nnz * q           # two GEMMs + dot and axpy per entry
    line = {
        'metric': 'IMC evaluations/sec (cost+gradient, Hessian-vector), config 3 shape',
        'config': {'workload': 'IMC n=%d m=%d nnz=%d p=%d q=%d %s' % (n, m, nnz, p, q, args.dtype)},
        'dtype': 'f64' if args.dtype == 'float64' else 'f32', 'data': 'synthetic',
        'cost_grad': {'ms': t_grad * 1e3, 'evals_per_sec': 1 / t_grad,
                      'nnz_per_sec': nnz / t_grad, 'GBps': algo / t_grad / 1e9,
                      'frac_of_8TBps': algo / t_grad / 8e12, 'TFLOPs': flops / t_grad / 1e12},
        'hessp': {'ms': t_hess * 1e3, 'evals_per_sec': 1 / t_hess, 'GBps': algo / t_hess / 1e9,
                  'frac_of_8TBps': algo / t_hess / 8e12},
        'algorithmic_bytes_per_eval': algo,
    }
    if args.low_rank:
        # cost + both gradients of the rank-constrained form: two passes over the samples that
        # are `low_rank` wide (user-major against Y V, item-major against X U), host
        # round trip of the p x r and q x r results included
        from alsmi.imc import ImcProblem
        r = args.low_rank
        prob2 = ImcProblem.from_device_csr(indptr, items, ratings, X, Y[:, :q], dtype=args.dtype)
        rs = np.random.RandomState(0)
        U0, V0 = rs.randn(p, r) * 0.1, rs.randn(q, r) * 0.1
        prob2.low_rank_eval(U0, V0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            prob2.low_rank_eval(U0, V0)
        dt = (time.perf_counter() - t0) / args.steps
        ldr = _lib.load().alsmi_imc_padded_width(r)
        algo_lr = 2 * nnz * (ldr * s + 4 + s) + 2 * (n + m) * ldr * s
        line['low_rank'] = {'rank': r, 'ms': dt * 1e3, 'evals_per_sec': 1 / dt,
                            'nnz_per_sec': nnz / dt, 'algorithmic_bytes_per_eval': algo_lr,
                            'GBps': algo_lr / dt / 1e9, 'frac_of_8TBps': algo_lr / dt / 8e12,
                            'what': 'cost + dU + dV of Z = U V^T'}
    if args.cpu_frac > 0:
        from oracle import c_oracle as co
        rows = max(1, int(n * args.cpu_frac))
        ip = indptr[:rows + 1].cpu().numpy()
        cnt = int(ip[-1])
        threads = int(os.environ.get('OMP_NUM_THREADS', '0')) or min(16, os.cpu_count() or 1)
        t0 = time.perf_counter()
        co.imc_eval(0, ip, items[:cnt].cpu().numpy(), ratings[:cnt].double().cpu().numpy(),
                    X[:rows].double().cpu().numpy(), Y[:, :q].double().cpu().numpy(),
                    Z.double().cpu().numpy(), threads=threads)
        dt = (time.perf_counter() - t0) * nnz / max(cnt, 1)
        line['cpu_baseline'] = {'value': 1 / dt, 'unit': 'cost+gradient evaluations/s',
                                'cores': threads, 'kind': 'port',
                                'sample': 'first %d users (%d entries), extrapolated' % (rows, cnt)}
    print(json.dumps(line))


if __name__ == '__main__':
    main()
