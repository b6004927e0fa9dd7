#!/usr/bin/env python3
"""Per-item rating sums (the preamble's item means, src/als.py:143-147) three ways on the device:
index_add_ (atomics), CSR->CSC + torch.segment_reduce, CSR->CSC + cumulative-sum differences.
Times and the differences against a host bincount."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'recommendation-models_amd')):
    sys.path.insert(0, p)
import numpy as np
import torch
from alsmi import device as dev, synth

device = dev.require_gpu()
for (n, m, nnz) in ((138493, 26744, 20000000), (250000, 200000, 12500000)):
    prob = synth.sharded_problem(20242, n, m, nnz, 1, 0, device)
    indptr, indices, vals = prob['csr']
    vals = vals.to(torch.float64) * 1.37
    cols = indices.to(torch.int64)
    want = np.bincount(indices.cpu().numpy(), weights=vals.cpu().numpy(), minlength=m)

    def t_index_add():
        return torch.zeros(m, dtype=torch.float64, device=device).index_add_(0, cols, vals)

    def t_segment():
        cptr, cidx, cvals = dev.csr_to_csc(indptr, indices, vals, (n, m))
        return torch.segment_reduce(cvals, 'sum', offsets=cptr, initial=0.0)

    def t_cumsum():
        cptr, cidx, cvals = dev.csr_to_csc(indptr, indices, vals, (n, m))
        run = torch.zeros(cvals.numel() + 1, dtype=torch.float64, device=device)
        torch.cumsum(cvals, 0, out=run[1:])
        return run[cptr[1:]] - run[cptr[:-1]]

    for name, fn in (('index_add_', t_index_add), ('csc + segment_reduce', t_segment),
                     ('csc + cumsum', t_cumsum)):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            out = fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        got = out.cpu().numpy()
        rel = np.abs(got - want).max() / np.abs(want).max()
        again = fn().cpu().numpy()
        print('%d x %d, %d entries  %-22s %8.2f ms   max diff %.2e of the largest sum   repeatable %s'
              % (n, m, nnz, name, dt * 1e3, rel, np.array_equal(got, again)), flush=True)
