#!/usr/bin/env python3
"""Run the same half-sweep several times and report whether the outputs are bit-identical.

    python tools/determinism.py [float32|float64 [rank ...]]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'recommendation-models_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
from alsmi import device as dev, synth  # noqa: E402

device = dev.require_gpu()
dtype = sys.argv[1] if len(sys.argv) > 1 else 'float32'
td = torch.float64 if dtype == 'float64' else torch.float32
ranks = [int(a) for a in sys.argv[2:]] or [64, 10, 32, 128]
for (n, m, nnz, rank) in [(138493, 26744, 20000000, r) for r in ranks]:
    prob = synth.sharded_problem(20242, n, m, nnz, 1, 0, device)
    for name, tri, ncols in (('user', prob['csr'], m), ('item', prob['csc'], n)):
        side = dev.SparseSide(tri[0], tri[1], tri[2], len(tri[0]) - 1, ncols, dtype, device)
        plan = dev.SweepPlan(side, rank)
        kp = dev.padded_rank(rank)
        g = torch.Generator(device=device); g.manual_seed(1)
        src = torch.zeros((ncols, kp), device=device, dtype=td)
        src[:, :rank] = torch.rand((ncols, rank), generator=g, device=device, dtype=td)
        outs = []
        for rep in range(8):
            dst = torch.zeros((side.n_rows, kp), device=device, dtype=td)
            plan.half_sweep(src, dst, 0.1)
            outs.append(dst)
        cur = outs[0].clone()
        errs = []
        for rep in range(3):
            dst = torch.zeros((side.n_rows, kp), device=device, dtype=td)
            errs.append(float(plan.half_sweep_err(src, cur, dst, 0.1).cpu()[0]))
            outs.append(dst)
        same = [bool(torch.equal(outs[0], o)) for o in outs[1:8]]
        same_err = [bool(torch.equal(outs[8], o)) for o in outs[9:]]
        diff = max(float((outs[0] - o).abs().max()) for o in outs[1:8])
        bad_rows = int(((outs[0] != outs[1]).any(dim=1)).sum())
        ptr = side.indptr_host
        for o in outs[1:8]:
            rows = torch.nonzero((outs[0] != o).any(dim=1)).flatten().cpu().numpy()
            if rows.size:
                print('   ref', outs[0][rows[0], :6].cpu().numpy(), 'other', o[rows[0], :6].cpu().numpy(), 'n cols differing', int((outs[0][rows[0]] != o[rows[0]]).sum()))
                print('   differing rows', rows[:8], 'degrees', [int(ptr[r + 1] - ptr[r]) for r in rows[:8]], 'info', plan.info())
        print('%s n=%d rank=%d: plain identical %s (max diff %.3g, rows differing %d); err-variant identical %s, sse %s'
              % (name, side.n_rows, rank, same, diff, bad_rows, same_err, errs))
        plan.close()
