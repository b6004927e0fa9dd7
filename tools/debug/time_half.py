#!/usr/bin/env python3
"""Times the two half-sweeps on FIXED, valid input factors (restored before every call), so
that timing-only builds whose results are wrong can be compared with the production library:
    ALSMI_LIB=build/libalsmi_x.so python tools/debug/time_half.py [config] [dtype]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'recommendation-models_amd')):
    sys.path.insert(0, p)
import numpy as np
import torch
import bench
from alsmi import synth, device as dev
from alsmi.engine import AlsEngine

cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'c2']
dtype = sys.argv[2] if len(sys.argv) > 2 else 'float32'
device = dev.require_gpu()
prob = synth.sharded_problem(20243, cfg['n'], cfg['m'], cfg['nnz'], 1, 0, device)
eng = AlsEngine(prob['csr'], prob['csc'], (cfg['n'], cfg['m']), cfg['rank'], 0.1,
                dtype=np.dtype(dtype), total_nnz=int(prob['csr'][0][-1]))
bench.init_factors(eng, cfg['rank'], device)
g = torch.Generator(device=device)
g.manual_seed(1)
V0 = eng.V.clone()
U0 = torch.zeros_like(eng.U)
U0[:, :cfg['rank']] = torch.randn((eng.n, cfg['rank']), generator=g, device=device).to(U0.dtype) * 0.3
def timed(fn, restore, n=10):
    ts = []
    for _ in range(n + 2):
        restore()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts[2:]))
def restore():
    eng.U.copy_(U0); eng.V.copy_(V0)
print('user half (plain) %.4f ms' % timed(eng.half_sweep_users, restore))
print('user half (error) %.4f ms' % timed(lambda: eng.half_sweep_users_err(), restore))
print('item half         %.4f ms' % timed(eng.half_sweep_items, restore))
