#!/usr/bin/env python3
"""gpurun_out/<tag>/ (written by tools/profile_round.sh) -> the committed profiles/ files:
bench line, kernel stats, the PMC CSVs restricted to the library's kernels, and traffic.json
(HBM bytes per half-sweep launch: FETCH_SIZE in KB, doubled as gfx950 requires, + WRITE_SIZE KB).
usage: tools/profile_collect.py <tag>"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, 'gpurun_out', tag)
dst = os.path.join(root, 'profiles')
shutil.copy(os.path.join(src, 'bench_c2.json'), os.path.join(dst, '%s_bench_c2.json' % tag))
shutil.copy(os.path.join(src, 'stats', 'c2_kernel_stats.csv'),
            os.path.join(dst, '%s_bench_c2_kernel_stats.csv' % tag))


def rows_of(name):
    path = glob.glob(os.path.join(src, name, '*counter_collection.csv'))[0]
    return [r for r in csv.DictReader(open(path)) if 'alsmi::' in r['Kernel_Name']]


for name, out in (('pmc_fetch_size', 'fetch_size'), ('pmc_write_size', 'write_size'),
                  ('pmc_sq_waves', 'sq_issue'), ('pmc_sq_wave_cycles', 'sq_wait')):
    rows = rows_of(name)
    with open(os.path.join(dst, '%s_bench_c2_pmc_%s.csv' % (tag, out)), 'w', newline='') as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)


def per_launch(name, counter):
    """Sum of `counter` over the kernels of one half-sweep, split by side: a half-sweep is the
    run of dispatches that ends with a solve_rows kernel; the user side is the ERR variant."""
    sides = defaultdict(list)
    run = 0.0
    rows = [r for r in rows_of(name) if r['Counter_Name'] == counter]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    for r in rows:
        k = r['Kernel_Name']
        if not any(s in k for s in ('gram_chunks', 'sum_slabs', 'solve_rows', 'max_abs', 'sum_err')):
            continue
        run += float(r['Counter_Value'])
        if 'solve_rows' in k:
            side = 'user_side' if ', true>' in k.split('(')[0] else 'item_side'
            sides[side].append(run)
            run = 0.0
        if 'sum_err' in k and sides['user_side']:
            sides['user_side'][-1] += float(r['Counter_Value'])
            run = 0.0
    return {s: sum(v) / len(v) for s, v in sides.items()}


fetch = per_launch('pmc_fetch_size', 'FETCH_SIZE')
write = per_launch('pmc_write_size', 'WRITE_SIZE')
traffic = {
    'source': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, bench.py '
              '--steps 3 --warmup 1 --cpu-frac 0 (config c2, f32); tools/profile_round.sh',
    'units': 'counters are KB; FETCH_SIZE doubled (gfx950 reports half the bytes of wide '
             'coalesced reads), WRITE_SIZE as is; per half-sweep launch = max_abs + gram_chunks '
             '+ sum_slabs + solve_rows (+ sum_err on the user side)',
}
total = 0.0
for side in ('user_side', 'item_side'):
    f, w = fetch[side] * 1024.0, write[side] * 1024.0
    traffic[side] = {'fetch_size_raw_bytes': f, 'write_size_bytes': w, 'hbm_bytes': 2 * f + w}
    total += 2 * f + w
traffic['half_sweep_hbm_bytes_per_launch'] = total / 2
json.dump(traffic, open(os.path.join(dst, 'traffic.json'), 'w'), indent=1)
print(json.dumps(traffic, indent=1))
