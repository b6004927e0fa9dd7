#!/usr/bin/env python3
"""gpurun_out/<tag>_<config>_<dtype>/ (written by tools/profile_round.sh) -> the committed
profiles/ files: bench line, kernel stats, the PMC CSVs restricted to the library's kernels,
and profiles/<tag>_summary.json, the one-screen digest bench.py reads back:
per configuration and per side of the sweep the HBM bytes of one half-sweep launch (FETCH_SIZE
in KB, doubled as gfx950 requires for wide coalesced reads, + WRITE_SIZE KB), the MFMA
utilisation (SQ_VALU_MFMA_BUSY_CYCLES over SIMD cycles) and the kernels' average times, stamped
with the commit and a hash of the kernel sources they were measured on.
usage: tools/profile_collect.py <tag> <config>[/<dtype>] [<config>[/<dtype>] ...]"""
import csv
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys
from collections import defaultdict

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, 'profiles')
HALF = ('max_abs', 'gram_chunks', 'sum_slabs', 'solve_rows', 'sum_err')
N_SIMD = 1024  # 256 CUs x 4


def source_sha():
    """The same hash as bench.py's kernel_source_sha: the half-sweep kernels' sources."""
    h = hashlib.sha256()
    for name in ('als_sweep.hip', 'alsmi_common.h', 'alsmi_common.hip', 'Makefile'):
        h.update(open(os.path.join(root, 'recommendation-models_amd', 'csrc', name), 'rb').read())
    return h.hexdigest()[:16]


def rows_of(src, name):
    path = glob.glob(os.path.join(src, name, '**', '*counter_collection.csv'), recursive=True)[0]
    return [r for r in csv.DictReader(open(path)) if 'alsmi::' in r['Kernel_Name']]


def launches(rows, counter):
    """Per side the list of per-launch sums of `counter`: a half-sweep launch is the run of
    dispatches that starts with max_abs_kernel (fp32) or with the first chunk / solve kernel
    after a sum_err / solve kernel of another kind; the user side is the one whose kernels are
    the error-summing variants."""
    rows = [r for r in rows if r['Counter_Name'] == counter]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    sides = defaultdict(list)
    cur, is_user, seen_solve = 0.0, False, False

    def close():
        nonlocal cur, is_user, seen_solve
        if seen_solve:
            sides['user_side' if is_user else 'item_side'].append(cur)
        cur, is_user, seen_solve = 0.0, False, False

    for r in rows:
        k = r['Kernel_Name']
        if not any(s in k for s in HALF):
            continue
        head = k.split('(')[0]
        starts = 'max_abs' in k or (seen_solve and ('gram_chunks' in k or
                                                    ('solve_rows_kernel' in k)))
        if starts:
            close()
        cur += float(r['Counter_Value'])
        if 'solve_rows' in k:
            seen_solve = True
            if head.rstrip().endswith('true>'):
                is_user = True
        if 'sum_err' in k:
            is_user = True
    close()
    return {s: sum(v) / len(v) for s, v in sides.items() if v}


def collect(spec):
    config, _, dtype = spec.partition('/')
    dtype = dtype or 'float32'
    src = os.path.join(root, 'gpurun_out', '%s_%s_%s' % (tag, config, dtype))
    stem = '%s_bench_%s%s' % (tag, config, '' if dtype == 'float32' else '_f64')
    shutil.copy(os.path.join(src, 'bench.json'), os.path.join(dst, stem + '.json'))
    stats = glob.glob(os.path.join(src, 'stats', '**', '*kernel_stats.csv'), recursive=True)[0]
    shutil.copy(stats, os.path.join(dst, stem + '_kernel_stats.csv'))
    for name, out in (('pmc_fetch_size', 'fetch_size'), ('pmc_write_size', 'write_size'),
                      ('pmc_sq_waves', 'sq_issue'), ('pmc_sq_wave_cycles', 'sq_wait')):
        rows = rows_of(src, name)
        with open(os.path.join(dst, '%s_pmc_%s.csv' % (stem, out)), 'w', newline='') as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(rows)
    fetch = launches(rows_of(src, 'pmc_fetch_size'), 'FETCH_SIZE')
    write = launches(rows_of(src, 'pmc_write_size'), 'WRITE_SIZE')
    issue = rows_of(src, 'pmc_sq_waves')
    busy = launches(issue, 'SQ_VALU_MFMA_BUSY_CYCLES')
    gui = launches(issue, 'GRBM_GUI_ACTIVE')
    entry = {}
    for side in ('user_side', 'item_side'):
        f, w = fetch[side] * 1024.0, write[side] * 1024.0
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        cycles = gui[side] / 8.0
        entry[side] = {'fetch_size_raw_bytes': f, 'write_size_bytes': w, 'hbm_bytes': 2 * f + w,
                       'mfma_busy_cycles': busy[side], 'gpu_cycles': cycles,
                       'mfma_util': busy[side] / (cycles * N_SIMD)}
    kern = {}
    for r in csv.DictReader(open(stats)):
        if 'alsmi::' in r['Name']:
            kern[r['Name'].split('(')[0].replace('void ', '')] = {
                'calls': int(r['Calls']), 'avg_us': float(r['AverageNs']) / 1e3}
    entry['kernels'] = kern
    line = json.loads(open(os.path.join(src, 'bench.json')).read().strip().splitlines()[-1])
    entry['bench_ms_per_step'] = line['ms_per_step']
    # The bench line of this very call was printed before its counters existed: stamp them in,
    # the way bench.py does when it finds a summary measured on the running sources.
    roof = line['roofline']
    for side in ('user_side', 'item_side'):
        r, c = roof[side], entry[side]
        r['hbm_measured_bytes'] = c['hbm_bytes']
        r['hbm_measured_GBps'] = c['hbm_bytes'] / (r['ms'] * 1e-3) / 1e9
        r['hbm_measured_frac_of_peak'] = r['hbm_measured_GBps'] / 8000.0
        r['mfma_util'] = c['mfma_util']
        r['limited_by'] = 'hbm' if c['hbm_bytes'] >= 0.5 * r['algorithmic_bytes'] else \
            'instruction issue / cache gather (the gathered table is cache resident)'
    roof['traffic'] = (entry['user_side']['hbm_bytes'] + entry['item_side']['hbm_bytes']) / 2
    roof['counters'] = {'kernels': kern, 'file': 'profiles/%s_summary.json' % tag,
                        'measured': 'same gpurun call as this line (tools/profile_round.sh)'}
    with open(os.path.join(dst, stem + '.json'), 'w') as f:
        f.write(json.dumps(line) + '\n')
    return '%s/%s' % (config, dtype), entry


def collect_imc(dtype):
    """gpurun_out/<tag>_imc_c3_<dtype>/ (tools/profile_imc.sh) -> profiles/<tag>_bench_imc_c3[_f32]*
    and the summary entry: per kernel of an evaluation the mean HBM bytes per launch (2 x
    FETCH_SIZE + WRITE_SIZE), its average time and matrix-pipe utilisation."""
    src = os.path.join(root, 'gpurun_out', '%s_imc_c3_%s' % (tag, dtype))
    stem = '%s_bench_imc_c3%s' % (tag, '' if dtype == 'float64' else '_f32')
    stats = glob.glob(os.path.join(src, 'stats', '**', '*kernel_stats.csv'), recursive=True)[0]
    shutil.copy(stats, os.path.join(dst, stem + '_kernel_stats.csv'))
    per = defaultdict(lambda: defaultdict(list))
    for name, out in (('pmc_fetch_size', 'fetch_size'), ('pmc_write_size', 'write_size'),
                      ('pmc_sq_waves', 'sq_issue')):
        rows = rows_of(src, name)
        with open(os.path.join(dst, '%s_pmc_%s.csv' % (stem, out)), 'w', newline='') as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(rows)
        for r in rows:
            per[r['Kernel_Name'].split('(')[0].replace('void ', '')][r['Counter_Name']].append(
                float(r['Counter_Value']))
    mean = lambda v: sum(v) / len(v) if v else 0.0
    kern = {}
    for r in csv.DictReader(open(stats)):
        name = r['Name'].split('(')[0].replace('void ', '')
        if 'alsmi::' not in r['Name'] or name not in per:
            continue
        c = per[name]
        cycles = mean(c['GRBM_GUI_ACTIVE']) / 8.0
        kern[name] = {'calls': int(r['Calls']), 'avg_us': float(r['AverageNs']) / 1e3,
                      'hbm_bytes': 2 * mean(c['FETCH_SIZE']) * 1024 + mean(c['WRITE_SIZE']) * 1024,
                      'mfma_util': (mean(c['SQ_VALU_MFMA_BUSY_CYCLES']) / (cycles * N_SIMD))
                      if cycles else None}
        kern[name]['hbm_GBps'] = kern[name]['hbm_bytes'] / (kern[name]['avg_us'] * 1e-6) / 1e9
    line = json.loads(open(os.path.join(src, 'bench.json')).read().strip().splitlines()[-1])
    # per evaluation: one launch of each of these kernels
    per_eval = sum(k['hbm_bytes'] for n, k in kern.items()
                   if any(t in n for t in ('imc_rows', 'gemm_tiles', 'imc_borders',
                                           'reduce_partials')))
    line['counters'] = {'kernels': kern, 'hbm_measured_bytes_per_eval': per_eval,
                        'hbm_measured_GBps': per_eval / (line['cost_grad']['ms'] * 1e-3) / 1e9,
                        'note': 'measured HBM-side traffic (FETCH_SIZE x 2 + WRITE_SIZE, per launch, '
                                'summed over the kernels of one evaluation) beside the algorithmic '
                                'bytes: the item attribute table is cache resident, so the '
                                'algorithmic rate is not an HBM rate',
                        'file': 'profiles/%s_summary.json' % tag}
    with open(os.path.join(dst, stem + '.json'), 'w') as f:
        f.write(json.dumps(line) + '\n')
    return 'imc_c3/%s' % dtype, {'kernels': kern, 'hbm_measured_bytes_per_eval': per_eval,
                                 'algorithmic_bytes_per_eval': line['algorithmic_bytes_per_eval'],
                                 'cost_grad_ms': line['cost_grad']['ms'],
                                 'hessp_ms': line['hessp']['ms']}


def main():
    path = os.path.join(dst, '%s_summary.json' % tag)
    doc = json.load(open(path)) if os.path.exists(path) else {}
    doc.update({
        'commit': subprocess.check_output(['git', '-C', root, 'rev-parse', '--short', 'HEAD'],
                                          text=True).strip(),
        'kernel_source_sha': source_sha(),
        'source': 'tools/profile_round.sh: rocprofv3 --kernel-trace --stats and four separate '
                  '--pmc passes (FETCH_SIZE; WRITE_SIZE; two SQ sets) of bench.py',
        'units': 'FETCH_SIZE / WRITE_SIZE are KB; hbm_bytes = 2 x FETCH_SIZE (gfx950 reports half '
                 'the bytes of wide coalesced reads) + WRITE_SIZE, per half-sweep launch; '
                 'mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)',
    })
    doc.setdefault('configs', {})
    for spec in sys.argv[2:]:
        if spec.startswith('imc_c3'):
            key, entry = collect_imc(spec.partition('/')[2] or 'float64')
            doc['configs'][key] = entry
            print(key, json.dumps({n: {k: v[k] for k in ('avg_us', 'hbm_bytes', 'mfma_util')}
                                   for n, v in entry['kernels'].items()}))
            continue
        key, entry = collect(spec)
        doc['configs'][key] = entry
        print(key, json.dumps({s: {k: entry[s][k] for k in ('hbm_bytes', 'mfma_util')}
                               for s in ('user_side', 'item_side')}))
    json.dump(doc, open(path, 'w'), indent=1)


if __name__ == '__main__':
    main()
