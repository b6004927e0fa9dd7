#!/bin/bash
# Accumulate-only / solve-only timing builds of the padded-rank-128 row kernel: kernel statistics
# of the plain half-sweeps at the config-4 slice for ALSMI_ABLATE = 0, 1 (no solve), 2 (no
# accumulation).  Run on the GPU box from the repo root: tools/debug/ablate_r128.sh
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
export TMPDIR=/tmp
cd /tmp
for a in 0 1 2; do
    export ALSMI_ABLATE=$a
    out=$root/gpurun_out/abl$a
    mkdir -p $out
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o abl -- \
        python3 $root/tools/debug/time_half.py c4slice > $out/log.txt 2>&1
    echo "ALSMI_ABLATE=$a"; cat $out/log.txt | grep half
    python3 - $out/abl_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'solve_rows_kernel' in r['Name'] or 'gram_chunks' in r['Name']:
        print('   %-70s calls %4s avg %9.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
