#!/bin/bash
# Kernel statistics of the fp64 short-row kernels for timing builds of the library (built with
# -DDUAL_D_ABLATE=1/2/4: no solve / no second pass / no matrix instructions), given as
# arguments; run on the GPU box from the repo root.
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
export TMPDIR=/tmp
cd /tmp
for lib in "$@"; do
    export ALSMI_LIB=$(realpath $root/$lib)
    out=$root/gpurun_out/parts_$(basename $lib .so)
    mkdir -p $out
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- \
        python3 $root/tools/debug/time_half.py r128 float64 > $out/log.txt 2>&1
    echo "== $lib"
    python3 - $out/p_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'dual_d' in r['Name'] or 'solve_rows_kernel' in r['Name']:
        print('   %-70s calls %4s avg %9.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
