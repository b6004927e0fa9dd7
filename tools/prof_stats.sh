#!/bin/bash
# rocprofv3 kernel statistics of one bench.py command (run on the GPU box from the repo root):
#   tools/prof_stats.sh <tag> <bench args...>    -> gpurun_out/<tag>/stats/<tag>_kernel_stats.csv
set -e
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o "$tag" -- \
    python3 "$root/bench.py" --cpu-frac 0 "$@" > "$out/bench.json" 2> "$out/bench.err"
python3 - "$out/stats/${tag}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print('%-90s calls %6s  avg %10.1f us  total %8.2f ms  %5s%%' % (
        r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6,
        r['Percentage']))
PY
