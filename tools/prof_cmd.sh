#!/bin/bash
# rocprofv3 kernel statistics of an arbitrary python script of this repository (run on the GPU
# box from the repo root):  tools/prof_cmd.sh <tag> <script.py> [args...]
#   -> gpurun_out/<tag>/stats/<tag>_kernel_stats.csv and the script's stdout in out.json
set -e
tag=$1; script=$2; shift; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o "$tag" -- \
    python3 "$root/$script" "$@" > "$out/out.json" 2> "$out/err.txt"
python3 - "$out/stats/${tag}_kernel_stats.csv" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print('%-100s calls %6s  avg %10.1f us  total %8.2f ms' % (
        r['Name'][:100], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
