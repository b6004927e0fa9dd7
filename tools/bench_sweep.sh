#!/bin/bash
# Usage: tools/bench_sweep.sh "<label>=<extra bench.py args>" ...   (one bench run per argument)
# Prints: label, ms/step, user-side ms, item-side ms, roofline frac, last RMSE.
set -e
cd "$(dirname "$0")/.."
for spec in "$@"; do
    label="${spec%%=*}"
    args="${spec#*=}"
    timeout -k 10 300 python -u bench.py --steps 10 --warmup 2 --cpu-frac 0 $args 2>/dev/null > /tmp/bench_sweep.json
    LABEL="$label" python - <<'PY'
import json, os
d = json.loads(open('/tmp/bench_sweep.json').read().strip().splitlines()[-1])
r = d['roofline']
print('%-14s ms/step %.3f  user %.3f  item %.3f  frac %.3f  rmse %.7f' % (
    os.environ['LABEL'], d['ms_per_step'], r['user_side']['ms'], r['item_side']['ms'],
    r['frac'], d['train_rmse']['last']))
PY
done
