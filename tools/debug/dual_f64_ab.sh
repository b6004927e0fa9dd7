#!/bin/bash
# fp64 half-sweeps with and without the neighbour-by-neighbour form of short rows (run on the
# GPU box from the repo root): tools/debug/dual_f64_ab.sh [config ...]
set -e
for cfg in "${@:-r128}"; do
    for on in 0 1 0 1; do
        ALSMI_DUAL_F64=$on timeout -k 10 200 python3 bench.py --config $cfg --dtype float64 --cpu-frac 0 --steps 5 |
            python3 -c "
import json, sys
j = json.loads(sys.stdin.readlines()[-1])
r = j.get('roofline') or {}
print('$cfg dual_f64=$on  ms/step %.3f  user %.3f  item %.3f  rmse %s' % (
    j['ms_per_step'], (r.get('user_side') or {}).get('ms', float('nan')),
    (r.get('item_side') or {}).get('ms', float('nan')), (j.get('train_rmse') or {}).get('last')))"
    done
done
