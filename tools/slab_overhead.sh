#!/bin/bash
# Cost of cutting a half-sweep into row slabs on ONE GPU (forced one-rank collectives):
# bench.py at the given config for several user/item slab counts.
# usage: tools/slab_overhead.sh <outdir> [config]
out=${1:-gpurun_out/slabs}; cfg=${2:-c2}
mkdir -p "$out"
for us in 1 2 4; do for is in 1 2; do
  ALSMI_FORCE_COLLECTIVES=1 ALSMI_USER_SLABS=$us ALSMI_ITEM_SLABS=$is timeout -k 10 300 \
    python bench.py --config "$cfg" --steps 10 --warmup 3 --no-f64 --cpu-frac 0 \
    > "$out/${cfg}_u${us}_i${is}.json" 2> "$out/${cfg}_u${us}_i${is}.err" || exit 1
  python - "$out/${cfg}_u${us}_i${is}.json" $us $is <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d['roofline']
print('user slabs %s item slabs %s: %.3f ms/sweep (user %.3f item %.3f) exposed %.3f'
      % (sys.argv[2], sys.argv[3], d['ms_per_step'], r['user_side']['ms'], r['item_side']['ms'],
         d.get('allgather_exposed_ms') or 0))
PY
done; done
