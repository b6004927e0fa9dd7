#!/bin/bash
# Everything the committed profiles/ files come from, in one GPU call (run from the repo root on
# the GPU box):  tools/profile_round.sh <tag>      e.g. r01
# Writes under gpurun_out/<tag>/: the default bench line, rocprofv3 kernel stats of the same
# command, and one --pmc pass each for FETCH_SIZE, WRITE_SIZE and two SQ sets.
set -e
tag=$1
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
timeout -k 10 300 python3 bench.py > "$out/bench_c2.json"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o c2 -- \
    python3 "$root/bench.py" --cpu-frac 0 > /dev/null 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE" \
    "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" \
    "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
    name=$(echo "$set" | awk '{print tolower($1)}')
    timeout -s KILL 150 rocprofv3 --pmc $set --output-format csv -d "$out/pmc_$name" -o pmc -- \
        python3 "$root/bench.py" --steps 3 --warmup 1 --cpu-frac 0 > /dev/null 2>&1
done
echo done
