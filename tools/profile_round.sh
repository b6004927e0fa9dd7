#!/bin/bash
# Everything the committed profiles/ files of one configuration come from, in one GPU call (run
# from the repo root on the GPU box):
#     tools/profile_round.sh <tag> <config> [dtype]      e.g.  r02 c2   |   r02 c4slice
# Writes under gpurun_out/<tag>_<config>_<dtype>/: the bench line, rocprofv3 kernel stats of the
# same command, and one --pmc pass each for FETCH_SIZE, WRITE_SIZE and two SQ sets.
set -e
tag=$1; config=${2:-c2}; dtype=${3:-float32}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/${tag}_${config}_${dtype}
mkdir -p "$out"
cd "$root"
steps=10; [ "$config" = c4slice ] && steps=5
timeout -k 10 300 python3 bench.py --config $config --dtype $dtype --steps $steps > "$out/bench.json"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o run -- \
    python3 "$root/bench.py" --config $config --dtype $dtype --steps $steps --cpu-frac 0 --no-f64 > /dev/null 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE" \
    "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" \
    "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
    name=$(echo "$set" | awk '{print tolower($1)}')
    timeout -s KILL 200 rocprofv3 --pmc $set --output-format csv -d "$out/pmc_$name" -o pmc -- \
        python3 "$root/bench.py" --config $config --dtype $dtype --steps 3 --warmup 1 --cpu-frac 0 --no-f64 > /dev/null 2>&1
done
echo done $tag $config $dtype
