#!/bin/bash
# Kernel statistics and counter passes of the IMC evaluation benchmark at BASELINE config 3's
# shape (run from the repo root on the GPU box):   tools/profile_imc.sh <tag> <dtype>
# Writes under gpurun_out/<tag>_imc_c3_<dtype>/: the bench line, rocprofv3 kernel stats of the same
# command, and one --pmc pass each for FETCH_SIZE, WRITE_SIZE and an SQ set.
set -e
tag=$1; dtype=${2:-float64}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/${tag}_imc_c3_${dtype}
mkdir -p "$out"
cd "$root"
timeout -k 10 300 python3 tools/bench_imc.py --dtype $dtype > "$out/bench.json"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o run -- \
    python3 "$root/tools/bench_imc.py" --dtype $dtype --cpu-frac 0 > /dev/null 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE" \
    "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
    name=$(echo "$set" | awk '{print tolower($1)}')
    timeout -s KILL 200 rocprofv3 --pmc $set --output-format csv -d "$out/pmc_$name" -o pmc -- \
        python3 "$root/tools/bench_imc.py" --dtype $dtype --steps 2 --cpu-frac 0 > /dev/null 2>&1
done
echo done $tag imc_c3 $dtype
