#!/bin/bash
# LDS-related counters of a short bench.py run:  tools/pmc_lds.sh <out-prefix> [bench.py args]
set -e
prefix=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
    --output-format csv -d "$root/gpurun_out/${prefix}_lds" -o pmc -- \
    python3 "$root/bench.py" --steps 3 --warmup 1 --cpu-frac 0 "$@" > /dev/null 2>&1
echo done
