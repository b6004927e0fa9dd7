#!/bin/bash
# Memory-path counters (vector L1, address and data units) of a short bench.py run:
#   tools/pmc_mem.sh <out-prefix> [bench.py args]     (summarise with tools/pmc_dispatch.py)
set -e
prefix=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 150 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ_LATENCY TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TD_TD_BUSY TD_TC_STALL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE \
    --output-format csv -d "$root/gpurun_out/${prefix}_mem" -o pmc -- \
    python3 "$root/bench.py" --steps 3 --warmup 1 --cpu-frac 0 "$@" > /dev/null 2>&1
echo done
