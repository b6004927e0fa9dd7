#!/bin/bash
# Hardware-counter passes over a short bench.py run (run on the GPU box, from the repo root):
#   tools/pmc_sweep.sh <out-prefix> [bench.py args]
# One rocprofv3 run per counter set (the tool does not split sets over passes); CSV files land in
# gpurun_out/<out-prefix>_<set>/.  Summarise with tools/pmc_summary.py.
set -e
prefix=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd /tmp
run() {  # name, counters...
    local name=$1; shift
    timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d "$root/gpurun_out/${prefix}_$name" -o pmc -- \
        python3 "$root/bench.py" --steps 3 --warmup 1 --cpu-frac 0 $BENCH_ARGS > /dev/null 2>&1
}
BENCH_ARGS="$*"
run issue SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
run wait SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE
echo done
