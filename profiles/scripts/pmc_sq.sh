#!/bin/bash
# One PMC pass with the SQ instruction counters of the bench step (no tracing domains).  Usage: bash profiles/scripts/pmc_sq.sh <tag>
set -u
TAG=${1:-sq}
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > $OUT/pmc_sq.log 2>&1
python3 profiles/summarize_rocprof.py $OUT 2>&1 | grep "k1_cells_kernel<false, false>"
