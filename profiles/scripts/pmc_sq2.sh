#!/bin/bash
# Second SQ counter group (waits / LDS / issue) of the bench step.  Usage: bash profiles/scripts/pmc_sq2.sh <tag>
set -u
TAG=${1:-sq2}
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_a -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_LEVEL_LDS --output-format csv -d $OUT/pmc_b -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > $OUT/b.log 2>&1
python3 profiles/summarize_rocprof.py $OUT 2>&1 | grep "k1_cells_kernel<false, false>"
tail -3 $OUT/b.log
