#!/bin/bash
# Collect the rocprofv3 evidence bench.py's roofline block refers to.  Run on the GPU box:
#   gpurun -- 'bash profiles/run_rocprof.sh r01_v0'
# Writes under gpurun_out/prof_<tag>/ ; the summaries worth keeping are copied to profiles/ by hand.
set -u
TAG=${1:-run}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# the bench asks for 8 hardware queues through os.environ; under rocprofv3 the runtime is up before Python starts, so the shell sets it
export GPU_MAX_HW_QUEUES=16
# (the kernel trace runs the bench as it is -- the headline's CYP2D6 context with persistent consensus kernels --; the counter passes run one kernel at a time, under which
#  two kernels that wait for each other cannot run: they get SP_BENCH_HEADLINE_PERSISTENT=0, i.e. a launch pair per step; K1, the kernel the roofline is about, is the same in both)
# six steps = one turn through the configs[2] scenarios; SP_PROF_STEPS = steps + warm-up (summarize_rocprof.py divides the pass sums by it)
ARGS="bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-extra-legs"
export SP_PROF_STEPS=7
# 1. kernel trace + stats (no counters in this pass)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
# 2. PMC passes, one small counter group each (no tracing domains combined with --pmc)
export SP_BENCH_HEADLINE_PERSISTENT=0
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_lds -- python3 $ARGS > $OUT/pmc_lds.log 2>&1
python3 profiles/summarize_rocprof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
