# round 5: kernel stats of the configs[4] cohort workload (256 samples, one pass + one warm-up pass)
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
export GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out/prof_r05c
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r05c/cohort -- python3 bench.py --workload cohort --steps 1 --warmup 1 > gpurun_out/prof_r05c/cohort.log 2>&1
tail -2 gpurun_out/prof_r05c/cohort.log | cut -c1-600
find gpurun_out/prof_r05c -name "*kernel_stats.csv" | head
