# round 4 evidence: kernel trace + stats and the PMC passes of the headline workload, then the kernel stats of the cohort workload
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
bash profiles/run_rocprof.sh r04 > gpurun_out/prof_r04.log 2>&1
export GPU_MAX_HW_QUEUES=16
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r04/cohort -- python3 bench.py --workload cohort --steps 1 --warmup 1 > gpurun_out/prof_r04/cohort.log 2>&1
ls -R gpurun_out/prof_r04 | head -60
