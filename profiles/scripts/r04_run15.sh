export SP_K8_PERSISTENT=1
timeout 1500 python -m pytest tests/test_gpu_consensus.py tests/test_gpu_cyp_real.py tests/test_gpu_cyp.py tests/test_gpu_hla_pipeline.py tests/test_gpu_concordance.py tests/test_gpu_cohort_rank.py tests/test_gpu_sample.py -x -q 2>&1 | tail -3
timeout 900 python profiles/scripts/k8fuzz.py 2>&1 | tail -2
unset SP_K8_PERSISTENT
for m in 0 1 0 1; do
SP_BENCH_COHORT_PERSISTENT=$m python bench.py --workload cohort --steps 2 --warmup 1 > gpurun_out/r04_co_$m.json 2> gpurun_out/r04_co_$m.err
python -c "
import json;d=json.loads(open('gpurun_out/r04_co_$m.json').read().strip().splitlines()[-1]);c=d['cohort'];print($m, round(c['samples_per_s'],1), round(c['ms_per_step'],1), c['rank0_host_seconds_per_pass'], c['calls_equal_truth'])"
done
