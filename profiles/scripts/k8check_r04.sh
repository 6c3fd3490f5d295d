timeout 1500 python -m pytest tests/test_gpu_consensus.py tests/test_gpu_cyp_real.py tests/test_gpu_cyp_pipeline.py tests/test_gpu_hla_pipeline.py -q 2>&1 | tail -8
timeout 1500 python profiles/scripts/k8fuzz.py 1,2,3,4,5,6,7,8,1001,1002,1003,1004,1005,2001,2002,2003,2004,2005,2006,2007,2008 2>&1 | tail -12
python bench.py --no-cpu-baseline > gpurun_out/r04_bench_b.json 2> gpurun_out/r04_bench_b.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_bench_b.json'))
print('value', d['value'], 'ms', d['ms_per_step'])
print(d['kernel_ms'])
L=d['legs']
print({k:(v['ms'], v['launch_triples'], v['expansions'], v['call_equals_truth']) for k,v in L['cyp2d6']['scenarios'].items()})
print('hla_resident', L['hla_resident']['value'], 'cohort', L['cohort']['samples_per_s'], L['cohort']['calls_equal_truth'], L['cohort']['rank0_host_seconds_per_pass'])
PY
tail -3 gpurun_out/r04_bench_b.err
