timeout 1200 python -m pytest tests/test_gpu_concordance.py -q -s 2>&1 | grep -E "K1 same|consensus pairs|K3 reads|K4 segments|passed|failed|Error" > gpurun_out/r04_concord.log
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r04_gputest_c.log
timeout 2400 python bench.py --no-cpu-baseline > gpurun_out/r04_bench_d.json 2> gpurun_out/r04_bench_d.err
cat gpurun_out/r04_concord.log gpurun_out/r04_gputest_c.log; tail -3 gpurun_out/r04_bench_d.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_bench_d.json'))
print('value', d['value'], 'ms', d['ms_per_step'])
print(json.dumps(d['critical_path'], indent=0)[:1800])
L=d['legs']
print({k:(round(v['ms'],1), v['launch_triples'], v['expansions'], v['call_equals_truth']) for k,v in L['cyp2d6']['scenarios'].items()})
c=L['cohort']; print('cohort', c['samples_per_s'], c['calls_equal_truth'], c['rank0_host_seconds_per_pass']); print(json.dumps(c['by_share_size']))
print('in flight', L['samples_in_flight'].get('value'), 'hla_resident', L['hla_resident']['value'])
PY
