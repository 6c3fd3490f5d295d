timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 900 python profiles/scripts/k8fuzz.py 2>&1 | tail -3
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_k8c.json 2> gpurun_out/bench_k8c.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_k8c.json'))
print('value', d['value'], 'ms', d['ms_per_step'])
print(d['kernel_ms'])
print(d['consensus'])
print('cyp', d['cyp2d6']['value'], d['cyp2d6']['calls_equal_truth'], 'cohort', d['cohort']['value'], d['cohort']['ms'], d['cohort']['calls_equal_truth'], 'k5', d['k5_chain_pairs']['value'])
PY
tail -3 gpurun_out/bench_k8c.err
