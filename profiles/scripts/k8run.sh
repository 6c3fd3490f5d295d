timeout 800 python profiles/scripts/k8fuzz.py 2>&1 | tail -12
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_k8b.json 2> gpurun_out/bench_k8b.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_k8b.json'))
print('value', d['value'], 'ms', d['ms_per_step'])
print(d['kernel_ms'])
print(d['consensus'])
for k,v in d['cyp2d6']['scenarios'].items(): print(k, {a:(round(b,1) if isinstance(b,float) else b) for a,b in v.items()})
print('cyp', d['cyp2d6']['value'], d['cyp2d6']['calls_equal_truth'], 'cohort', d['cohort']['value'], d['cohort']['ms'], d['cohort']['calls_equal_truth'])
PY
tail -3 gpurun_out/bench_k8b.err
bash profiles/scripts/prof_e2e.sh k8s2 2>&1 | head -6
