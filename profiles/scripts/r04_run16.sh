python bench.py --no-cpu-baseline > gpurun_out/bench_r04_h.json 2> gpurun_out/bench_r04_h.err
tail -2 gpurun_out/bench_r04_h.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r04_h.json').read().strip().splitlines()[-1])
print(round(d['value']), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms']['cyp2d6'].items()}, round(d['kernel_ms']['hla']['cons_steps'],2), round(d['kernel_ms']['hla']['k1_cells'],2))
print({k:round(v,1) for k,v in d['host_wall_ms']['cyp2d6'].items()}); print({k:round(v,1) for k,v in d['host_wall_ms']['hla'].items()})
print('crit', d['critical_path']['cyp2d6'])
for k,v in d['legs'].items():
    print(k, {a:b for a,b in v.items() if a in ('value','unit','ms','ms_per_step','by_share_size','calls_equal_truth','samples_per_s')} if isinstance(v,dict) else v)
PY
timeout 900 python -m pytest tests/test_gpu_bench.py -x -q 2>&1 | tail -2
