python bench.py > gpurun_out/bench_r04_g.json 2> gpurun_out/bench_r04_g.err
tail -3 gpurun_out/bench_r04_g.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r04_g.json').read().strip().splitlines()[-1])
print(round(d['value']), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms']['cyp2d6'].items()}, round(d['kernel_ms']['hla']['cons_steps'],2), round(d['kernel_ms']['hla']['k1_cells'],2))
print({k:round(v,1) for k,v in d['host_wall_ms']['cyp2d6'].items()}); print({k:round(v,1) for k,v in d['host_wall_ms']['hla'].items()})
print('roofline', {k:d['roofline'][k] for k in ('achieved','frac','traffic','avg_launch_ms')}); print('valu', d['roofline_valu'])
print('cpu', d['cpu_baseline']); print('crit', d['critical_path'])
for k,v in d['legs'].items():
    print(k, {a:b for a,b in v.items() if a in ('value','unit','ms','ms_per_step','by_share_size','calls_equal_truth','samples_per_s')} if isinstance(v,dict) else v)
PY
