for m in 1 0; do
SP_K8_PERSISTENT=$m timeout 900 python bench.py --no-cpu-baseline --no-extra-legs --steps 10 --warmup 3 > gpurun_out/r04_ab_$m.json 2> gpurun_out/r04_ab_$m.err
python - <<PY
import json
d=json.load(open('gpurun_out/r04_ab_$m.json'))
c=d['critical_path']['cyp2d6']
print('persistent=$m value', round(d['value']), 'ms', round(d['ms_per_step'],1), 'cyp cons', round(d['kernel_ms']['cyp2d6']['cons_steps'],1), 'hla cons', round(d['kernel_ms']['hla']['cons_steps'],1), 'steps', c['dependent_steps'], {k: round(v,1) for k,v in c.get('per_step_us',{}).items()}, d['concordance'])
PY
done
