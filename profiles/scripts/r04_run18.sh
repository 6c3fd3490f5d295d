# the control kernel's workgroup size (classic mode): a 1,024-thread workgroup at 128 registers needs an empty CU
for t in 1024 256 1024 256 512 128; do
SP_K8_CTL_THREADS=$t python bench.py --no-cpu-baseline --no-extra-legs > gpurun_out/r04_ct.json 2> gpurun_out/r04_ct.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r04_ct.json').read().strip().splitlines()[-1])
c=d['critical_path']
print($t, 'headline', round(d['value']), round(d['ms_per_step'],2), 'cyp cons', round(d['kernel_ms']['cyp2d6']['cons_steps'],2), {k:round(v,1) for k,v in c['cyp2d6']['per_step_us'].items()}, 'hla cons', round(d['kernel_ms']['hla']['cons_steps'],2), {k:round(v,1) for k,v in c['hla']['per_step_us'].items()}, d['concordance']['cyp2d6_call_equals_truth'])
PY
done
for t in 1024 256; do SP_K8_CTL_THREADS=$t timeout 300 python profiles/scripts/k8persist_dbg3.py "*1/*2" 2>&1 | grep classic | tail -1; done
