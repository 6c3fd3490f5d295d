# first window of a node born in an expansion: 64 columns (default) / 128 / 256 / 32
for v in default wr128 wr256 wr32 default wr128; do
  if [ $v = default ]; then unset SP_LIB_PATH; else export SP_LIB_PATH=$PWD/build/variants/lib_$v.so; fi
  python bench.py --no-cpu-baseline --no-extra-legs > gpurun_out/k8wr.json 2> gpurun_out/k8wr.err
  python - "$v" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/k8wr.json').read().strip().splitlines()[-1]); c = d['critical_path']
print(sys.argv[1], 'reads/s', round(d['value']), 'ms', round(d['ms_per_step'], 2), 'cyp cons', round(d['kernel_ms']['cyp2d6']['cons_steps'], 2), 'steps', round(c['cyp2d6']['dependent_steps'],1), {k:round(x,1) for k,x in c['cyp2d6']['per_step_us'].items()}, 'hla cons', round(d['kernel_ms']['hla']['cons_steps'], 2), d['concordance']['cyp2d6_call_equals_truth'])
PY
  timeout 300 python profiles/scripts/k8persist_dbg3.py "*1/*2" "*4+*68/*1" 2>&1 | grep classic | awk 'NR%2==0' | cut -c1-150
done
