# K8 window length 256 (default) / 320 / 384 columns: parity of the variants on the consensus tests, then the headline and two scenarios
for v in cw320 cw384; do
  SP_LIB_PATH=$PWD/build/variants/lib_$v.so timeout 600 python -m pytest tests/test_gpu_consensus.py -x -q 2>&1 | tail -1
done
for v in default cw320 cw384 default cw320 cw384; do
  if [ $v = default ]; then unset SP_LIB_PATH; else export SP_LIB_PATH=$PWD/build/variants/lib_$v.so; fi
  python bench.py --no-cpu-baseline --no-extra-legs > gpurun_out/k8cw.json 2> gpurun_out/k8cw.err
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
d = json.loads(open('gpurun_out/k8cw.json').read().strip().splitlines()[-1]); c = d['critical_path']
print(v, 'reads/s', round(d['value']), 'ms', round(d['ms_per_step'], 2), 'cyp cons', round(d['kernel_ms']['cyp2d6']['cons_steps'], 2), round(c['cyp2d6']['dependent_steps'],1), {k:round(x,1) for k,x in c['cyp2d6']['per_step_us'].items()}, 'hla cons', round(d['kernel_ms']['hla']['cons_steps'], 2), round(c['hla']['dependent_steps'],1), d['concordance']['cyp2d6_call_equals_truth'], d['concordance']['hla_diplotypes_equal_truth'])
PY
  timeout 300 python profiles/scripts/k8persist_dbg3.py "*1/*2" "*4+*68/*1" 2>&1 | grep classic | awk 'NR%2==0' | cut -c1-160
done
