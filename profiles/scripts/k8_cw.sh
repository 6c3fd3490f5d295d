# K8 window length: the library at 64 / 128 / 256 columns per window (build/variants/lib_cw*.so are built on the CPU box beforehand:
# hipcc ... -DSP_K8_CW=64 -c sp_consensus.hip), the headline step each, then the suite's consensus tests and the fuzz on the default build
mkdir -p gpurun_out
for v in cw64 cw128 default; do
  if [ $v = default ]; then unset SP_LIB_PATH; else export SP_LIB_PATH=$PWD/build/variants/lib_$v.so; fi
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra-legs > gpurun_out/k8cw_$v.json 2> gpurun_out/k8cw_$v.err
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
d = json.loads(open(f'gpurun_out/k8cw_{v}.json').read().strip().splitlines()[-1])
print(v, 'reads/s', round(d['value']), 'ms', round(d['ms_per_step'], 2), 'cons_steps', round(d['kernel_ms']['cons_steps'], 2), d['consensus'], d['concordance']['diplotypes_equal_truth'])
PY
done
unset SP_LIB_PATH
