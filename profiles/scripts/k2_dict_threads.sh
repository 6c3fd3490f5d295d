# K2 dictionary anchors with 128 / 256 (default) / 512 threads per pair: bench step and the 32-sample cohort call
for v in t128 default t512; do
  if [ $v = default ]; then unset SP_LIB_PATH; else export SP_LIB_PATH=$PWD/build/variants/lib_$v.so; fi
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs > gpurun_out/bv_$v.json 2> gpurun_out/bv_$v.err
  python -c "
import json;d=json.loads(open('gpurun_out/bv_$v.json').read().strip().splitlines()[-1]);print('$v',round(d['value']),{k:round(x,3) for k,x in d['kernel_ms'].items() if 'anchor' in k}, d['concordance']['diplotypes_equal_truth'])" || tail -3 gpurun_out/bv_$v.err
  python profiles/scripts/cohort_profile.py 2>&1 | grep "samples_per_s\|anchor_k2"
done
