for v in 4_1 2_1 1_1; do
  SP_LIB_PATH=$PWD/build/variants/lib_$v.so python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs > gpurun_out/bv_$v.json 2> gpurun_out/bv_$v.err
  python -c "
import json;d=json.loads(open('gpurun_out/bv_$v.json').read().strip().splitlines()[-1]);print('$v',round(d['value']),round(d['kernel_ms']['cons_steps'],2),d['concordance']['diplotypes_equal_truth'])"
done
