# anchor kernel with one phase taken out at a time (results are wrong; only the "anchor" time is read):
# a1 no binary search steps, a2 no single-occurrence votes, a3 peak sweep over 1,024 bins only, a4 no histogram clearing
for v in a1 a2 a3 a4; do
  SP_LIB_PATH=$PWD/build/variants/lib_$v.so python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs > gpurun_out/bv_$v.json 2> gpurun_out/bv_$v.err
  python -c "
import json;d=json.loads(open('gpurun_out/bv_$v.json').read().strip().splitlines()[-1]);print('$v',round(d['value']),round(d['kernel_ms']['anchor'],3))" || tail -3 gpurun_out/bv_$v.err
done
