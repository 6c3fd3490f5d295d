# cache of read-set buffers: 64 against 4,096 idle buffers, cohort workload, alternating
for i in 1 2 3; do
for v in c64 default; do
if [ $v = default ]; then unset SP_LIB_PATH; else export SP_LIB_PATH=$PWD/build/variants/lib_$v.so; fi
python bench.py --workload cohort --steps 3 --warmup 1 > gpurun_out/r04_cc.json 2> gpurun_out/r04_cc.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r04_cc.json").read().strip().splitlines()[-1])
print("$v", round(d["value"],1), round(d["ms_per_step"],1))
PY
done
done
