# K9 with its tables and alternative bases in LDS: the K9 tests, then two bench lines
timeout 1500 python -m pytest tests/test_gpu_cyp.py tests/test_gpu_cyp_real.py -x -q 2>&1 | tail -5
for i in 1 2; do
python bench.py --no-cpu-baseline > gpurun_out/r04_k9_$i.json 2> gpurun_out/r04_k9_$i.err; echo "run $i rc $?"; tail -2 gpurun_out/r04_k9_$i.err | cut -c1-300
python - <<PY
import json
d=json.loads(open("gpurun_out/r04_k9_$i.json").read().strip().splitlines()[-1])
print(round(d["value"]), round(d["ms_per_step"],2), d["kernel_ms"]["cyp2d6"], {k:round(v,2) for k,v in d["host_wall_ms"]["cyp2d6"].items() if not isinstance(v, dict)}, round(d["legs"]["cohort"]["samples_per_s"],1), round((d["legs"].get("headline_with_persistent_consensus") or d["legs"].get("headline_with_launch_pairs"))["value"]))
PY
done
