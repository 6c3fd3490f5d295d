# finalize kernel copies the winner's bytes four at a time: consensus + CYP tests, two bench lines
timeout 1500 python -m pytest tests/test_gpu_consensus.py tests/test_gpu_cyp.py tests/test_gpu_cyp_real.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
for i in 1 2; do
python bench.py --no-cpu-baseline > gpurun_out/r04_fz_$i.json 2> gpurun_out/r04_fz_$i.err; echo "run $i rc $?"
python - <<PY
import json
d=json.loads(open("gpurun_out/r04_fz_$i.json").read().strip().splitlines()[-1])
L=d["legs"]
print(round(d["value"]), round(d["ms_per_step"],2), "| pairs", round(L["headline_with_launch_pairs"]["value"]), "| cons_steps", round(d["kernel_ms"]["cyp2d6"]["cons_steps"],2), {k:(round(v,2) if not isinstance(v, dict) else v) for k,v in d["host_wall_ms"]["cyp2d6"].items()}, round(L["cohort"]["samples_per_s"],1), round(L["samples_in_flight"]["value"]), round(L["cyp2d6"]["value"]), d["concordance"]["cyp2d6_call_equals_truth"])
PY
done
