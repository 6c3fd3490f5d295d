# control search without the 64-bit remainders, the single waiting node read out of its lane: consensus + CYP tests, K8 fuzz, two bench lines
timeout 1500 python -m pytest tests/test_gpu_consensus.py tests/test_gpu_cyp.py tests/test_gpu_cyp_pipeline.py tests/test_gpu_hla.py -x -q 2>&1 | grep -iE "passed|failed|error" | tail -3
timeout 600 python profiles/scripts/k8fuzz.py 2000 2>&1 | tail -1
for i in 1 2; do
python bench.py --no-cpu-baseline > gpurun_out/r04_pm_$i.json 2> gpurun_out/r04_pm_$i.err; echo "run $i rc $?"
python - <<PY
import json
d=json.loads(open("gpurun_out/r04_pm_$i.json").read().strip().splitlines()[-1])
L=d["legs"]; c=d["critical_path"]["cyp2d6"]; o=L["headline_with_launch_pairs"]["critical_path_cyp2d6"]
print(round(d["value"]), round(d["ms_per_step"],2), "| pairs", round(L["headline_with_launch_pairs"]["value"]), "| chain", round(c["chain_ms"],2), {a:round(b,1) for a,b in c["per_step_us"].items()}, {a:round(b,1) for a,b in c["control_parts_us_per_step"].items()}, "| pairs:", {a:round(b,1) for a,b in o["control_parts_us_per_step"].items()}, round(L["cohort"]["samples_per_s"],1), round(L["samples_in_flight"]["value"]))
PY
done
