# the wide-band retry of untraced cells on 8 instead of 2 workgroups per CU: alignment + CYP tests, three bench lines
timeout 1500 python -m pytest tests/test_gpu_align.py tests/test_gpu_cyp.py tests/test_gpu_cyp_real.py tests/test_gpu_cyp_pipeline.py tests/test_gpu_concordance.py tests/test_gpu_hla.py -x -q 2>&1 | grep -iE "passed|failed|error" | tail -3
for i in 1 2 3; do
python bench.py --no-cpu-baseline > gpurun_out/r04_wd.json 2> gpurun_out/r04_wd.err; rc=$?
python - <<PY
import json
d=json.loads(open("gpurun_out/r04_wd.json").read().strip().splitlines()[-1]); L=d["legs"]
print($i, "rc", $rc, round(d["value"]), round(d["ms_per_step"],2), {a:round(b,1) for a,b in d["host_wall_ms"]["cyp2d6"].items() if not isinstance(b, dict)}, "| pairs", round(L["headline_with_launch_pairs"]["value"]), round(L["cohort"]["samples_per_s"],1), round(L["samples_in_flight"]["value"]), round(L["cyp2d6"]["value"]), {k: v["host_wall_ms"]["weights"] for k, v in L["cyp2d6"]["scenarios"].items()})
PY
done
