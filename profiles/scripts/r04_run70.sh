# cons_finalize_kernel with one wavefront per workgroup: consensus-related tests, then four bench lines (result_wait - chain = what a batch takes behind its last step)
timeout 1500 python -m pytest tests/test_gpu_consensus.py tests/test_gpu_cyp.py tests/test_gpu_cyp_pipeline.py tests/test_gpu_hla.py tests/test_gpu_hla_pipeline.py tests/test_gpu_cohort_rank.py -x -q 2>&1 | grep -iE "passed|failed|error" | tail -3
for i in 1 2 3 4; do
python bench.py --no-cpu-baseline --no-extra-legs > gpurun_out/r04_f1.json 2> gpurun_out/r04_f1.err; rc=$?
python - <<PY
import json
d=json.loads(open("gpurun_out/r04_f1.json").read().strip().splitlines()[-1])
c=d["critical_path"]["cyp2d6"]; k=d["host_wall_ms"]["cyp2d6"]["k8"]
print($i, "rc", $rc, round(d["value"]), round(d["ms_per_step"],2), "chain", round(c["chain_ms"],2), "result_wait", k["result_wait"][0], "loop", k["loop"][0], {a:round(b,1) for a,b in d["host_wall_ms"]["cyp2d6"].items() if not isinstance(b, dict)}, d["concordance"]["cyp2d6_call_equals_truth"], d["concordance"]["hla_diplotypes_equal_truth"])
PY
done
