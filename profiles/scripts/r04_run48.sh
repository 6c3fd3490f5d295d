# K9: tables in LDS, run up the column without the test, pooled buffers in sp_align_batch, only CYP2D6-typed sequences on the backbone: CYP tests, the whole-call fuzz, two bench lines
timeout 2400 python -m pytest tests/test_gpu_cyp.py tests/test_gpu_cyp_real.py tests/test_gpu_cyp_pipeline.py tests/test_gpu_concordance.py tests/test_gpu_cohort_rank.py -x -q 2>&1 | tail -5
for i in 1 2; do
python bench.py --no-cpu-baseline > gpurun_out/r04_k9_$i.json 2> gpurun_out/r04_k9_$i.err; echo "run $i rc $?"; tail -2 gpurun_out/r04_k9_$i.err | cut -c1-300
python - <<PY
import json
d=json.loads(open("gpurun_out/r04_k9_$i.json").read().strip().splitlines()[-1])
print(round(d["value"]), round(d["ms_per_step"],2), d["kernel_ms"]["cyp2d6"], {k:round(v,2) for k,v in d["host_wall_ms"]["cyp2d6"].items() if not isinstance(v, dict)}, round(d["legs"]["cohort"]["samples_per_s"],1), round((d["legs"].get("headline_with_persistent_consensus") or d["legs"].get("headline_with_launch_pairs"))["value"]), round(d["legs"]["samples_in_flight"]["value"]), round(d["legs"]["cyp2d6"]["value"]))
print({k: (round(v["ms"],1), v["host_wall_ms"]["merge"]) for k, v in d["legs"]["cyp2d6"]["scenarios"].items()})
PY
done
