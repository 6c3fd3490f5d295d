# read-set buffers cached in the context instead of hipFree / hipMalloc per sample: whole GPU suite, thread stress, then two bench lines with the lanes' host time
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -5
timeout 600 python profiles/scripts/thread_stress.py 4 60 2>&1 | tail -3
for i in 1 2; do
python bench.py --no-cpu-baseline > gpurun_out/r04_dc_$i.json 2> gpurun_out/r04_dc_$i.err; echo "run $i rc $?"; tail -2 gpurun_out/r04_dc_$i.err | cut -c1-300
python - <<PY
import json
d=json.loads(open("gpurun_out/r04_dc_$i.json").read().strip().splitlines()[-1])
print(round(d["value"]), round(d["ms_per_step"],2), d["host_wall_ms"]["lanes_hla_cyp2d6"], round(d["legs"]["cohort"]["samples_per_s"],1), round(d["legs"]["samples_in_flight"]["value"]), round(d["legs"]["cyp2d6"]["value"]), round(d["legs"]["hla_resident"]["value"]), round((d["legs"].get("headline_with_persistent_consensus") or d["legs"].get("headline_with_launch_pairs"))["value"]))
PY
done
