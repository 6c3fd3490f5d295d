# the cache of read-set buffers holds up to 4,096 buffers (a cohort pass frees a thousand at once): cohort workload three times, then the upload tests
for i in 1 2 3; do
python bench.py --workload cohort --steps 3 --warmup 1 > gpurun_out/r04_co_$i.json 2> gpurun_out/r04_co_$i.err; echo "rc $?"
python - <<PY
import json
d=json.loads(open("gpurun_out/r04_co_$i.json").read().strip().splitlines()[-1])
print(round(d["value"],1), d["unit"], round(d["ms_per_step"],1), {k: v for k, v in d.items() if k in ("calls_equal_truth",)}, str(d.get("host_seconds_per_pass", d.get("seconds", "")))[:200])
PY
done
timeout 900 python -m pytest tests/test_gpu_upload.py tests/test_gpu_sample.py -x -q 2>&1 | grep -iE "passed|failed|error" | tail -3
