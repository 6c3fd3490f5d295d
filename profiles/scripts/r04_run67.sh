# the bench's fall-back from persistent consensus kernels to launch pairs: (1) two hardware queues for the whole process -- the mode still runs, slower; (2) an injected failure
GPU_MAX_HW_QUEUES=2 timeout 600 python bench.py --no-cpu-baseline --no-extra-legs --steps 3 --warmup 1 > gpurun_out/r04_fb.json 2> gpurun_out/r04_fb.err; echo "two queues: rc $?"
python - <<PY
import json
d=json.loads(open("gpurun_out/r04_fb.json").read().strip().splitlines()[-1])
print(round(d["value"]), round(d["ms_per_step"],2), d["config"]["cyp2d6_consensus"][:40], "queues", d["context"]["hw_queues"], d["concordance"]["cyp2d6_call_equals_truth"])
PY
SP_BENCH_INJECT_FAILURE=1 timeout 600 python bench.py --no-cpu-baseline --no-extra-legs --steps 3 --warmup 1 > gpurun_out/r04_fb2.json 2> gpurun_out/r04_fb2.err; echo "injected failure: rc $?"
grep -v amdgpu.ids gpurun_out/r04_fb2.err | tail -2 | cut -c1-300
python - <<PY
import json
d=json.loads(open("gpurun_out/r04_fb2.json").read().strip().splitlines()[-1])
print(round(d["value"]), round(d["ms_per_step"],2), d["config"]["cyp2d6_consensus"][:140], d["concordance"]["cyp2d6_call_equals_truth"])
PY
