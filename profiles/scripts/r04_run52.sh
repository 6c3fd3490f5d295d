# the control step by part, in the headline and for a sample alone
python bench.py --no-cpu-baseline > gpurun_out/r04_cp.json 2> gpurun_out/r04_cp.err; echo "rc $?"
python - <<PY
import json
d=json.loads(open("gpurun_out/r04_cp.json").read().strip().splitlines()[-1])
print(round(d["value"]), round(d["ms_per_step"],2))
for k in ("cyp2d6","hla"):
    c=d["critical_path"][k]; print(k, c["dependent_steps"], {a:round(b,1) for a,b in c["per_step_us"].items()}, {a:round(b,1) for a,b in c["control_parts_us_per_step"].items()})
PY
