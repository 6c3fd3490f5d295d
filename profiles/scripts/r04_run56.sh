# a lane's host time per step outside the library calls
python bench.py --no-cpu-baseline --no-extra-legs > gpurun_out/r04_lane.json 2> gpurun_out/r04_lane.err; echo "rc $?"
python - <<PY
import json
d=json.loads(open("gpurun_out/r04_lane.json").read().strip().splitlines()[-1])
print(round(d["value"]), round(d["ms_per_step"],2), d["host_wall_ms"]["lanes_hla_cyp2d6"])
PY
