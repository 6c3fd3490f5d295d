# the default headline (persistent consensus on the CYP2D6 context) ten times in a row: value, ms, mode, diplotypes
for i in 1 2 3 4 5 6 7 8 9 10; do
python bench.py --no-cpu-baseline --no-extra-legs > gpurun_out/r04_st_$i.json 2> gpurun_out/r04_st_$i.err; rc=$?
python - <<PY
import json
d=json.loads(open("gpurun_out/r04_st_$i.json").read().strip().splitlines()[-1])
print($i, "rc", $rc, round(d["value"]), round(d["ms_per_step"],2), d["config"]["cyp2d6_consensus"][:18], round(d["kernel_ms"]["cyp2d6"]["cons_steps"],2), d["concordance"]["hla_diplotypes_equal_truth"], d["concordance"]["cyp2d6_call_equals_truth"])
PY
done
