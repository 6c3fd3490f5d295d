# lanes x steps: is the choice robust against the length of the run?
mkdir -p gpurun_out/r06i
for st in "20 5" "24 2" "30 5"; do
for cfg in "1 4" "2 4" "2 6" "2 7" "2 8" "3 6" "3 8"; do
  set -- $cfg $st
  python bench.py --steps $3 --warmup $4 --hla-lanes $1 --cyp-lanes $2 --no-cpu-baseline --no-extra-legs --full-out gpurun_out/r06i/full.json > /dev/null 2> gpurun_out/r06i/err.txt
  python - <<PY
import json
d=json.load(open("gpurun_out/r06i/full.json"))
print("steps %d hla %d cyp %d: value %.0f ms/step %.2f | lanes work %s" % ($3, $1, $2, d["value"], d["ms_per_step"], [round(x["work"],1) for x in d["host_wall_ms"]["lanes_hla_cyp2d6"]]))
PY
done
done
