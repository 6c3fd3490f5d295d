# the headline's stream (driver's call: --steps 20 --warmup 5) with other numbers of lanes, three runs each
mkdir -p gpurun_out/r06i
for rep in 1 2 3; do
for cfg in "1 4" "1 5" "1 6" "1 8" "2 4" "2 5" "2 6" "2 8"; do
  set -- $cfg
  python bench.py --steps 20 --warmup 5 --hla-lanes $1 --cyp-lanes $2 --no-cpu-baseline --no-extra-legs --full-out gpurun_out/r06i/full.json > /dev/null 2> gpurun_out/r06i/err.txt
  python - <<PY
import json
d=json.load(open("gpurun_out/r06i/full.json"))
print("hla %d cyp %d rep $rep: value %.0f ms/step %.2f | lanes work %s" % ($1, $2, d["value"], d["ms_per_step"], [round(x["work"],1) for x in d["host_wall_ms"]["lanes_hla_cyp2d6"]]))
PY
done
done
