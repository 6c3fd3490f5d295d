mkdir -p gpurun_out/r06g
for cfg in "1 4" "2 4" "2 3" "2 5" "1 4" "2 4"; do
  set -- $cfg
  python bench.py --steps 20 --warmup 5 --hla-lanes $1 --cyp-lanes $2 --no-cpu-baseline --no-extra-legs --full-out gpurun_out/r06g/full_h$1_c$2.json > /dev/null 2> gpurun_out/r06g/err.txt
  python - <<PY
import json
d=json.load(open("gpurun_out/r06g/full_h$1_c$2.json"))
cp=d["critical_path"]["cyp2d6"]
print("hla lanes $1 cyp lanes $2: value %.0f ms/step %.2f | cyp chain_ms %.1f | hla k8 %.1f k1 %.1f | lanes work %s" % (d["value"], d["ms_per_step"], cp["chain_ms"],
   d["host_wall_ms"]["hla"]["k8_loop"], d["host_wall_ms"]["hla"]["k1_total"], [round(x["work"],1) for x in d["host_wall_ms"]["lanes_hla_cyp2d6"]]))
PY
done
