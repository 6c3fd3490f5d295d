mkdir -p gpurun_out/r06c
export SP_K8_COMPOUND=25 SP_K8_SIDE_ORDERS=1
for hl in 2; do for lanes in 4 6 8; do
  python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-extra-legs --cyp-lanes $lanes --hla-lanes $hl --full-out gpurun_out/r06c/full_h${hl}_l$lanes.json > /dev/null 2> gpurun_out/r06c/err.txt
  python - <<PY
import json
d=json.load(open("gpurun_out/r06c/full_h${hl}_l$lanes.json"))
cp=d["critical_path"]["cyp2d6"]
print("hla lanes $hl cyp lanes $lanes: value %.0f ms/step %.2f | chain steps %.0f chain_ms %.1f per_step %s | lanes work %s" % (d["value"], d["ms_per_step"], cp["dependent_steps"], cp["chain_ms"], {k: round(v,1) for k,v in cp["per_step_us"].items()},
   [round(x["work"],1) for x in d["host_wall_ms"]["lanes_hla_cyp2d6"]]))
PY
done; done
