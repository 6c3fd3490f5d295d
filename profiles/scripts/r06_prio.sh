# (an experiment of round 6 -- profiles/r06/lanes_matrix.txt (d): no effect; the switch it sets was taken out of the library afterwards)
mkdir -p gpurun_out/r06f
for pr in 0 2 3 0 2 3; do
  SP_K8_CTL_PRIO=$pr python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --full-out gpurun_out/r06f/full_p$pr.json > /dev/null 2> gpurun_out/r06f/err.txt
  python - <<PY
import json
d=json.load(open("gpurun_out/r06f/full_p$pr.json"))
cp=d["critical_path"]["cyp2d6"]
print("prio $pr: value %.0f ms/step %.2f | cyp chain_ms %.1f per_step %s | hla k8 %.1f k1 %.1f | lanes work %s" % (d["value"], d["ms_per_step"], cp["chain_ms"], {k: round(v,1) for k,v in cp["per_step_us"].items()},
   d["host_wall_ms"]["hla"]["k8_loop"], d["host_wall_ms"]["hla"]["k1_total"], [round(x["work"],1) for x in d["host_wall_ms"]["lanes_hla_cyp2d6"]]))
PY
done
