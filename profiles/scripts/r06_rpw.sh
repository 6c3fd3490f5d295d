# (an experiment of round 6 -- profiles/r06/boundaries.txt (4); the switch / build variants it uses were taken out again)
# experiment: reads per wave of the step kernel (SP_K8_RPW: 4 = a quarter of the workgroups per launch, each wave takes four reads one after the other) with eight lanes in flight
mkdir -p gpurun_out/r06p
for r in 1 2 4 8 1 4; do
  SP_K8_RPW=$r python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --full-out gpurun_out/r06p/full.json > /dev/null 2> gpurun_out/r06p/err.txt
  python - <<PY
import json
d=json.load(open("gpurun_out/r06p/full.json"))
for k in ("cyp2d6","hla"):
    cp=d["critical_path"][k]
    print("rpw $r %s: value %.0f ms/step %.2f | chain_ms %.1f per_step %s boundary %s" % (k, d["value"], d["ms_per_step"], cp["chain_ms"], {a: round(v,1) for a,v in cp["per_step_us"].items()}, {a: round(v,1) for a,v in cp["boundary_us"].items()}))
PY
done
