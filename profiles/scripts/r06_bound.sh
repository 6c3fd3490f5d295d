mkdir -p gpurun_out/r06o
for cfg in "2 6" "1 1" "2 6"; do
  set -- $cfg
  python bench.py --steps 20 --warmup 5 --hla-lanes $1 --cyp-lanes $2 --no-cpu-baseline --no-extra-legs --full-out gpurun_out/r06o/full.json > /dev/null 2> gpurun_out/r06o/err.txt
  python - <<PY
import json
d=json.load(open("gpurun_out/r06o/full.json"))
for k in ("cyp2d6","hla"):
    cp=d["critical_path"][k]
    print("lanes $1+$2 %s: value %.0f | steps %.0f chain_ms %.1f per_step %s boundary %s" % (k, d["value"], cp["dependent_steps"], cp["chain_ms"], {a: round(v,1) for a,v in cp["per_step_us"].items()}, {a: round(v,1) for a,v in cp["boundary_us"].items()}))
PY
done
