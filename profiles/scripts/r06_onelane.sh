for hp in auto 0 1; do for lanes in 1 2; do
  if [ $hp = auto ]; then unset SP_BENCH_HEADLINE_PERSISTENT; else export SP_BENCH_HEADLINE_PERSISTENT=$hp; fi
  python bench.py --steps 18 --warmup 3 --no-cpu-baseline --no-extra-legs --cyp-lanes $lanes --full-out gpurun_out/r06e/full_one.json > /dev/null 2> gpurun_out/r06e/err.txt
  python - <<PY
import json
d=json.load(open("gpurun_out/r06e/full_one.json"))
cp=d["critical_path"]["cyp2d6"]
print("persistent $hp lanes $lanes: value %.0f ms/step %.2f | mode %s | chain steps %.0f chain_ms %.1f per_step %s | cyp host %s" % (d["value"], d["ms_per_step"], d["config"]["cyp2d6_consensus"][:24], cp["dependent_steps"], cp["chain_ms"], {k: round(v,1) for k,v in cp.get("per_step_us",{}).items()}, {k: (round(v,1) if not isinstance(v, dict) else "") for k,v in d["host_wall_ms"]["cyp2d6"].items()}))
PY
done; done
