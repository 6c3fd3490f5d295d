mkdir -p gpurun_out/r06b
for so in 0 1 3; do
  SP_K8_SIDE_ORDERS=$so python bench.py --steps 18 --warmup 4 --no-cpu-baseline --full-out gpurun_out/r06b/full_so$so.json > gpurun_out/r06b/line_so$so.json 2> gpurun_out/r06b/err_so$so.txt
  python - <<PY
import json
d=json.load(open("gpurun_out/r06b/full_so$so.json"))
cp=d["critical_path"]["cyp2d6"]
print("side orders $so: value %.0f ms/step %.2f | chain steps %.0f chain_ms %.1f per_step %s | lanes %s | cohort %s" % (d["value"], d["ms_per_step"], cp["dependent_steps"], cp["chain_ms"], {k: round(v,1) for k,v in cp["per_step_us"].items()},
   {k:(round(v["value"]) if isinstance(v,dict) else v) for k,v in d["legs"]["cyp2d6_lanes"].items() if k in "124"}, {k: round(v["samples_per_s"]) for k,v in (d["legs"]["cohort"].get("by_share_size") or {}).items()}), d["legs"]["cohort"].get("samples_per_s"), {k: round(v["ms"],1) for k,v in d["legs"]["cyp2d6"]["scenarios"].items()})
PY
done
