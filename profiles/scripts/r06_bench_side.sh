mkdir -p gpurun_out/r06b
for cfg in "0 0" "25 0" "25 1" "25 2"; do
  set -- $cfg
  SP_K8_COMPOUND=$1 SP_K8_SIDE_ORDERS=$2 python bench.py --steps 18 --warmup 4 --no-cpu-baseline --full-out gpurun_out/r06b/full_c$1_s$2.json > gpurun_out/r06b/line_c$1_s$2.json 2> gpurun_out/r06b/err_c$1_s$2.txt
  python - <<PY
import json
d=json.load(open("gpurun_out/r06b/full_c$1_s$2.json"))
cp=d["critical_path"]["cyp2d6"]
print("compound $1 side $2: value %.0f ms/step %.2f | chain steps %.0f chain_ms %.1f per_step %s | lanes %s | shares %s cohort %.0f | scen %s | hla lane %.1f" % (d["value"], d["ms_per_step"], cp["dependent_steps"], cp["chain_ms"], {k: round(v,1) for k,v in cp["per_step_us"].items()},
   {k:(round(v["value"]) if isinstance(v,dict) else v) for k,v in d["legs"]["cyp2d6_lanes"].items() if k in "124"}, {k: round(v["samples_per_s"]) for k,v in (d["legs"]["cohort"].get("by_share_size") or {}).items()}, d["legs"]["cohort"].get("samples_per_s"), {k: round(v["ms"],1) for k,v in d["legs"]["cyp2d6"]["scenarios"].items()}, d["host_wall_ms"]["lanes_hla_cyp2d6"][0]["work"]))
PY
done
