mkdir -p gpurun_out/r06e
for cfg in "0 0" "1 0" "0 1" "1 1" "0 0"; do
  set -- $cfg
  SP_K8_CTL_PRIO=$1 SP_K8_CTL_LDS_PAD=$2 python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-extra-legs --full-out gpurun_out/r06e/full_p$1_l$2.json > /dev/null 2> gpurun_out/r06e/err.txt
  python - <<PY
import json
d=json.load(open("gpurun_out/r06e/full_p$1_l$2.json"))
cp=d["critical_path"]["cyp2d6"]; ch=d["critical_path"]["hla"]
print("ctl prio $1 lds pad $2: value %.0f ms/step %.2f | cyp chain steps %.0f chain_ms %.1f per_step %s | hla chain_ms %.1f per_step %s | lanes work %s" % (d["value"], d["ms_per_step"], cp["dependent_steps"], cp["chain_ms"], {k: round(v,1) for k,v in cp["per_step_us"].items()},
   ch["chain_ms"], {k: round(v,1) for k,v in ch.get("per_step_us", {}).items()}, [round(x["work"],1) for x in d["host_wall_ms"]["lanes_hla_cyp2d6"]]))
PY
done
