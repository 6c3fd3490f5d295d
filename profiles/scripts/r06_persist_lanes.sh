# the headline's stream with the CYP2D6 contexts' consensus forced to persistent kernels (SP_BENCH_HEADLINE_PERSISTENT=1), by lanes in flight
mkdir -p gpurun_out/r06v
for cfg in "0 2 6" "1 2 6" "1 2 4" "1 2 3" "1 2 2"; do
  set -- $cfg
  SP_BENCH_HEADLINE_PERSISTENT=$1 timeout 300 python bench.py --steps 20 --warmup 5 --hla-lanes $2 --cyp-lanes $3 --no-cpu-baseline --no-extra-legs --full-out gpurun_out/r06v/full.json > /dev/null 2> gpurun_out/r06v/err.txt
  echo "rc $?"; tail -2 gpurun_out/r06v/err.txt | cut -c1-300
  python - <<PY
import json
d=json.load(open("gpurun_out/r06v/full.json"))
cp=d["critical_path"]["cyp2d6"]
print("persistent $1 lanes $2+$3: value %.0f ms/step %.2f | %s | steps %.0f chain_ms %.1f per_step %s | lanes %s" % (d["value"], d["ms_per_step"], cp["mode"][:30], cp["dependent_steps"], cp["chain_ms"], {a: round(v,1) for a,v in cp.get("per_step_us",{}).items()}, [round(x["work"],1) for x in d["host_wall_ms"]["lanes_hla_cyp2d6"]]))
PY
done
