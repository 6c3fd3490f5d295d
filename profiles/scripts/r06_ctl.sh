# (an experiment of round 6 -- profiles/r06/lanes_matrix.txt (d): no effect; the switch it sets was taken out of the library afterwards)
mkdir -p gpurun_out/r06e
for th in 1024 256 512 1024 256; do
  SP_K8_CTL_THREADS=$th python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-extra-legs --full-out gpurun_out/r06e/full_t$th.json > /dev/null 2> gpurun_out/r06e/err.txt
  python - <<PY
import json
d=json.load(open("gpurun_out/r06e/full_t$th.json"))
cp=d["critical_path"]["cyp2d6"]; ch=d["critical_path"]["hla"]
print("ctl threads $th: value %.0f ms/step %.2f | cyp chain steps %.0f chain_ms %.1f per_step %s | hla chain_ms %.1f per_step %s | lanes work %s" % (d["value"], d["ms_per_step"], cp["dependent_steps"], cp["chain_ms"], {k: round(v,1) for k,v in cp["per_step_us"].items()},
   ch["chain_ms"], {k: round(v,1) for k,v in ch.get("per_step_us", {}).items()}, [round(x["work"],1) for x in d["host_wall_ms"]["lanes_hla_cyp2d6"]]))
PY
done
