# hardware queues of the process (GPU_MAX_HW_QUEUES; bench.py asks for 16) with eight lanes in flight: do the lanes' streams share queues?
mkdir -p gpurun_out/r06n
for q in 16 32 64 8 16 32; do
  GPU_MAX_HW_QUEUES=$q python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --full-out gpurun_out/r06n/full.json > /dev/null 2> gpurun_out/r06n/err.txt
  python - <<PY
import json
d=json.load(open("gpurun_out/r06n/full.json"))
cp=d["critical_path"]["cyp2d6"]
print("queues $q: value %.0f ms/step %.2f | cyp chain_ms %.1f per_step %s | lanes work %s" % (d["value"], d["ms_per_step"], cp["chain_ms"], {k: round(v,1) for k,v in cp["per_step_us"].items()}, [round(x["work"],1) for x in d["host_wall_ms"]["lanes_hla_cyp2d6"]]))
PY
done
