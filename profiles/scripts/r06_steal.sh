# (an experiment of round 6 -- profiles/r06/boundaries.txt (7); the option it sets was taken out again: the capped launches were slower)
# step launches capped at the wave slots of the device, the waves drawing the reads beyond their own (context option k8_step_blocks, SP_K8_STEP_BLOCKS; 65535 = no cap: the launches as they were)
mkdir -p gpurun_out/r06w
for sb in 65535 0 256 65535 0; do
  SP_K8_STEP_BLOCKS=$sb python bench.py --steps 20 --warmup 5 --no-cpu-baseline --full-out gpurun_out/r06w/full.json > gpurun_out/r06w/line.json 2> gpurun_out/r06w/err.txt
  python - <<PY
import json
d=json.load(open("gpurun_out/r06w/full.json")); l=json.loads(open("gpurun_out/r06w/line.json").read().strip().splitlines()[-1])
h=d["critical_path"]["hla"]; c=d["critical_path"]["cyp2d6"]
print("step blocks $sb: value %.0f ms/step %.2f | hla chain %.1f ms per_step %s | cyp chain %.1f | hla alone resident %s | one lane %s | cohort %s shares %s" % (d["value"], d["ms_per_step"], h["chain_ms"], {a: round(v,1) for a,v in h["per_step_us"].items()}, c["chain_ms"], l["summary"]["hla_resident_reads_per_s"], l["summary"]["one_lane_reads_per_s"], l["cohort"]["samples_per_s"], l["cohort"]["share_rate_over_cohort_rate"]))
PY
done
