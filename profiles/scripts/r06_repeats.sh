# the driver's command three times in a row on one box: size of the line, value, ms per step, roofline.frac, cpu_baseline, cohort shares, one-lane leg, seconds of the whole run
mkdir -p gpurun_out/r06m
for i in 1 2 3; do
  t0=$(date +%s)
  python bench.py --gpus 1 --steps 20 --warmup 5 --full-out gpurun_out/r06m/full$i.json > gpurun_out/r06m/line$i.json 2> gpurun_out/r06m/err$i.txt
  rc=$?
  t1=$(date +%s)
  python - <<PY
import json
l=[x for x in open("gpurun_out/r06m/line$i.json").read().splitlines() if x.startswith("{")]
d=json.loads(l[-1])
print("run $i rc $rc, %d s: line %d bytes, value %.0f, ms/step %.2f, roofline.frac %s, cpu_baseline %s, shares %s, one lane %s" % ($t1 - $t0, len(l[-1]), d["value"], d["ms_per_step"], d["roofline"]["frac"], d["cpu_baseline"]["value"], d["cohort"]["share_rate_over_cohort_rate"], d["summary"]["one_lane_reads_per_s"]))
PY
done
