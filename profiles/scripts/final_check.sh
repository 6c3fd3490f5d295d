# the round's closing run: GPU suite, the default bench line, the cohort at the share sizes of 8 / 4 / 2 ranks
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
python bench.py > gpurun_out/bench_r03_final.json 2> gpurun_out/bench_r03_final.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_r03_final.json").read().strip().splitlines()[-1])
print(round(d["value"]), round(d["ms_per_step"], 2), d["stale_counter_files"], round(d["roofline"]["frac"], 3), round(d["roofline_valu"]["frac"], 3), round(d["cpu_baseline"]["value"], 1), d["cpu_baseline"]["diplotypes_identical"])
print({k: (round(v.get("value", 0)), v.get("samples_per_s"), v.get("ms_per_step")) for k, v in d["legs"].items()})
print(d["legs"]["cohort"]["rank0_host_seconds_per_pass"], d["legs"]["cohort"]["calls_equal_truth"])
PY
for n in 32 64 128; do
  python bench.py --workload cohort --cohort-samples $n --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['cohort']; print(c['samples'], round(c['samples_per_s'],1), round(c['ms_per_step'],1))"
done
