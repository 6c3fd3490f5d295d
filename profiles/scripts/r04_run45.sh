for st in 8 8 8 8; do
SP_BENCH_CYP_PERSISTENT=1 SP_BENCH_CYP_STREAMS=$st python bench.py --workload cohort --steps 3 --warmup 1 > gpurun_out/r04_q.json 2> gpurun_out/r04_q.err; rc=$?
python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r04_q.json').read().strip().splitlines()[-1]); print($st, 'rc', $rc, round(d['cohort']['samples_per_s'],1))
except Exception as e: print($st, 'rc', $rc, 'failed:', open('gpurun_out/r04_q.err').read().strip().splitlines()[-1][:160])
PY
done
