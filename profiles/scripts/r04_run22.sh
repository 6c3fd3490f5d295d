for m in 6 8 6 8 6 8; do
SP_BENCH_CYP_STREAMS=$m python bench.py --workload cohort --steps 3 --warmup 1 > gpurun_out/r04_cs.json 2> gpurun_out/r04_cs.err
python -c "
import json;d=json.loads(open('gpurun_out/r04_cs.json').read().strip().splitlines()[-1]);c=d['cohort'];print($m, round(c['samples_per_s'],1), round(c['ms_per_step'],1), {k:round(v,3) for k,v in c['rank0_host_seconds_per_pass'].items()}, c['calls_equal_truth']['cyp2d6'])"
done
