# exact-occurrence placement: parity and speed
timeout 900 python -m pytest tests/test_gpu_consensus.py tests/test_gpu_cyp_real.py tests/test_gpu_cyp.py -x -q 2>&1 | tail -3
timeout 900 python profiles/scripts/k8fuzz.py 2>&1 | tail -2
timeout 600 python profiles/scripts/k8persist_dbg3.py "*1/*2" "*4+*68/*1" 2>&1 | grep -E "classic|persist" 
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_r04e.json 2> gpurun_out/bench_r04e.err
python -c "
import json;d=json.loads(open('gpurun_out/bench_r04e.json').read().strip().splitlines()[-1]);print(round(d['value']),d['ms_per_step'],d['kernel_ms']['cons_steps'],d['concordance']);print('cyp', d['cyp2d6']['value'], d['cyp2d6']['calls_equal_truth'], 'cohort', d['cohort']['value'], d['cohort']['ms'], d['cohort']['calls_equal_truth']); print(d.get('critical_path'))"
