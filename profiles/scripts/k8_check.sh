timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 900 python profiles/scripts/k8fuzz.py 2>&1 | tail -2
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_k8d.json 2> gpurun_out/bench_k8d.err
python -c "
import json;d=json.loads(open('gpurun_out/bench_k8d.json').read().strip().splitlines()[-1]);print(round(d['value']),d['ms_per_step'],d['kernel_ms']['cons_steps'],d['concordance']);print('cyp', d['cyp2d6']['value'], d['cyp2d6']['calls_equal_truth'], 'cohort', d['cohort']['value'], d['cohort']['ms'], d['cohort']['calls_equal_truth'], 'k5', d['k5_chain_pairs']['value'])"
python -c "
import json;d=json.loads(open('gpurun_out/bench_k8d.json').read().strip().splitlines()[-1]);print('inflight', d['samples_in_flight']); print(d['host_wall_ms'])"
