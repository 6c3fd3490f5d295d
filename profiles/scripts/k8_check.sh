
timeout 900 python -m pytest tests/test_gpu_consensus.py tests/test_gpu_hla_pipeline.py tests/test_gpu_cyp_pipeline.py -x -q 2>&1 | tail -3
timeout 900 python profiles/scripts/k8fuzz.py 2>&1 | tail -2
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs > gpurun_out/bench_k8d.json 2> gpurun_out/bench_k8d.err
python -c "
import json;d=json.loads(open('gpurun_out/bench_k8d.json').read().strip().splitlines()[-1]);print(round(d['value']),d['ms_per_step'],d['kernel_ms']['cons_steps'],d['concordance'],d['consensus'])"
bash profiles/scripts/k8_gaps.sh
