timeout 900 python -m pytest tests/test_gpu_consensus.py tests/test_gpu_cyp_real.py tests/test_gpu_hla_pipeline.py -x -q -k "not stated_size" 2>&1 | grep -E "passed|failed"
timeout 900 python profiles/scripts/k8fuzz.py 2>&1 | tail -1
timeout 600 python profiles/scripts/k8persist_dbg3.py "*1/*2" "*4/*4" "*4+*68/*1" "*10+*36/*10" 2>&1 | grep -E "classic" | awk 'NR%2==0' | cut -c1-170
python bench.py --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['critical_path']; print('headline', round(d['value']), round(d['ms_per_step'],2), 'cyp', round(d['kernel_ms']['cyp2d6']['cons_steps'],2), round(c['cyp2d6']['dependent_steps'],1), {k:round(v,1) for k,v in c['cyp2d6']['per_step_us'].items()}, 'hla', round(d['kernel_ms']['hla']['cons_steps'],2), {k:round(v,1) for k,v in c['hla']['per_step_us'].items()})"
