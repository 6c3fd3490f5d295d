timeout 1500 python -m pytest tests/test_gpu_cyp.py tests/test_gpu_cyp_pipeline.py tests/test_gpu_cyp_real.py tests/test_gpu_cohort_rank.py -x -q -k "not stated_size" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_concordance.py -x -q -s 2>&1 | grep -E "K4|passed|failed" | head -12
python profiles/scripts/cyp_kernels.py 0 | grep -E "total|k4|weights|merge"
python bench.py --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],2), {k:round(v,1) for k,v in d['host_wall_ms']['cyp2d6'].items()}, d['concordance'])"
