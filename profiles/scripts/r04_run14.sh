timeout 900 python -m pytest tests/test_gpu_consensus.py -x -q 2>&1 | tail -3
timeout 600 python profiles/scripts/k8persist_dbg3.py "*1/*2" "*4+*68/*1" 2>&1 | grep -E "classic|persist|rror"
for m in 0 1 0 1; do
SP_BENCH_CYP_PERSISTENT=$m python bench.py --no-cpu-baseline --no-extra-legs > gpurun_out/r04_cp_$m.json 2> gpurun_out/r04_cp_$m.err
python -c "
import json;d=json.loads(open('gpurun_out/r04_cp_$m.json').read().strip().splitlines()[-1]);print($m, round(d['value']),round(d['ms_per_step'],2),round(d['kernel_ms']['cyp2d6']['cons_steps'],2),round(d['kernel_ms']['hla']['cons_steps'],2),round(d['kernel_ms']['hla']['k1_cells'],2), {k:round(v,1) for k,v in d['host_wall_ms']['cyp2d6'].items()}, d['concordance'])"
done
timeout 600 python profiles/scripts/cyp_share_probe.py 32 2>&1 | cut -c1-200
