# the headline with the persistent consensus kernels on the CYP2D6 context only, alternating with the default
for m in 0 1 0 1; do
SP_BENCH_CYP_PERSISTENT=$m python bench.py --no-cpu-baseline --no-extra-legs > gpurun_out/r04_cp_$m.json 2> gpurun_out/r04_cp_$m.err
python -c "
import json;d=json.loads(open('gpurun_out/r04_cp_$m.json').read().strip().splitlines()[-1]);print($m, round(d['value']),round(d['ms_per_step'],2),round(d['kernel_ms']['cyp2d6']['cons_steps'],2),round(d['kernel_ms']['hla']['cons_steps'],2),round(d['kernel_ms']['hla']['k1_cells'],2), d['host_wall_ms']['cyp2d6'])"
done
