# what the re-scored numbers of K1 / K2 cost the headline step: alternating runs
for m in 1 0 1 0; do
SP_BENCH_MM2_RESCORE=$m python bench.py --no-cpu-baseline --no-extra-legs > gpurun_out/r04_rs_$m.json 2> gpurun_out/r04_rs_$m.err
python -c "
import json;d=json.loads(open('gpurun_out/r04_rs_$m.json').read().strip().splitlines()[-1]);print($m, round(d['value']),round(d['ms_per_step'],2),round(d['kernel_ms']['cyp2d6']['cons_steps'],2),round(d['kernel_ms']['hla']['cons_steps'],2),round(d['kernel_ms']['hla']['k1_cells'],2), {k:round(v,1) for k,v in d['host_wall_ms']['cyp2d6'].items()}, {k:round(v,1) for k,v in d['host_wall_ms']['hla'].items()})"
done
