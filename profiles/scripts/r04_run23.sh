# the headline with the two HLA genes on one stream (0) or two (1): three dependent chains at once or two
for m in 1 0 1 0 1 0; do
SP_BENCH_HLA_SPLIT=$m python bench.py --no-cpu-baseline --no-extra-legs > gpurun_out/r04_hs.json 2> gpurun_out/r04_hs.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r04_hs.json').read().strip().splitlines()[-1]); c=d['critical_path']
print($m, 'headline', round(d['value']), round(d['ms_per_step'],2), 'cyp cons', round(d['kernel_ms']['cyp2d6']['cons_steps'],2), {k:round(v,1) for k,v in c['cyp2d6']['per_step_us'].items()}, 'cyp stages', {k:round(v,1) for k,v in d['host_wall_ms']['cyp2d6'].items() if v > 0.5}, 'hla', round(d['host_wall_ms']['hla']['k1_total'],1), round(d['host_wall_ms']['hla']['hla_genes_total'],1), d['concordance']['hla_diplotypes_equal_truth'])
PY
done
