# the whole bench line with the CYP2D6 contexts' consensus as persistent kernels (1) and as launch pairs (0), alternating on one box
for m in 0 1 0 1 0 1; do
SP_BENCH_CYP_PERSISTENT=$m python bench.py --no-cpu-baseline > gpurun_out/r04_full_$m.json 2> gpurun_out/r04_full_$m.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r04_full_$m.json').read().strip().splitlines()[-1])
L=d['legs']
print($m, 'headline', round(d['value']), round(d['ms_per_step'],2), 'cyp cons', round(d['kernel_ms']['cyp2d6']['cons_steps'],2), 'hla cons', round(d['kernel_ms']['hla']['cons_steps'],2), 'cyp stages', {k:round(v,1) for k,v in d['host_wall_ms']['cyp2d6'].items() if v > 0.5}, 'hla', round(d['host_wall_ms']['hla']['k1_total'],1), round(d['host_wall_ms']['hla']['hla_genes_total'],1),
      '| resident', round(L['hla_resident']['value']), 'in flight', round(L['samples_in_flight']['value']), 'cyp leg', round(L['cyp2d6']['value']), 'cohort', round(L['cohort']['samples_per_s'],1), {k:round(v['samples_per_s']) for k,v in L['cohort']['by_share_size'].items()})
PY
done
