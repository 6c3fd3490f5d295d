# K1 at 8 / 6 / 5 waves per SIMD (amdgpu_waves_per_eu; the default build: 7 by its 72 registers)
for v in default occ8 default occ8 occ6 occ5; do
  if [ $v = default ]; then unset SP_LIB_PATH; else export SP_LIB_PATH=$PWD/build/variants/lib_$v.so; fi
  python bench.py --no-cpu-baseline --no-extra-legs > gpurun_out/k1occ.json 2> gpurun_out/k1occ.err
  python - "$v" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/k1occ.json').read().strip().splitlines()[-1])
print(sys.argv[1], 'reads/s', round(d['value']), 'ms', round(d['ms_per_step'], 2), 'k1_cells', round(d['kernel_ms']['hla']['k1_cells'], 3), 'deep', round(d['kernel_ms']['hla']['k1_cells_deep'], 3), d['concordance']['hla_diplotypes_equal_truth'], d['concordance']['k1_gene_correct'])
PY
done
