# anchor look-ups: buckets by the top 10 (default) / 12 / 13 / 14 bits of the code
for v in default am default am; do
  if [ $v = default ]; then unset SP_LIB_PATH; else export SP_LIB_PATH=$PWD/build/variants/lib_$v.so; fi
  python profiles/scripts/cyp_kernels.py 1 2>/dev/null | grep -E "total|anchor " | tr '\n' ' '; echo " [$v]"
  python bench.py --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   headline', round(d['value']), round(d['ms_per_step'],2), 'anchor_k1', round(d['kernel_ms']['hla']['anchor_k1'],3), 'anchor_k2', round(d['kernel_ms']['hla']['anchor_k2'],3), 'regions', round(d['host_wall_ms']['cyp2d6']['regions'],2), d['concordance']['cyp2d6_call_equals_truth'], d['concordance']['hla_diplotypes_equal_truth'])"
done
