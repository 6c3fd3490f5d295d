# first window of a child that has lookahead votes from its expansion: 32 columns (default) / 16 / 64
for v in default rp4 rp8 default rp4; do
  if [ $v = default ]; then unset SP_LIB_PATH; else export SP_LIB_PATH=$PWD/build/variants/lib_$v.so; fi
  echo "== $v"
  timeout 600 python profiles/scripts/k8persist_dbg3.py "*1/*2" "*4/*4" "*4+*68/*1" "*10+*36/*10" 2>&1 | grep -E "classic" | awk 'NR%2==0' | cut -c1-150
  python bench.py --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['critical_path']; print('headline', round(d['value']), round(d['ms_per_step'],2), 'cyp', round(d['kernel_ms']['cyp2d6']['cons_steps'],2), round(c['cyp2d6']['dependent_steps'],1), {k:round(v,1) for k,v in c['cyp2d6']['per_step_us'].items()}, 'hla', round(d['kernel_ms']['hla']['cons_steps'],2), round(c['hla']['dependent_steps'],1))"
done
