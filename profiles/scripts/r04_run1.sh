timeout 1200 python -m pytest tests/test_gpu_concordance.py -q -s 2>&1 | tail -40 > gpurun_out/r04_concord.log
for v in 0 1 2 3 4 10 11 13 14; do SP_MB_COPY_VARIANT=$v python - <<'PY' >> gpurun_out/r04_copy_variants.txt 2>&1
import os, sys
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge
pkg = ge.load_package(); ctx = pkg.Context(0)
print("variant", os.environ["SP_MB_COPY_VARIANT"], "hbm_copy GB/s", round(ctx.microbench("hbm_copy") / 1e9, 1), ctx.info())
PY
done
timeout 2400 python bench.py > gpurun_out/r04_bench_c.json 2> gpurun_out/r04_bench_c.err
tail -30 gpurun_out/r04_concord.log; cat gpurun_out/r04_copy_variants.txt; tail -3 gpurun_out/r04_bench_c.err
