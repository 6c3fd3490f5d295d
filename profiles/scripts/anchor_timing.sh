# anchor kernel, -DSP_ANCHOR_TIMING build: shader clocks of thread 0 per phase, summed over all pairs of 3 steps
# (dbg0 clear + table, dbg1 look-ups + votes, dbg2 barrier wait, dbg3 peak rounds, dbg4 pairs, dbg5 sum of B lengths, dbg6 sum of A lengths)
SP_BENCH_DBG=1 SP_LIB_PATH=$PWD/build/variants/lib_at.so python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs > gpurun_out/bat.json 2> gpurun_out/bat.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bat.json').read().strip().splitlines()[-1])
c = d["dbg_counters"]; n = max(1, c[4])
print("pairs", c[4], "mean B", c[5] / n, "mean A", c[6] / n)
for name, v in zip(("clear+table", "lookups+votes", "barrier", "peaks"), c[:4]):
    print(f"{name:14s} {v / n:9.0f} clocks per pair")
print({k: round(v, 3) for k, v in d["kernel_ms"].items() if "anchor" in k})
PY
