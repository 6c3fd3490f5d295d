# (an experiment of round 6 -- profiles/r06/boundaries.txt (4); the switch / build variants it uses were taken out again)
# experiment: seeded K1 in smaller batches (SP_K1S_BATCH_MB of anchors per batch; 256 = the default: three batches per 10,000 reads) -- shorter kernels, shorter stalls of the chains behind them?
mkdir -p gpurun_out/r06r
for mb in 256 64 32 128 256 64; do
  SP_K1S_BATCH_MB=$mb python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --full-out gpurun_out/r06r/full.json > /dev/null 2> gpurun_out/r06r/err.txt
  python - <<PY
import json
d=json.load(open("gpurun_out/r06r/full.json"))
cp=d["critical_path"]["cyp2d6"]
print("K1 batch $mb MB: value %.0f ms/step %.2f | cyp chain_ms %.1f per_step %s boundary %s | k1_total %.1f" % (d["value"], d["ms_per_step"], cp["chain_ms"], {a: round(v,1) for a,v in cp["per_step_us"].items()}, {a: round(v,1) for a,v in cp["boundary_us"].items()}, d["host_wall_ms"]["hla"]["k1_total"]))
PY
done
