timeout 900 python -m pytest tests/test_gpu_hla.py tests/test_gpu_concordance.py -q -s 2>&1 | grep -E "K1 |passed|failed|Error|assert" | head -20
python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from pb_starphase_amd import synth
ctx = pkg.Context(0)
fx = synth.HlaFixture(); db = fx.make_db(pkg, ctx)
wl = synth.Config2Workload(fx, n_reads=10000, seed=1000)
R = ctx.upload(wl.reads)
for m in (1, 0, 1, 0):
    ctx.set_option("mm2_rescore", m)
    db.realign_reads(R); ctx.profile_reset(); ctx.synchronize(); t0 = time.time()
    for _ in range(5): db.realign_reads(R)
    ctx.synchronize(); dt = (time.time() - t0) / 5
    print("mm2_rescore", m, "K1 ms", round(1e3 * dt, 2), "trace ms", round(ctx.profile_get("k1_af_trace")[0] / 5, 2), "dp ms", round(ctx.profile_get("k1_af_dp")[0] / 5, 2), "dp pairs", ctx.profile_get("k1_af_dp_pairs")[2] / 5, "of", ctx.profile_get("k1_af_pairs")[2] / 5)
PY
