timeout 1500 python -m pytest tests/test_gpu_cyp.py tests/test_gpu_cyp_real.py tests/test_gpu_concordance.py -x -q -m gpu -s 2>&1 | grep -E "K3 reads|passed|failed|Error|assert" | head -30
python - <<'PY'
import sys, os, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from pb_starphase_amd import synth
import cyp_cases_real as cr
cfg, gene_def = cr.load_db(); locus = synth.Chr22Locus(cfg, gene_def, seed=3)
ctx = pkg.Context(0); db = pkg.ffi.CypDb(ctx, cfg, gene_def, locus.sequence, locus.start)
tm = db.templates(); T = ctx.upload([t[3] for t in tm]); ttype = np.array([t[0] for t in tm], np.int32)
for name, haps, exp in cr.scenarios(locus):
    reads = locus.sample(np.random.default_rng(7), haps, 2000); R = ctx.upload(reads)
    ctx.cyp_find_regions(T, ttype, R, 0.5)
    ctx.profile_reset(); ctx.synchronize(); t0 = time.perf_counter()
    h = ctx.cyp_find_regions(T, ttype, R, 0.5)
    dt = time.perf_counter() - t0
    print(name, "regions %.2f ms, hits %d, critical placements %d, kernel ms:" % (1e3 * dt, len(h), ctx.profile_get("k3_critical_placements")[2]), {k: round(ctx.profile_get(k)[0], 2) for k in ("k3_region_cells", "anchor", "k3_af_crit_trace", "k3_af_crit_dp", "k3_af_trace", "k3_af_dp")})
PY
