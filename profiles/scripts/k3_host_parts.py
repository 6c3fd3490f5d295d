"""where sp_cyp_find_regions' wall time goes (host marks), a 2,000-read sample alone on the device"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from pb_starphase_amd import synth
import cyp_cases_real as cr
cfg, gene_def = cr.load_db(); locus = synth.Chr22Locus(cfg, gene_def, seed=3)
ctx = pkg.Context(0); db = pkg.ffi.CypDb(ctx, cfg, gene_def, locus.sequence, locus.start)
tm = db.templates(); T = ctx.upload([t[3] for t in tm]); ttype = np.array([t[0] for t in tm], np.int32)
for name, haps, exp in cr.scenarios(locus)[:2]:
    reads = locus.sample(np.random.default_rng(7), haps, 2000); R = ctx.upload(reads)
    ctx.cyp_find_regions(T, ttype, R, 0.5)
    ctx.profile_reset(); ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        h = ctx.cyp_find_regions(T, ttype, R, 0.5)
    dt = (time.perf_counter() - t0) / 3
    print(name, "regions %.2f ms" % (1e3 * dt), {k: round(ctx.profile_get("host:k3_" + k)[0] / 3, 2) for k in ("cells", "retry", "list", "mark", "crit_rescore", "collapse", "rest_rescore")},
          {k: round(ctx.profile_get(k)[0] / 3, 2) for k in ("anchor", "k3_region_cells", "k3_af_crit_trace", "k3_af_crit_dp")})
