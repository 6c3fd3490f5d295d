"""where sp_anchor_kernel's time goes in the CYP2D6 region search (anchor-timing build: build/variants/lib_at.so, -DSP_ANCHOR_TIMING): clocks of thread 0 per phase, per pair.
argv[1]: reads per sample (2000 = the headline's shape, reads of 3-8 kb; 100 with lo/hi 8000-16000 = the cohort's)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from pb_starphase_amd import synth
import cyp_cases_real as cr
ctx = pkg.Context(0)
cfg, gene_def = cr.load_db()
locus = synth.Chr22Locus(cfg, gene_def, seed=3)
db = pkg.ffi.CypDb(ctx, cfg, gene_def, locus.sequence, locus.start)
name, haps, expected = cr.scenarios(locus)[0]
for shape, n, kw in (("headline", 2000, {}), ("cohort", 1600, dict(lo=8000, hi=16000))):
    reads = locus.sample(np.random.default_rng(7), haps, n, **kw)
    R = ctx.upload(reads)
    db.diplotype(R)
    ctx.profile_reset(); ctx.synchronize()
    db.diplotype(R)
    c = [ctx.profile_get("count:dbg%d" % k)[2] for k in range(7)]
    pairs = max(1, c[4])
    print(shape, "reads", n, "mean read", round(np.mean([len(r) for r in reads])), "| anchor launches: pairs", c[4], "mean B", round(c[5] / pairs), "mean A", round(c[6] / pairs),
          "| 100 MHz or shader clocks per pair: clear+table", round(c[0] / pairs), "lookups+votes", round(c[1] / pairs), "barrier", round(c[2] / pairs), "peaks", round(c[3] / pairs),
          "| anchor ms", round(ctx.profile_get("anchor")[0], 3), "k3 cells ms", round(ctx.profile_get("k3_region_cells")[0], 3), "regions host ms", round(ctx.profile_get("host:cyp_regions")[0], 2))
