# the K9 jobs of one CYP2D6 sample (SP_K9_DEBUG=1 prints their sizes)
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
pkg = ge.load_package()
from pb_starphase_amd import synth
import cyp_cases_real as cr
ctx = pkg.Context(0)
cfg, gene_def = cr.load_db()
locus = synth.Chr22Locus(cfg, gene_def, seed=3)
cdb = pkg.ffi.CypDb(ctx, cfg, gene_def, locus.sequence, locus.start)
scen = cr.scenarios(locus)
R = ctx.upload(locus.sample(np.random.default_rng(5), scen[0][1], 2000))       # (the reads of the bench: 3-8 kb)
for _ in range(3):
    print("--- call", file=sys.stderr, flush=True)
    call, cons, labels = cdb.diplotype(R)
print(call.hap1, call.hap2, [len(c) for c in cons])
