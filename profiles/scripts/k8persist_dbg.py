import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from pb_starphase_amd import synth
import cyp_cases_real as cr
ctx = pkg.Context(0)
print(ctx.info())
rng = np.random.default_rng(1)
base = "".join(rng.choice(list("ACGT"), 1500))
for n in (8, 64, 300, 1000, 2000):
    reads = [synth.hifi_errors(rng, base) for _ in range(n)]
    ctx.profile_reset()
    t0 = time.time()
    try:
        out = ctx.consensus(ctx.upload(reads), pkg.ffi.sp_cons_config(3, 100, 1, 1, 400, 50, 0.10, 20, 10, 1000, 0))
        print(n, "reads ok", out["cons"][0] == base, "persistent batches", ctx.profile_get("cons_persistent_batches")[2], "steps", ctx.profile_get("cons_windows")[2], round(time.time() - t0, 3), flush=True)
    except Exception as e:
        print(n, "reads FAILED", e, round(time.time() - t0, 3), flush=True)
cfg, gene_def = cr.load_db()
locus = synth.Chr22Locus(cfg, gene_def, seed=3)
db = pkg.ffi.CypDb(ctx, cfg, gene_def, locus.sequence, locus.start)
sc = {n: (h, e) for n, h, e in cr.scenarios(locus)}
for nr in (100, 400, 1000, 2000):
    reads = locus.sample(np.random.default_rng(7), sc["*1/*2"][0], nr)
    ctx.profile_reset(); t0 = time.time()
    try:
        call, cons, labels = db.diplotype(ctx.upload(reads))
        print("cyp", nr, call.hap1, call.hap2, "persistent batches", ctx.profile_get("cons_persistent_batches")[2], "cons ms", round(ctx.profile_get("cons_steps")[0], 2), round(time.time() - t0, 3), flush=True)
    except Exception as e:
        print("cyp", nr, "FAILED", e, round(time.time() - t0, 3), flush=True)
