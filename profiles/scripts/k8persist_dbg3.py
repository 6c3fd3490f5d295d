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
sc = {n: (h, e) for n, h, e in cr.scenarios(locus)}
names = sys.argv[1:] or ["*1/*2"]
for name in names:
    reads = locus.sample(np.random.default_rng(7), sc[name][0], 2000)
    R = ctx.upload(reads)
    for mode in (1, 0, 1, 0):
        ctx.set_option("k8_persistent", mode)
        ctx.profile_reset(); ctx.synchronize(); t0 = time.time()
        call, cons, labels = db.diplotype(R)
        dt = time.time() - t0
        g = lambda n: ctx.profile_get(n)[2]
        steps = g("cons_path_steps")
        print(name, "persistent" if mode else "classic   ", call.hap1.decode(), call.hap2.decode(), "total ms", round(1e3 * dt, 1), "cons ms", round(ctx.profile_get("cons_steps")[0], 2),
              "batches", g("cons_persistent_batches"), "path steps", steps, "step us", round(g("cons_path_step_ticks") / 100 / max(1, steps), 1),
              "control us", round(g("cons_path_control_ticks") / 100 / max(1, steps), 1), "gap us/step", round(g("cons_path_gap_ticks") / 100 / max(1, steps), 1),
              "host loop ms", round(ctx.profile_get("host:k8_loop")[0], 2), "result wait ms", round(ctx.profile_get("host:k8_result_wait")[0], 2), flush=True)
