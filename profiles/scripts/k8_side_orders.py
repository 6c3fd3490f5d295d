"""The consensus launch chains of the six configs[2] scenarios and the HLA sample: dependent steps, side orders made, expansions adopted without a launch.
Run on the GPU box:  SP_K8_SIDE_ORDERS=0..3 python profiles/scripts/k8_side_orders.py [n_reads]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge

pkg = ge.load_package()
from pb_starphase_amd import synth
import cyp_cases_real as cr

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
only = sys.argv[2:]


def path(ctx):
    """the chain of the slowest problem of every batch: dependent steps, us per step in the step kernel / the control kernel (its four parts) / between the kernels"""
    st = max(1, ctx.profile_get("cons_path_steps")[2])
    us = lambda k: ctx.profile_get(k)[2] / 100.0 / st
    return "path steps=%d step=%.1f ctl=%.1f (load %.1f result %.1f search %.1f tail %.1f) gap=%.1f us" % (st, us("cons_path_step_ticks"), us("cons_path_control_ticks"), us("cons_ticks_reduce"),
                                                                                                   us("cons_ticks_result"), us("cons_ticks_search"), us("cons_ticks_tail"), us("cons_path_gap_ticks"))
cfg, gene_def = cr.load_db()
locus = synth.Chr22Locus(cfg, gene_def, seed=3)
ctx = pkg.Context(0)
ctx.set_option("k8_persistent", int(os.environ.get("K8P", "0")))
db = pkg.ffi.CypDb(ctx, cfg, gene_def, locus.sequence, locus.start)
names = ("cons_windows", "cons_expansions", "cons_cut_windows", "cons_side_windows", "cons_side_expansions", "cons_adopted", "cons_compound", "cons_compound_ok", "cons_columns")
for name, haps, expected in cr.scenarios(locus):
    if only and name not in only:
        continue
    reads = locus.sample(np.random.default_rng(7), haps, n)
    R = ctx.upload(reads)
    db.diplotype(R)
    best = None
    for _ in range(3):
        ctx.profile_reset(); ctx.synchronize(); t0 = time.perf_counter()
        call, cons, labels = db.diplotype(R)
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, ctx.profile_get('cons_steps')[0], [ctx.profile_get(k)[2] for k in names], path(ctx))
    got = sorted([call.hap1.decode(), call.hap2.decode()])
    print(f"{name:12s} {best[0] * 1e3:7.1f} ms  cons {best[1]:7.1f} ms " + " ".join(f"{k[5:]}={v}" for k, v in zip(names, best[2])) + ("" if got == sorted(expected) else "  CALL != TRUTH") + "\n             " + best[3], flush=True)
if only and "HLA" not in only:
    sys.exit(0)
fx = synth.HlaFixture()
wl = synth.Config2Workload(fx, n_reads=10000, seed=1000)
hdb = fx.make_db(pkg, ctx)
R = ctx.upload(wl.reads)
o = hdb.realign_reads(R)
hdb.diplotype_genes(list(range(len(fx.genes))), R, o)
ctx.profile_reset(); ctx.synchronize(); t0 = time.perf_counter()
hdb.diplotype_genes(list(range(len(fx.genes))), R, o)
dt = time.perf_counter() - t0
print(f"{'HLA-A/-B':12s} {dt * 1e3:7.1f} ms  cons {ctx.profile_get('cons_steps')[0]:7.1f} ms " + " ".join(f"{k[5:]}={ctx.profile_get(k)[2]}" for k in names), flush=True)
