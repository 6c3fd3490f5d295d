"""How many of a CYP2D6 sample's consensus launches are for a node that stood idle at the end of its tape when the launch before was ordered -- the steps a launch
carrying several work orders per problem could save.  Needs a library built with -DSP_K8_PF_PROBE (profiles/scripts/k8_prefetch_probe.sh).
Run on the GPU box:  python profiles/scripts/k8_prefetch_probe.py [n_reads]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge

pkg = ge.load_package()
from pb_starphase_amd import synth
import cyp_cases_real as cr

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
cfg, gene_def = cr.load_db()
locus = synth.Chr22Locus(cfg, gene_def, seed=3)
ctx = pkg.Context(0)
ctx.set_option("k8_persistent", 0)
db = pkg.ffi.CypDb(ctx, cfg, gene_def, locus.sequence, locus.start)
names = ("cons_windows", "cons_all_windows", "cons_expansions", "cons_cut_windows", "cons_pf_win", "cons_pf_replay", "cons_pf_exp", "cons_pf_chain", "cons_pf_chain_win", "cons_pf_chain_all")
for name, haps, expected in cr.scenarios(locus):
    reads = locus.sample(np.random.default_rng(7), haps, n)
    R = ctx.upload(reads)
    db.diplotype(R)
    ctx.profile_reset(); ctx.synchronize(); t0 = time.perf_counter()
    call, cons, labels = db.diplotype(R)
    dt = time.perf_counter() - t0
    got = sorted([call.hap1.decode(), call.hap2.decode()])
    print(f"{name:12s} {dt * 1e3:7.1f} ms  cons {ctx.profile_get('cons_steps')[0]:7.1f} ms " + " ".join(f"{k[5:]}={ctx.profile_get(k)[2]}" for k in names) + ("" if got == sorted(expected) else "  CALL != TRUTH"), flush=True)
# the HLA sample (configs[1], 2 x 5,000 reads): linear searches
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
