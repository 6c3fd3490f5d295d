"""the configs[2] hybrid sample on a second random locus (seed 2003): library vs the oracle pipeline at a given number of reads (DESIGN.md section 9:
at 2,000 reads the bounded search of the four-class mixture gives up, in the oracle exactly as in the library).  usage: cyp_other_locus.py <reads>"""
import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from pb_starphase_amd import synth
import oracle_ffi as of, cyp_cases_real as cr, cyp_pipeline as cp
oracle = of.load(); ctx = pkg.Context(0)
cfg, gene_def = cr.load_db()
locus = synth.Chr22Locus(cfg, gene_def, seed=2003)
db = pkg.ffi.CypDb(ctx, cfg, gene_def, locus.sequence, locus.start)
tm, vs = db.templates(), db.variants(); names, rows = db.alleles(); bb = cfg["cyp_coordinates"]["CYP2D6_wfa_backbone"]
odb = cp.Db([t[2] for t in tm], [t[0] for t in tm], [t[1] for t in tm], [t[3] for t in tm], [t[4] for t in tm], locus.slice(bb["start"], bb["end"]),
            [(p - bb["start"], r, a) for p, r, a, _l, _v in vs], [v[4] for v in vs], names, rows, var_labels=[v[3] for v in vs])
name, haps, expected = [s for s in cr.scenarios(locus) if s[0] == "*4+*68/*1"][0]
reads = locus.sample(np.random.default_rng(2007), haps, 2000)
call, cons, labels = db.diplotype(ctx.upload(reads))
print("library:", call.status, call.hap1.decode(), "/", call.hap2.decode(), "expected", expected)
print("labels", labels, "chains", list(call.chain1[:call.n1]), list(call.chain2[:call.n2]))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
sub = locus.sample(np.random.default_rng(2007), haps, n)
t0 = time.time()
exp = cp.diplotype(oracle, odb, sub, cfg=db.cfg)
call2, cons2, labels2 = db.diplotype(ctx.upload(sub))
print(f"{n} reads: oracle pipeline {time.time()-t0:.0f} s:", exp["hap1"], "/", exp["hap2"], "| library", call2.hap1.decode(), "/", call2.hap2.decode(), "| consensus equal", cons2 == exp["consensus"], "labels equal", labels2 == [(int(t), s) for t, s in exp["labels"]])
