"""library vs oracle-assembled pipeline on one configs[2] scenario: where the consensus sets differ (debugging aid; run on the GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from pb_starphase_amd import synth
import oracle_ffi as of, cyp_cases_real as cr, cyp_pipeline as cp
o = of.load()
ctx = pkg.Context(0)
cfg, gene_def = cr.load_db()
locus = synth.Chr22Locus(cfg, gene_def, seed=3)
db = pkg.ffi.CypDb(ctx, cfg, gene_def, locus.sequence, locus.start)
tm, vs = db.templates(), db.variants()
names, rows = db.alleles()
bb = cfg["cyp_coordinates"]["CYP2D6_wfa_backbone"]
odb = cp.Db([t[2] for t in tm], [t[0] for t in tm], [t[1] for t in tm], [t[3] for t in tm], [t[4] for t in tm],
            locus.slice(bb["start"], bb["end"]), [(p - bb["start"], r, a) for p, r, a, _l, _v in vs], [v[4] for v in vs], names, rows, var_labels=[v[3] for v in vs])
name = sys.argv[1]; n = int(sys.argv[2]); seed = int(sys.argv[3])
haps, expected = {s[0]: (s[1], s[2]) for s in cr.scenarios(locus)}[name]
reads = locus.sample(np.random.default_rng(seed), haps, n)
st = {}
exp = cp.diplotype(o, odb, reads, cfg=db.cfg, stages=st)
call, cons, labels = db.diplotype(ctx.upload(reads))
print("status", call.status, exp["status"], "gave_up", call.searches_gave_up)
print("library", len(cons), [len(c) for c in cons], labels)
print("oracle ", len(exp["consensus"]), [len(c) for c in exp["consensus"]], [(int(t), s) for t, s in exp["labels"]])
print("calls", call.hap1.decode(), call.hap2.decode(), "|", exp.get("hap1"), exp.get("hap2"), "| truth", expected)
for i, (a, b) in enumerate(zip(cons, exp["consensus"])):
    if a != b:
        p = next((k for k in range(min(len(a), len(b))) if a[k] != b[k]), min(len(a), len(b)))
        print("consensus", i, "differs at", p, a[max(0, p - 15):p + 15], b[max(0, p - 15):p + 15])
print("group sizes oracle", np.bincount(st["group_of"]).tolist())
