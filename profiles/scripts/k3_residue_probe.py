"""The two reads of the six configs[2] samples whose region hit list differs from the CPU port's (tests/test_gpu_concordance.py K3_RESIDUE): the port's hits, the library's hits, and
every placement of the 39 templates on the read -- the unit-cost cell's numbers and the same placement re-scored the reference's way (two-piece affine, 256 diagonals).
Run on the GPU box:  python profiles/scripts/k3_residue_probe.py"""
import gzip, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge

pkg = ge.load_package()
from pb_starphase_amd import synth
import cyp_cases_real as cr

gold = json.load(gzip.open(os.path.join(ROOT, "tests", "golden", "concordance.json.gz"), "rt"))
cfg, gene_def = cr.load_db()
locus = synth.Chr22Locus(cfg, gene_def, seed=3)
ctx = pkg.Context(0)
db = pkg.ffi.CypDb(ctx, cfg, gene_def, locus.sequence, locus.start)
tm = db.templates()
tseqs = [t[3] for t in tm]
ttype = np.array([t[0] for t in tm], np.int32)
T = ctx.upload(tseqs)
PEN = {7, 1, 4}                                    # SP_CYP_DELETION, REP6, REP7: the types whose unmapped bases count (haplotyper.rs:185-191)
for name, haps, expected in cr.scenarios(locus):
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        continue
    g = gold["cyp"]["scenarios"][name]
    reads = locus.sample(np.random.default_rng(7), haps, 2000)
    R = ctx.upload(reads)
    hits = ctx.cyp_find_regions(T, ttype, R, 0.5)
    mine = [[] for _ in reads]
    for h in hits:
        mine[int(h["read"])].append((int(h["template_idx"]), int(h["start"]), int(h["end"]), int(h["nm"]), int(h["unmapped"])))
    bad = [r for r, (a, b) in enumerate(zip(mine, g["regions"])) if [x[:3] for x in a] != [tuple(y[:3]) for y in b]]
    print(name, "reads that differ:", bad, flush=True)
    for r in bad:
        print("  read", r, "len", len(reads[r]))
        print("   port   :", [tuple(y) for y in g["regions"][r]])
        print("   library:", mine[r])
        Rr = ctx.upload([reads[r]])
        nt = len(tseqs)
        diag, votes = ctx.anchor_batch_topk(T, Rr, np.arange(nt), np.zeros(nt, np.uint32), 4)
        rows = []
        for t in range(nt):
            for k in range(4):
                if votes[t][k] < 4:
                    continue
                cap = int(0.05 * len(tseqs[t])) + 1
                al = ctx.align_batch(T, Rr, [t], [0], [int(diag[t][k])], [min(cap, 511)])[0]
                if not al["ok"]:
                    rows.append((t, k, int(votes[t][k]), "lost"))
                    continue
                tl = len(tseqs[t])
                unm = tl - (int(al["a_end"]) - int(al["a_start"]))
                pen = int(ttype[t]) in PEN
                sc = max(int(al["nm"]) + (unm if pen else 0), 0.1) / (tl if pen else tl - unm)
                d = ((int(al["b_start"]) - int(al["a_start"])) + (int(al["b_end"]) - int(al["a_end"]))) // 2
                af = ctx.affine_rescore(T, Rr, [(t, 0, d)], a=1, band=256)[0]
                unm2 = tl - (int(af["a_end"]) - int(af["a_start"]))
                sc2 = max(int(af["nm"]) + (unm2 if pen else 0), 0.1) / max(1, (tl if pen else tl - unm2)) if af["score"] > 0 else None
                rows.append((t, k, int(votes[t][k]), int(al["b_start"]), int(al["b_end"]), int(al["nm"]), unm, round(sc, 5), "| mm2", int(af["b_start"]), int(af["b_end"]), int(af["nm"]), unm2,
                             None if sc2 is None else round(sc2, 5), "type", int(ttype[t])))
        for row in sorted(rows, key=lambda x: (x[3] if isinstance(x[3], int) else 1 << 30, x[0])):
            print("     ", row)
