"""K3: (read, template) pairs none of whose top-4 anchor cells is found on 64 diagonals, by the votes of their best anchor -- what a wide-band retry rule would have to run"""
import os, sys
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
for sc in (0, 3):
    name, haps, expected = cr.scenarios(locus)[sc]
    reads = locus.sample(np.random.default_rng(7), haps, 2000)
    tt = db.templates(); tmpl = [x[3] for x in tt]
    T, R = ctx.upload(tmpl), ctx.upload(reads)
    nT, nR = len(tmpl), len(reads)
    ai = np.tile(np.arange(nT, dtype=np.uint32), nR); bi = np.repeat(np.arange(nR, dtype=np.uint32), nT)
    diag, votes = ctx.anchor_batch_topk(T, R, ai, bi, 4)
    tl = np.array([len(t) for t in tmpl])
    cap = np.minimum((0.05 * tl[ai]).astype(int) + 1, 511)
    ok = np.zeros((len(ai), 4), bool)
    for k in range(4):
        d = np.where(votes[:, k] >= 4, diag[:, k], -(2 ** 31))
        al = ctx.align_batch(T, R, ai, bi, d, cap)
        ok[:, k] = al["ok"] != 0
    none = ~ok.any(1)
    print(name, "pairs", len(ai), "no cell found", int(none.sum()))
    for thr in (8, 16, 32, 64, 128, 256):
        sel = none & (votes[:, 0] >= thr)
        print("   best anchor >= %3d votes: %6d pairs (%.2f %% of all)   mean template length %.0f" % (thr, sel.sum(), 100.0 * sel.sum() / len(ai), tl[ai][sel].mean() if sel.any() else 0))
