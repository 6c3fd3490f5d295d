"""Which restriction of K1's argmin comes closest to the seeded map's winner (best_n 5)?  Exhaustive cell matrix of configs[1] on the GPU, candidate rules in numpy,
compared with the port's winners of tests/golden/concordance.json.gz."""
import gzip, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from pb_starphase_amd import synth
ctx = pkg.Context(0)
ctx.set_option("mm2_rescore", 0)
fx = synth.HlaFixture(); db = fx.make_db(pkg, ctx)
wl = synth.Config2Workload(fx, n_reads=10000, seed=1000)
gold = json.load(gzip.open(os.path.join(ROOT, "tests", "golden", "concordance.json.gz"), "rt"))["hla"]
win = np.array(gold["winner"])
N = 2500
R = ctx.upload(wl.reads[:N])
out, cell = db.realign_reads(R, cells=True)
valid = cell != 0xFFFFFFFF
nm = (cell >> 16).astype(np.float64); span = (cell & 0xFFFF).astype(np.float64)
alen = np.array([len(s) for s in fx.dna], np.float64)[None, :]
ratio = np.where(valid & (span > 0), np.maximum(nm, 0.1) / np.maximum(span, 1.0), np.inf)
acc = valid & ((nm + (alen - span)) / np.maximum(alen, 1) <= 0.5) & (ratio <= 0.03)
ratio = np.where(acc, ratio, np.inf)
w = win[:N]
print("current winner == seeded winner:", int((out["best_allele"] == w).sum()), "of", N)
smax = np.where(acc, span, 0).max(1)
for tol in (0, 4, 16, 64, 128, 256, 512):
    ok = acc & (span >= (smax[:, None] - tol))
    pick = np.where(ok, ratio, np.inf).argmin(1)
    print("rule span >= max acceptable span -", tol, ":", int((pick == w).sum()), "same allele;", int((np.array(fx.gene_of)[pick] == np.array(fx.gene_of)[w]).sum()), "same gene")
lmax = np.where(acc, alen, 0).max(1)
for tol in (0, 16, 64, 256):
    ok = acc & (alen >= (lmax[:, None] - tol))
    pick = np.where(ok, ratio, np.inf).argmin(1)
    print("rule allele length >= max acceptable length -", tol, ":", int((pick == w).sum()))
# where the seeded winner stands among the acceptable cells by span and by ratio
rows = np.arange(N)
ws = span[rows, w]; wr = ratio[rows, w]
print("seeded winner has the maximal acceptable span:", int((ws >= smax).sum()), "; is acceptable under the contract:", int(np.isfinite(wr).sum()))
rank = (ratio < wr[:, None]).sum(1)
print("alleles strictly better in ratio than the seeded winner: median", int(np.median(rank)), "zero for", int((rank == 0).sum()))
ties = (np.where(acc & (span >= smax[:, None]), ratio, np.inf) == np.where(acc & (span >= smax[:, None]), ratio, np.inf).min(1)[:, None]).sum(1)
print("ties at the top of rule tol 0: median", int(np.median(ties)), "max", int(ties.max()))
print("---- top-K by a chain-score proxy, then the acceptance loop in output order (by alignment score)")
score_dp = np.where(acc, span - 5.0 * nm, -np.inf)          # ~ dp score: matches - 4 X ... (a = 1): span - nm - 4 nm
for wgt in (5, 10, 19, 30, 40):
    for K in (6,):
        C = np.where(acc, span - wgt * nm, -np.inf)
        # ties -> lowest allele index: stable argsort on (-C)
        top = np.argsort(-C, axis=1, kind="stable")[:, :K]
        rr = np.take_along_axis(ratio, top, 1)
        sd = np.take_along_axis(score_dp, top, 1)
        # output order: by dp score descending (stable); first minimum of the ratio wins
        order = np.argsort(-sd, axis=1, kind="stable")
        rr_o = np.take_along_axis(rr, order, 1); top_o = np.take_along_axis(top, order, 1)
        pick = top_o[rows, rr_o.argmin(1)]
        print("proxy span -", wgt, "* nm, top", K, ":", int((pick == w).sum()), "same allele")
