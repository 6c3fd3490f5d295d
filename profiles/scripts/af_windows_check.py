"""The re-score over the rows around the clustered edits (context option mm2_rescore 1) against the DP over all rows (2): the same numbers?  And what each costs.
K1 winners of configs[1] (10,000 reads), K3 hits of two configs[2] scenarios (2,000 reads, 256 diagonals)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from pb_starphase_amd import synth
import cyp_cases_real as cr
ctx = pkg.Context(0)
fx = synth.HlaFixture(); db = fx.make_db(pkg, ctx)
wl = synth.Config2Workload(fx, n_reads=10000, seed=1000)
R = ctx.upload(wl.reads)
F = ["mm2_score", "mm2_nm", "mm2_t_start", "mm2_t_end", "mm2_q_start", "mm2_q_end"]
res = {}
for mode in (2, 1, 2, 1):
    ctx.set_option("mm2_rescore", mode)
    ctx.profile_reset(); ctx.synchronize(); t0 = time.time()
    out = db.realign_reads(R)
    dt = time.time() - t0
    res[mode] = out
    print("K1 mode", mode, "wall ms", round(1e3 * dt, 2), "dp ms", round(ctx.profile_get("k1_af_dp")[0], 3), "trace ms", round(ctx.profile_get("k1_af_trace")[0], 3), flush=True)
diff = np.zeros(len(wl.reads), bool)
for f in F: diff |= res[1][f] != res[2][f]
print("K1 reads whose re-scored numbers differ between the two modes:", int(diff.sum()), "of", len(diff))
for r in np.nonzero(diff)[0][:10]: print("  read", r, [int(res[1][f][r]) for f in F], [int(res[2][f][r]) for f in F])

cfg, gene_def = cr.load_db()
locus = synth.Chr22Locus(cfg, gene_def, seed=3)
cdb = pkg.ffi.CypDb(ctx, cfg, gene_def, locus.sequence, locus.start)
tm = cdb.templates(); tset = ctx.upload([t[3] for t in tm]); ttype = np.array([t[0] for t in tm], np.int32)
sc = {n: (h, e) for n, h, e in cr.scenarios(locus)}
G = ["mm2_score", "mm2_nm", "mm2_start", "mm2_end", "mm2_q_start", "mm2_q_end"]
for name in ("*1/*2", "*4+*68/*1", "*10+*36/*10"):
    reads = locus.sample(np.random.default_rng(7), sc[name][0], 2000)
    Rc = ctx.upload(reads)
    hh = {}
    for mode in (2, 1, 2, 1):
        ctx.set_option("mm2_rescore", mode)
        ctx.profile_reset(); ctx.synchronize(); t0 = time.time()
        hits = ctx.cyp_find_regions(tset, ttype, Rc, 0.5)
        dt = time.time() - t0
        hh[mode] = hits
        print(name, "K3 mode", mode, "wall ms", round(1e3 * dt, 2), "dp ms", round(ctx.profile_get("k3_af_dp")[0], 3), "hits", len(hits), flush=True)
    d = np.zeros(len(hh[1]), bool)
    for f in G: d |= hh[1][f] != hh[2][f]
    print(name, "K3 hits whose re-scored numbers differ:", int(d.sum()), "of", len(d))
    for r in np.nonzero(d)[0][:10]: print("  hit", r, [int(hh[1][f][r]) for f in G], [int(hh[2][f][r]) for f in G])
