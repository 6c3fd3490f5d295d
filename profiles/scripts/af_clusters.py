"""What keeps a K1 winner off the closed form of the re-score (edits < 16 apart or < 16 from an end)?  The traced edits of configs[1]'s winners, classified."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from pb_starphase_amd import synth
ctx = pkg.Context(0)
fx = synth.HlaFixture(); db = fx.make_db(pkg, ctx)
wl = synth.Config2Workload(fx, n_reads=10000, seed=1000)
N = 10000
R = ctx.upload(wl.reads[:N])
out = db.realign_reads(R)
ok = out["status"] == 0
idx = np.nonzero(ok)[0]
A = ctx.upload(fx.dna)
al = out["aln"][idx]
diag = ((al["b_start"] - al["a_start"]) + (al["b_end"] - al["a_end"])) // 2
res, ev = ctx.align_batch(A, R, out["best_allele"][idx].astype(np.uint32), idx.astype(np.uint32), diag.astype(np.int32), np.full(len(idx), 127, np.int32), events=True)
same = (res["nm"] == al["nm"]) & (res["a_start"] == al["a_start"]) & (res["b_end"] == al["b_end"])
print("winners", len(idx), "second run identical", int(same.sum()))
why = collections.Counter(); rows_needed = []; span_all = []
for x in range(len(idx)):
    r = res[x]; nm = int(r["nm"]); e = ev[x, :nm]
    pos = (e & 0x3FFFFFFF).astype(np.int64); typ = (e >> 30).astype(np.int64)
    order = np.argsort(pos, kind="stable"); pos = pos[order]; typ = typ[order]
    b0, b1 = int(r["b_start"]), int(r["b_end"])
    span_all.append(b1 - b0)
    if nm == 0: why["no edits"] += 1; rows_needed.append(0); continue
    near_end = (pos - b0 < 16) | (b1 - pos < 16)
    gaps = np.diff(pos)
    close = gaps < 16
    if not near_end.any() and not close.any(): why["isolated"] += 1; rows_needed.append(0); continue
    # clusters: maximal runs of edits < 16 apart
    cl = []; s = 0
    for k in range(1, nm + 1):
        if k == nm or gaps[k - 1] >= 16:
            cl.append((s, k)); s = k
    kinds = set()
    need = 0
    for (s, k) in cl:
        if k - s == 1 and not near_end[s]: continue
        t = typ[s:k]; p = pos[s:k]
        need += int(p[-1] - p[0]) + 64
        if near_end[s:k].any(): kinds.add("near an end")
        elif len(set(t.tolist())) == 1 and t[0] != 0 and (np.diff(p) <= 1).all(): kinds.add("one gap of several bases")
        elif (t == 0).all(): kinds.add("mismatches only")
        else: kinds.add("mixed")
    why[" + ".join(sorted(kinds))] += 1
    rows_needed.append(need)
for k, v in why.most_common(): print(f"{v:6d}  {k}")
rn = np.array(rows_needed); sp = np.array(span_all)
dp = rn > 0
print("pairs to the DP", int(dp.sum()), "mean span", float(sp[dp].mean()) if dp.any() else 0, "mean rows with 32-row margins around the clusters", float(rn[dp].mean()) if dp.any() else 0)
