"""The seeded K1 stage by stage against oracle/mm2.c on configs[1] reads (GPU).  usage: python profiles/scripts/k1_seeded_check.py [n_reads] [max_alleles_per_gene]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
pkg = ge.load_package()
from pb_starphase_amd import synth
import oracle_ffi, mm2_ffi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mx = int(sys.argv[2]) if len(sys.argv) > 2 else 0
o = oracle_ffi.load(); mm = mm2_ffi.Mm2(o)
fx = synth.HlaFixture(max_alleles_per_gene=mx or None)
wl = synth.Config2Workload(fx, n_reads=n, seed=1000)
dna_ids = [a for a in range(len(fx.ids)) if fx.dna[a]]
idx = mm2_ffi.Index(mm, [fx.dna_fwd(a) for a in dna_ids])
ctx = pkg.Context(0)
db = fx.make_db(pkg, ctx)
t0 = time.time()
info = db.seed_index_info()
print("index", info, "built in", round(time.time() - t0, 3), "s; oracle:", mm.L.omm_index_n_minimizers(idx.h), idx.mid_occ)
assert info["minimizers"] == mm.L.omm_index_n_minimizers(idx.h) and info["mid_occ"] == idx.mid_occ and info["sequences"] == len(dna_ids)
reads = ctx.upload(wl.reads)
# sketch
bad = 0
for r in range(min(n, 16)):
    h, p, s = reads.sketch(r)
    H, P, S = mm.sketch(wl.reads[r])
    if not (np.array_equal(h, H) and np.array_equal(p, P) and np.array_equal(s, S)):
        bad += 1; print("sketch differs on read", r, len(h), len(H))
print("sketch: reads differing", bad)
# chains + hits + pick
nbad_c = nbad_h = nbad_p = 0
for r in range(n):
    au = db.realign_seeded_audit(reads, r)
    regs, st = idx.chain_stage(wl.reads[r])
    exp = np.column_stack([regs[:, :8], (regs[:, 9] > 0).astype(np.int32)])
    got = np.column_stack([au["chains"][:, :8], au["chains"][:, 9]])
    if exp.shape != got.shape or not np.array_equal(exp, got):
        nbad_c += 1
        if nbad_c <= 3:
            print("chains differ on read", r, exp.shape, got.shape, "stats", st, au["counters"])
            k = min(len(exp), len(got))
            d = np.flatnonzero((exp[:k] != got[:k]).any(axis=1))
            print("  first rows", d[:5]); 
            for x in d[:3]: print("   exp", exp[x], "got", got[x])
    pick, hits, nc = idx.k1_seeded(wl.reads[r])
    gh = au["hits"]
    ok = len(gh) == len(hits)
    if ok:
        for a, b in zip(gh, hits):
            ea = (dna_ids[b["rid"]],) + tuple(int(b[k]) for k in mm2_ffi.SEED_HIT_FIELDS[1:])
            ga = tuple(int(a[k]) for k in pkg.ffi.K1_HIT_FIELDS)
            if ea != ga: ok = False; print("  hit differs read", r, "\n   exp", ea, "\n   got", ga); break
    if not ok: nbad_h += 1
    if pick != au["pick"]: nbad_p += 1; print("pick differs", r, pick, au["pick"])
print("reads", n, "chain lists differing", nbad_c, "hit lists differing", nbad_h, "picks differing", nbad_p, "counters", au["counters"])
