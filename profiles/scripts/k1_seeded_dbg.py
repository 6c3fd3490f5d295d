import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
pkg = ge.load_package()
from pb_starphase_amd import synth
import oracle_ffi, mm2_ffi, hla_expected as hx
import test_gpu_seeded as T
o = oracle_ffi.load()
fx = synth.HlaFixture(max_alleles_per_gene=300, seed=5)
ctx = pkg.Context(0); db = fx.make_db(pkg, ctx)
# grow the pools first, as a long test session does
big = synth.HlaFixture()
bdb = big.make_db(pkg, ctx)
wl = synth.Config2Workload(big, n_reads=3000, seed=1000)
Rb = ctx.upload(wl.reads); bdb.realign_reads(Rb)
rng = np.random.default_rng(12)
reads = T.varied_reads(fx, synth, rng, n_per=3)[:20]
R = ctx.upload(reads)
whole = db.realign_reads(R)
exp, _ = hx.k1_expected_seeded(o, fx, reads)
print("expected", [e["status"] for e in exp])
print("whole   ", whole["status"].tolist())
for sl in ("7", "3", "1"):
    os.environ["SP_K1_SLICE"] = sl
    for k in range(3):
        s = db.realign_reads(R)
        print("slice", sl, s["status"].tolist(), "same" if s.tobytes() == whole.tobytes() else "DIFFERENT", [i for i in range(20) if s[i].tobytes() != whole[i].tobytes()])
os.environ.pop("SP_K1_SLICE")
idx, dna_ids = hx.seed_index(o, fx)
pick, hits, nc = idx.k1_seeded(reads[19])
print("statement pick", pick)
for h in hits: print("  exp", tuple(int(h[k]) for k in mm2_ffi.SEED_HIT_FIELDS))
for k in range(6):
    au = db.realign_seeded_audit(R, 19)
    print("device pick", au["pick"], "n_hits", len(au["hits"]))
    for h in au["hits"]: print("  got", tuple(int(h[k]) for k in pkg.ffi.K1_HIT_FIELDS))
regs, st = idx.chain_stage(reads[19])
exp = np.column_stack([regs[:, :8], (regs[:, 9] > 0).astype(np.int32)])
print("statement chains", len(exp), "selected", exp[exp[:, 8] > 0].tolist())
for k in range(8):
    au = db.realign_seeded_audit(R, 19)
    got = np.column_stack([au["chains"][:, :8], au["chains"][:, 9]])
    if got.shape == exp.shape and np.array_equal(got, exp): print("run", k, "chains equal"); continue
    print("run", k, "chains DIFFER", got.shape, exp.shape, au["counters"])
    E = {tuple(x[:8]) for x in exp.tolist()}; G = {tuple(x[:8]) for x in got.tolist()}
    print("   only statement", sorted(E - G)[:6]); print("   only device", sorted(G - E)[:6])
    print("   device selected", got[got[:, 8] > 0].tolist())
