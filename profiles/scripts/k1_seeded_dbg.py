import os, sys
import numpy as np
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
pkg = ge.load_package()
from pb_starphase_amd import synth
import oracle_ffi, mm2_ffi, hla_expected as hx
import test_gpu_seeded as T
o = oracle_ffi.load()
fx = synth.HlaFixture(max_alleles_per_gene=300, seed=5)
ctx = pkg.Context(0); db = fx.make_db(pkg, ctx)
idx, dna_ids = hx.seed_index(o, fx)
rng = np.random.default_rng(9)
reads = T.varied_reads(fx, synth, rng)
R = ctx.upload(reads)
au = db.realign_seeded_audit(R, 0)
regs, st = idx.chain_stage(reads[0])
print("stats", st, au["counters"], "mid_occ", idx.mid_occ, db.seed_index_info())
E = {(int(x[0]), int(x[1])): x for x in regs}
G = {(int(x[0]), int(x[1])): x for x in au["chains"]}
miss = [k for k in E if k not in G]
print("missing", len(miss), miss[:10])
for k in miss[:5]: print(E[k])
X, Y = idx.anchors(reads[0])
key = (X >> np.uint64(32))
for k in miss[:3]:
    kk = (np.uint64(k[1]) << np.uint64(31)) | np.uint64(k[0])
    sel = key == kk
    print(k, "anchors", int(sel.sum()), [(int(x & np.uint64(0xffffffff)), int(y & np.uint64(0xffffffff))) for x, y in zip(X[sel], Y[sel])][:70])
