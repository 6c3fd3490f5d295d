"""what the second stage's re-score costs sp_hla_realign_reads (10,000 reads of configs[1], alone on the device)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from pb_starphase_amd import synth
ctx = pkg.Context(0)
fx = synth.HlaFixture()
wl = synth.Config2Workload(fx, n_reads=10000, seed=1000)
db = fx.make_db(pkg, ctx)
R = ctx.upload(wl.reads)
for _ in range(2):
    db.realign_reads(R)
ctx.profile_reset(); ctx.synchronize(); t0 = time.perf_counter()
for _ in range(3):
    o = db.realign_reads(R)
dt = (time.perf_counter() - t0) / 3
names = ("anchor_k1", "k1s_seeds", "k1s_groups", "k1s_dp", "k1s_dp_big", "k1s_select", "k1s_cells", "k1s_af_trace", "k1s_af_dp", "k1_finalize", "k1_seg_retry", "k1_seg_rescore", "k1_segrs_trace", "k1_segrs_dp")
print("realign_reads %.2f ms per call" % (1e3 * dt), {k: round(ctx.profile_get(k)[0] / 3, 3) for k in names})
