"""sp_hla_realign_reads in seeded mode on the 10,000 reads of configs[1]: winners against the CPU port's (tests/golden/concordance.json.gz), stage times, and the
exhaustive mode beside it.  usage: python profiles/scripts/k1_seeded_10k.py [n_reads]"""
import gzip
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
pkg = ge.load_package()
from pb_starphase_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
doc = json.load(gzip.open(os.path.join(ROOT, "tests/golden/concordance.json.gz"), "rt"))["hla"]
fx = synth.HlaFixture()
wl = synth.Config2Workload(fx, n_reads=10000, seed=1000)
ctx = pkg.Context(0)
db = fx.make_db(pkg, ctx)
reads = ctx.upload(wl.reads[:n])
names = ("anchor_k1", "k1s_seeds", "k1s_groups", "k1s_dp", "k1s_dp_big", "k1s_select", "k1s_cells", "k1s_af_trace", "k1s_af_dp", "k1_finalize", "host:k1_total", "k1_cells", "k1_cells_deep", "k1_af_trace", "k1_af_dp")
for mode in (5, 0, 5):
    ctx.set_option("k1_best_n", mode)
    out = db.realign_reads(reads)          # warm-up (index build, pools)
    ctx.profile_reset()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        out = db.realign_reads(reads)
    dt = (time.perf_counter() - t0) / reps
    prof = {k: round(ctx.profile_get(k)[0] / reps, 3) for k in names if ctx.profile_get(k)[1]}
    win = out["best_allele"]
    same = int(sum(int(win[r]) == doc["winner"][r] for r in range(n)))
    nums = int(sum(int(win[r]) == doc["winner"][r] and (win[r] < 0 or (int(out["mm2_nm"][r]) == doc["nm"][r] and int(out["mm2_t_end"][r] - out["mm2_t_start"][r]) == doc["span"][r])) for r in range(n)))
    st = np.bincount(out["status"], minlength=4).tolist()
    print(f"k1_best_n {mode}: {1e3 * dt:.2f} ms per call of {n} reads; same allele as the port {same} / {n}, same (nm, span) too {nums}; status counts {st}")
    print("   ", prof)
    if mode:
        print("    chains per read", float(out["k1_chains"].mean()), "mappings per read", float(out["k1_mappings"].mean()))
        bad = [r for r in range(n) if int(win[r]) != doc["winner"][r]][:10]
        for r in bad:
            print("    read", r, "library", int(win[r]), int(out["mm2_nm"][r]), "port", doc["winner"][r], doc["nm"][r], doc["span"][r])
