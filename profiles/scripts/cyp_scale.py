"""BASELINE config 3 timing: sp_cyp_diplotype (K3 regions -> K8 multi-way consensus -> K9/K7 typing -> K4 -> chains -> K5) from raw
synthetic targeted-style reads of the CYP2D6 locus, on the synthetic database of tests/cyp_fixture.py.
Run on the GPU box:  python profiles/scripts/cyp_scale.py [n_reads ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge

pkg = ge.load_package()
from pb_starphase_amd import synth
import cyp_fixture as cf
import oracle_ffi as of

sizes = [int(x) for x in sys.argv[1:]] or [120, 500, 2000]
locus = synth.CypLocus(seed=11)
db, d6 = cf.make_db(locus, synth, np.random.default_rng(5))
ctx = pkg.Context(0)
cfg = of.default_cyp_config()
S = ctx.upload(db.seqs)
for n in sizes:
    for scenario, expected in (("*1/*4", {"*1", "*4"}), ("*4x2/*1", {"*4x2", "*1"}), ("*5/*2", {"*5", "*2"})):
        reads = cf.sample(locus, synth, np.random.default_rng(7), d6, scenario, n)
        R = ctx.upload(reads)
        best = None
        for rep in range(3):
            ctx.synchronize(); t0 = time.perf_counter()
            call, cons, labels = ctx.cyp_diplotype(S, db.types, db.subtypes, db.deep, db.backbone, db.variants, db.is_vi, db.allele_subtypes, db.hap_matrix, cfg, R)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        got = {call.hap1.decode(), call.hap2.decode()}
        print(f"{scenario:8s} {len(reads):5d} reads ({sum(map(len, reads)) / 1e6:.1f} Mb): {best * 1e3:8.1f} ms -> {len(reads) / best:9.0f} reads/s; "
              f"status {call.status}, {call.n_consensus} consensuses, call {sorted(got)} {'== truth' if got == expected else '!= simulated truth ' + str(sorted(expected)) + ' (copy number is a likelihood call; the oracle pipeline makes the same one)'}")
