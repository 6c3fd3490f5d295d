"""BASELINE configs[2] at its stated shape: sp_cyp_diplotype on the synthetic chr22 locus with the database's own coordinates, the 39
templates and the real variant / star-allele table (sp_cyp_db_create), six scenarios, n reads each.
Run on the GPU box:  python profiles/scripts/cyp_real.py [n_reads] [scenario ...]"""
import gzip, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge

pkg = ge.load_package()
from pb_starphase_amd import synth
import cyp_cases_real as cr

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
want = sys.argv[2:]
cfg, gene_def = cr.load_db()
locus = synth.Chr22Locus(cfg, gene_def, seed=3)
ctx = pkg.Context(0)
db = pkg.ffi.CypDb(ctx, cfg, gene_def, locus.sequence, locus.start)
for name, haps, expected in cr.scenarios(locus):
    if want and name not in want:
        continue
    reads = locus.sample(np.random.default_rng(7), haps, n)
    R = ctx.upload(reads)
    best = None
    for rep in range(2):
        ctx.synchronize(); t0 = time.perf_counter()
        call, cons, labels = db.diplotype(R)
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    got = sorted([call.hap1.decode(), call.hap2.decode()])
    print(f"{name:12s} {len(reads):5d} reads ({sum(map(len, reads)) / 1e6:.1f} Mb): {best * 1e3:8.1f} ms -> {len(reads) / best:8.0f} reads/s; status {call.status}, "
          f"{call.n_consensus} consensuses, call {got} {'== truth' if got == sorted(expected) else '!= truth ' + str(sorted(expected))}", flush=True)
    if got != sorted(expected):
        for i, (t, s) in enumerate(labels):
            print("      cons", i, t, s, len(cons[i]))
        print("      chains", list(call.chain1[:call.n1]), list(call.chain2[:call.n2]), call.score)
