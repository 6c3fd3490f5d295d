"""CYP2D6 calls of several WGS-like samples (100 reads each) in flight on one GPU: one host thread + context per stream"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from pb_starphase_amd import synth
import cyp_cases_real as cr
cfg, gene_def = cr.load_db()
locus = synth.Chr22Locus(cfg, gene_def, seed=3)
n_threads = int(sys.argv[1]) if len(sys.argv) > 1 else 4
per = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n_reads = int(sys.argv[3]) if len(sys.argv) > 3 else 100
scen = cr.scenarios(locus)
workers = []
for t in range(n_threads):
    ctx = pkg.Context(0)
    db = pkg.ffi.CypDb(ctx, cfg, gene_def, locus.sequence, locus.start)
    samples = []
    for k in range(per):
        name, haps, expected = scen[(t * per + k) % 3]
        samples.append((ctx.upload(locus.sample(np.random.default_rng(100 * t + k), haps, n_reads, lo=8000, hi=16000)), expected))
    workers.append((ctx, db, samples))

def run(w, out):
    ctx, db, samples = w
    ok = 0
    for R, expected in samples:
        call, _c, _l = db.diplotype(R)
        ok += sorted([call.hap1.decode(), call.hap2.decode()]) == sorted(expected)
    out.append(ok)

for w in workers: run(w, [])
for n in (1, n_threads):
    outs = [[] for _ in range(n)]
    th = [threading.Thread(target=run, args=(workers[i], outs[i])) for i in range(n)]
    t0 = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    dt = time.perf_counter() - t0
    print(f"{n} stream(s): {n * per} samples of {n_reads} reads in {dt*1e3:.0f} ms = {dt*1e3/(n*per):.1f} ms per sample, {n*per/dt:.1f} samples/s, calls ok {sum(o[0] for o in outs)}/{n*per}")
