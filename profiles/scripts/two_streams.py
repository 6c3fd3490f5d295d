"""Two samples in flight on one GPU: two host threads, each with its own context (stream), database handle and reads, alternate through
the reads -> diplotype step.  K1 (VALU-bound) of one sample overlaps the consensus (latency-bound) of the other."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from pb_starphase_amd import synth

fx = synth.HlaFixture()
n_threads = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
workers = []
for t in range(n_threads):
    wl = synth.Config2Workload(fx, n_reads=10000, seed=int(os.environ.get("SEED0", "1000")) + t)
    ctx = pkg.Context(0)
    db = fx.make_db(pkg, ctx)
    reads = ctx.upload(wl.reads)
    workers.append((ctx, db, reads, wl))
genes = list(range(len(fx.genes)))

def run(w, k, out):
    ctx, db, reads, wl = w
    for _ in range(k):
        o = db.realign_reads(reads)
        calls = db.diplotype_genes(genes, reads, o)[0]
    out.append([(c.allele1, c.allele2) for c, _a, _b in calls])

for w in workers: run(w, 1, [])            # warm-up
for n in (1, n_threads):
    outs = [[] for _ in range(n)]
    th = [threading.Thread(target=run, args=(workers[i], steps, outs[i])) for i in range(n)]
    t0 = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    dt = time.perf_counter() - t0
    print(f"{n} thread(s): {n * steps} samples in {dt*1e3:.1f} ms = {dt*1e3/(n*steps):.2f} ms per sample, {10000*n*steps/dt:.0f} reads/s", outs[0][0])
