"""Several host threads, each with its own context (helper streams on), database handle and sample, run reads -> diplotype again and again; two more
threads share ONE database handle.  Every step's records and calls must equal the thread's first step's (run on the GPU box).
usage: thread_stress.py <threads> <steps>"""
import os, sys, threading, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from pb_starphase_amd import synth
n_threads, steps = int(sys.argv[1]), int(sys.argv[2])
fx = synth.HlaFixture()
genes = list(range(len(fx.genes)))
key = lambda k1, calls, is1: (k1.tobytes(), [(c.status, c.allele1, c.allele2, c.typed1, c.typed2, c.counts1, c.counts2, c.is_dual, c.dual_passed, c.maf, c.cdf, c1, c2) for c, c1, c2 in calls], is1.tobytes())
workers = []
shared_ctx = pkg.Context(0); shared_db = fx.make_db(pkg, shared_ctx)
for t in range(n_threads):
    wl = synth.Config2Workload(fx, n_reads=3000, seed=500 + t)
    if t >= n_threads - 2:                       # the last two threads: contexts of their own, ONE database handle between them
        c = pkg.Context(0); d = shared_db
    else:
        c = pkg.Context(0); d = fx.make_db(pkg, c)
    workers.append((c, d, c.upload(wl.reads)))
bad = [0] * n_threads
def run(t):
    c, d, R = workers[t]
    first = None
    for s in range(steps):
        # the shared handle is bound to shared_ctx in Python; its calls go through the thread's own context
        if d is shared_db:
            saved = d.ctx; k1 = None
        k1 = pkg.ffi.HlaDb.realign_reads(d, R) if d is not shared_db else None
        if d is shared_db:
            out = np.zeros(R.n, pkg.ffi.REALIGN_DTYPE)
            c.check(pkg.ffi.lib().sp_hla_realign_reads(c._h, d._h, R._h, pkg.ffi._ptr(out), None)); k1 = out
            import ctypes as C
            k = len(genes); g = np.ascontiguousarray(genes, np.uint32)
            cf = (pkg.ffi.sp_hla_call_config * k)(*[pkg.ffi.hla_call_config() for _ in range(k)]); calls_arr = (pkg.ffi.sp_hla_call * k)()
            cap = 65536; buf = C.create_string_buffer(2 * k * cap); is1 = np.zeros(R.n, np.uint8)
            c.check(pkg.ffi.lib().sp_hla_diplotype_genes(c._h, d._h, k, pkg.ffi._ptr(g), R._h, pkg.ffi._ptr(k1), cf, calls_arr, buf, cap, pkg.ffi._ptr(is1)))
            text = lambda j: buf.raw[j * cap:(j + 1) * cap].split(b"\0", 1)[0].decode()
            calls = [(calls_arr[i], text(2 * i), text(2 * i + 1)) for i in range(k)]; is1 = is1.astype(bool)
        else:
            calls, is1 = d.diplotype_genes(genes, R, k1)
        now = key(k1, calls, is1)
        if first is None: first = now
        elif now != first: bad[t] += 1
th = [threading.Thread(target=run, args=(t,)) for t in range(n_threads)]
t0 = time.time()
for x in th: x.start()
for x in th: x.join()
print(f"{n_threads} threads x {steps} steps in {time.time() - t0:.1f} s; steps that differ from the thread's first step: {bad}")
sys.exit(1 if any(bad) else 0)
