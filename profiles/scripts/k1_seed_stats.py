"""How big is the seeded map of a configs[1] read (oracle/mm2.c, the reference's call pattern)?  minimizers, seeds kept, anchors, targets, chains per read.
CPU only.  usage: python profiles/scripts/k1_seed_stats.py [n_reads]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
ge.load_package()
from pb_starphase_amd import synth
import oracle_ffi, mm2_ffi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
o = oracle_ffi.load(); mm = mm2_ffi.Mm2(o)
fx = synth.HlaFixture()
wl = synth.Config2Workload(fx, n_reads=10000, seed=1000)
dna_ids = [a for a in range(len(fx.ids)) if fx.dna[a]]
t0 = time.time()
idx = mm2_ffi.Index(mm, [fx.dna_fwd(a) for a in dna_ids])
print("index", time.time() - t0, "s; minimizers", mm.L.omm_index_n_minimizers(idx.h), "mid_occ", idx.mid_occ)
L = mm.L
L.omm_chain_stage.restype = C.c_int32
L.omm_chain_stage.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
rng = np.random.default_rng(0)
rows = []
cap = 20000
regs = np.zeros((cap, 10), np.int32); st = np.zeros(8, np.int64)
tt = 0.0
for r in rng.choice(len(wl.reads), n, replace=False):
    q = o.encode(wl.reads[r])
    t1 = time.time()
    nr = L.omm_chain_stage(idx.h, q.ctypes.data, len(q), C.byref(idx.o), regs.ctypes.data, cap, st.ctypes.data)
    tt += time.time() - t1
    sel = regs[:min(nr, cap)]
    nsel = int((sel[:, 9] > 0).sum())
    top = sel[0] if nr else None
    rows.append(list(st) + [len(q), nr, nsel, int(top[2]) if nr else 0, int(top[3]) if nr else 0, int(sel[min(nr, 6) - 1][2]) if nr else 0,
                            int(sel[:, 3].max()) if nr else 0, int((sel[:, 2] == (top[2] if nr else 0)).sum())])
A = np.array(rows)
names = ["minimizers", "seeds_in_index", "seeds_kept", "anchors", "targets", "chains", "selected", "mid_occ", "read_len", "nr", "nsel", "top_score", "top_cnt", "score6", "max_cnt", "ties_at_top"]
for i, nm in enumerate(names):
    c = A[:, i]
    print(f"{nm:16s} mean {c.mean():10.1f}  min {c.min():8d}  p50 {int(np.median(c)):8d}  p90 {int(np.percentile(c, 90)):8d}  max {c.max():8d}")
print("chain stage ms per read", 1e3 * tt / n)
