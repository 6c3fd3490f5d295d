"""K whole CYP2D6 samples (2,000 reads) in flight on K contexts, one host thread each: wall time per sample and the stages' host times summed over the contexts.
usage: cyp_in_flight.py [K=3] [reps=6]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from pb_starphase_amd import synth
import cyp_cases_real as cr
K = int(sys.argv[1]) if len(sys.argv) > 1 else 3
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
cfg, gene_def = cr.load_db()
locus = synth.Chr22Locus(cfg, gene_def, seed=3)
scen = cr.scenarios(locus)
lanes = []
for k in range(K):
    c = pkg.Context(0)
    c.set_option("hla_split_genes", 0 if K > 1 else 1)
    db = pkg.ffi.CypDb(c, cfg, gene_def, locus.sequence, locus.start)
    sets = [c.upload(locus.sample(np.random.default_rng(7 + j), scen[j][1], 2000)) for j in (0, 1)]
    for s in sets:
        db.diplotype(s)
    lanes.append((c, db, sets))
for c, _d, _s in lanes:
    c.profile_reset(); c.synchronize()
def work(c, db, sets):
    for i in range(reps):
        db.diplotype(sets[i % 2])
th = [threading.Thread(target=work, args=l) for l in lanes]
t0 = time.perf_counter()
for t in th: t.start()
for t in th: t.join()
dt = time.perf_counter() - t0
names = ("host:cyp_regions", "host:cyp_segments", "host:cyp_consensus", "host:cyp_merge", "host:cyp_weights", "host:cyp_chain_pair", "anchor", "k3_region_cells", "cons_steps", "k9_graph", "align_trace", "k4_weight_cells", "k5_pairs")
tot = {n: round(sum(c.profile_get(n)[0] for c, _d, _s in lanes) / (K * reps), 2) for n in names}
print(f"{K} in flight: {1e3 * dt / (K * reps):.1f} ms per sample (wall / samples), {1e3 * dt / reps:.1f} ms per sample on its lane; per sample: {tot}")
