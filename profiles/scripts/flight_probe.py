"""K whole samples in flight (an HLA lane and a CYP2D6 lane each, reads resident): the time every lane needs for its samples.
usage: flight_probe.py [K=3] [reps=5] [hla_lanes=K] [cyp_lanes=K]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from pb_starphase_amd import synth
import cyp_cases_real as cr
K = int(sys.argv[1]) if len(sys.argv) > 1 else 3
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n_hla = int(sys.argv[3]) if len(sys.argv) > 3 else K
n_cyp = int(sys.argv[4]) if len(sys.argv) > 4 else K
cfg, gene_def = cr.load_db()
locus = synth.Chr22Locus(cfg, gene_def, seed=3)
scen = cr.scenarios(locus)
fx = synth.HlaFixture()
wl = synth.Config2Workload(fx, n_reads=10000, seed=11)
lanes = []
for k in range(n_hla):
    c = pkg.Context(0); c.set_option("hla_split_genes", 0)
    db = fx.make_db(pkg, c)
    R = c.upload(wl.reads)
    genes = list(range(len(fx.genes)))
    f = (lambda db, R, genes: lambda: db.diplotype_genes(genes, R, db.realign_reads(R)))(db, R, genes)
    f(); lanes.append(("hla", c, f))
for k in range(n_cyp):
    c = pkg.Context(0); c.set_option("hla_split_genes", 0)
    db = pkg.ffi.CypDb(c, cfg, gene_def, locus.sequence, locus.start)
    sets = [c.upload(locus.sample(np.random.default_rng(7 + j), scen[j][1], 2000)) for j in (0, 1)]
    state = {"i": 0}
    def f(db=db, sets=sets, state=state):
        db.diplotype(sets[state["i"] % 2]); state["i"] += 1
    f(); f(); state["i"] = 0
    lanes.append(("cyp", c, f))
took = [0.0] * len(lanes)
def work(x):
    t0 = time.perf_counter()
    for _ in range(reps):
        lanes[x][2]()
    took[x] = time.perf_counter() - t0
th = [threading.Thread(target=work, args=(x,)) for x in range(len(lanes))]
t0 = time.perf_counter()
for t in th: t.start()
for t in th: t.join()
dt = time.perf_counter() - t0
for x, (kind, c, _f) in enumerate(lanes):
    if kind == "cyp":
        ticks = {k: round(c.profile_get("cons_ticks_" + k)[2] / 100.0 / 1e3 / max(1, reps + 2), 2) for k in ("reduce", "result", "search", "tail")}
        print("   cyp lane", x, "per sample: control-kernel device ms", ticks, "cons_steps ms", round(c.profile_get("cons_steps")[0] / (reps + 2), 1), "k9", round(c.profile_get("k9_graph")[0] / (reps + 2), 2),
              "anchor", round(c.profile_get("anchor")[0] / (reps + 2), 2))
print(f"{n_hla} hla + {n_cyp} cyp lanes, {reps} samples each: wall {1e3 * dt:.0f} ms; per sample on its lane:", [(lanes[x][0], round(1e3 * took[x] / reps, 1)) for x in range(len(lanes))])
