"""the chain of each gene of the bench sample on its own (sp_hla_diplotype_gene after one K1 call): where the two-stream step's longer side comes from"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from pb_starphase_amd import synth
ctx = pkg.Context(0)
fx = synth.HlaFixture()
db = fx.make_db(pkg, ctx)
wl = synth.Config2Workload(fx, n_reads=10000, seed=1)
R = ctx.upload(wl.reads)
k1 = db.realign_reads(R)
names = ["cons_steps", "anchor_k2", "anchor_type", "type_consensus_ref", "k2_cells_cdna", "k2_cells_dna", "k2_scan"] + \
        ["host:" + k for k in ("hla_select", "hla_segments", "hla_dual_hpc", "hla_dual_dna", "hla_groups", "hla_typing", "k2_setup", "k2_wait")]
for g in range(len(fx.genes)):
    db.diplotype_gene(g, R, k1)
    ctx.profile_reset(); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        call, c1, c2, _ = db.diplotype_gene(g, R, k1)
    dt = (time.perf_counter() - t0) / 5
    print(fx.genes[g], f"{1e3 * dt:.2f} ms alone; reads {call.n_reads}, consensus lengths {len(c1)} {len(c2)}, windows {ctx.profile_get('cons_windows')[2] / 5:.0f}")
    print("    launch triples", ctx.profile_get("cons_windows")[2] / 5, "consensus columns", ctx.profile_get("cons_columns")[2] / 5, "cut windows", ctx.profile_get("cons_cut_windows")[2] / 5,
          "expansions", ctx.profile_get("cons_expansions")[2] / 5)
    print("   ", {n.replace("host:", "h:"): round(ctx.profile_get(n)[0] / 5, 2) for n in names})
