"""per-kernel time of one sp_cyp_diplotype call (scenario 1 of the real-shape locus, 2,000 reads)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from pb_starphase_amd import synth
import cyp_cases_real as cr
ctx = pkg.Context(0)
cfg, gene_def = cr.load_db()
locus = synth.Chr22Locus(cfg, gene_def, seed=3)
db = pkg.ffi.CypDb(ctx, cfg, gene_def, locus.sequence, locus.start)
name, haps, expected = cr.scenarios(locus)[int(sys.argv[1]) if len(sys.argv) > 1 else 1]
R = ctx.upload(locus.sample(np.random.default_rng(7), haps, 2000))
db.diplotype(R)
ctx.profile_reset(); ctx.synchronize()
t0 = time.perf_counter(); call, _c, _l = db.diplotype(R); dt = time.perf_counter() - t0
print(name, "total ms", round(1e3 * dt, 1), call.hap1.decode(), call.hap2.decode())
for k in ("anchor", "k3_region_cells", "k3_af_trace", "k3_af_dp", "k4_af_trace", "k4_af_dp", "segments", "cons_steps", "align", "align_trace", "k9_graph", "k7_score", "k4_weight_cells", "k5_chain_reads", "k5_pairs",
          "host:cyp_regions", "host:cyp_segments", "host:cyp_consensus", "host:cyp_merge", "host:cyp_typing", "host:cyp_weights", "host:cyp_chains", "host:cyp_chain_pair"):
    ms, n, cells = ctx.profile_get(k)
    print(f"  {k:22s} {ms:8.3f} ms  launches {n:5d}  cells {cells}")
# where the control kernel's time goes (100 MHz ticks of the slowest problem of every batch, summed) and how the launches split
for k in ("cons_ticks_reduce", "cons_ticks_result", "cons_ticks_search", "cons_ticks_tail"):
    print(f"  {k:22s} {ctx.profile_get(k)[2] / 100.0 / 1e3:8.3f} ms")
for k in ("cons_windows", "cons_cut_windows", "cons_expansions", "cons_columns"):
    print(f"  {k:22s} {ctx.profile_get(k)[2]}")
