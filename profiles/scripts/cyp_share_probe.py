"""A rank's small share of configs[4] (32 samples): what the CYP2D6 cohort call costs by streams / samples per stream, and where its host time goes."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from pb_starphase_amd import synth
import bench, cyp_cases_real as cr
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ctx = pkg.Context(0)
fx = synth.HlaFixture()
cfg, gene_def = cr.load_db()
locus = synth.Chr22Locus(cfg, gene_def, seed=3)
cdb = pkg.ffi.CypDb(ctx, cfg, gene_def, locus.sequence, locus.start)
scen = cr.scenarios(locus)
panel = bench.VariantPanel(pkg)
sh = bench.CohortShare(pkg, fx, locus, scen, panel, list(range(N)))
sets = [ctx.upload_format(pkg.ffi.SP_SEQ_BAM4, *p) for p in sh.cyp_payloads]
stages = ["host:cyp_" + k for k in ("regions", "segments", "consensus", "merge", "typing", "weights", "chains", "chain_pair")] + ["host:k8_loop", "host:k8_result_wait", "cons_steps", "k9_graph"]
CASES = ((6, 12, 0), (6, 12, 1), (1, 64, 0), (1, 64, 1), (2, 8, 1), (4, 8, 0), (4, 8, 1), (6, 5, 1), (6, 12, 0), (6, 12, 1))
if len(sys.argv) > 2 and sys.argv[2] == "three":
    CASES = ((6, 12, 0), (3, 10, 0), (6, 12, 0), (3, 10, 0), (3, 10, 1))
if len(sys.argv) > 2 and sys.argv[2] == "auto":           # the library's own choice of consensus mode (k8_persistent 2), round 5
    CASES = ((2, 12, 2), (2, 12, 2), (2, 12, 0), (1, 64, 2), (4, 8, 2), (2, 12, 2))
if len(sys.argv) > 2 and sys.argv[2] == "streams":
    CASES = ((6, 12, 0), (2, 12, 0), (3, 12, 0), (4, 12, 0), (6, 12, 0), (8, 12, 0), (2, 12, 0), (3, 12, 0), (4, 12, 0))
for streams, mg, pers in CASES:
    ctx.set_option("cyp_cohort_streams", streams); ctx.set_option("cyp_cohort_min_group", mg); ctx.set_option("k8_persistent", pers)
    cdb.diplotype_cohort(sets)
    ctx.profile_reset(); ctx.synchronize(); t0 = time.perf_counter()
    calls = cdb.diplotype_cohort(sets)
    dt = time.perf_counter() - t0
    good = sum(sorted([c[0].hap1.decode(), c[0].hap2.decode()]) == sorted(e) for c, e in zip(calls, sh.cyp_expected))
    parts = min(max(1, N // mg), streams)
    print({0: "launches  ", 1: "persistent", 2: "auto      "}[pers], int(ctx.profile_get("cons_persistent_batches")[2]), f"{N} samples, {parts} stream(s) x groups of {-(-N // parts)}: {1e3 * dt:7.1f} ms  ({1e3 * dt / N:5.2f} ms per sample, {good}/{N} equal truth)  " +
          " ".join(f"{s.split(':')[-1].replace('cyp_', '')} {ctx.profile_get(s)[0]:.0f}" for s in stages) +
          f" | batches {ctx.profile_get('cons_steps')[1]} steps {ctx.profile_get('cons_windows')[2]} expansions {ctx.profile_get('cons_expansions')[2]} columns {ctx.profile_get('cons_columns')[2]}", flush=True)
