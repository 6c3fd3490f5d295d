"""sp_cyp_diplotype_cohort by number of streams ("cyp_cohort_streams"), against the same samples called one by one: the CYP2D6 share of the cohort leg"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from pb_starphase_amd import synth
import cyp_cases_real as cr
N = int(sys.argv[1]) if len(sys.argv) > 1 else 48
ctx = pkg.Context(0)
cfg, gene_def = cr.load_db()
locus = synth.Chr22Locus(cfg, gene_def, seed=3)
db = pkg.ffi.CypDb(ctx, cfg, gene_def, locus.sequence, locus.start)
scen = cr.scenarios(locus)
sets, expected = [], []
for s in range(N):
    sc = scen[s % 3]
    reads = locus.sample(np.random.default_rng(20_000 + s), sc[1], 100, lo=8000, hi=16000)
    sets.append(ctx.upload_format(pkg.ffi.SP_SEQ_BAM4, *pkg.ffi.encode_bam4(reads)))
    expected.append(sc[2])
def ok(calls): return sum(sorted([c.hap1.decode(), c.hap2.decode()]) == sorted(e) for c, e in zip(calls, expected))
for R in sets[:3]: db.diplotype(R)
t0 = time.perf_counter(); calls = [db.diplotype(R)[0] for R in sets]; dt = time.perf_counter() - t0
print(f"one by one: {N} samples in {dt*1e3:.0f} ms = {dt*1e3/N:.1f} ms per sample, ok {ok(calls)}/{N}", flush=True)
for streams in (1, 2, 4, 6, 8):
    ctx.set_option("cyp_cohort_streams", streams)
    db.diplotype_cohort(sets[:2 * streams])
    ctx.profile_reset()
    t0 = time.perf_counter(); out = db.diplotype_cohort(sets); dt = time.perf_counter() - t0
    host = {k: round(ctx.profile_get("host:" + k)[0], 1) for k in ("cyp_regions", "cyp_segments", "cyp_consensus", "cyp_merge", "cyp_typing", "cyp_weights", "cyp_chains", "cyp_chain_pair", "k8_loop", "k8_launch")}
    host["k8_launches"] = int(ctx.profile_get("host:k8_launch")[1])
    print(f"{streams} stream(s): {N} samples in {dt*1e3:.0f} ms = {dt*1e3/N:.1f} ms per sample, ok {ok([o[0] for o in out])}/{N}; host ms summed over streams {host}", flush=True)
