"""Times the consensus kernels at BASELINE config-2 scale: 10k reads -> K1 -> per-gene segments -> dual consensus (HPC, two pass)
-> one consensus per read group.  Run on the GPU box:  python profiles/scripts/consensus_scale.py [n_reads]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge

pkg = ge.load_package()
from pb_starphase_amd import synth

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
fx = synth.HlaFixture()
wl = synth.Config2Workload(fx, n_reads=n_reads, seed=1000)
ctx = pkg.Context(0)
db = fx.make_db(pkg, ctx)
reads = ctx.upload(wl.reads)
out = db.realign_reads(reads)
cfg = lambda **kw: pkg.ffi.sp_cons_config(kw.get("min_count", 3), 100, 1, kw.get("dual", 1), 400, 50, 0.10)
L = pkg.ffi.lib()
for g in range(len(fx.genes)):
    sel = np.flatnonzero((out["status"] == 0) & (out["gene"] == g))
    segs = [wl.reads[r][out[r]["seg_start"]:out[r]["seg_end"]] for r in sel]
    t0 = time.perf_counter()
    hpcs = []
    import ctypes as C
    for s in segs:
        buf = C.create_string_buffer(len(s) + 1)
        k = L.sp_hpc(s.encode(), len(s), buf)
        hpcs.append(buf.raw[:k].decode())
    t_hpc = time.perf_counter() - t0
    mn_h, mn_d = int(out[sel]["hpc_offset"].min()), int(out[sel]["dna_offset"].min())
    off_h = [None if int(out[r]["hpc_offset"]) == mn_h else int(out[r]["hpc_offset"]) - mn_h + 200 for r in sel]
    off_d = [None if int(out[r]["dna_offset"]) == mn_d else int(out[r]["dna_offset"]) - mn_d + 200 for r in sel]
    H, D = ctx.upload(hpcs), ctx.upload(segs)
    for rep in range(2):
        ctx.profile_reset()
        t0 = time.perf_counter(); dual = ctx.consensus(H, cfg(), offsets=off_h, two_pass=True); t_dual = time.perf_counter() - t0
        ms_dual = ctx.profile_get("cons_steps")
        ctx.profile_reset()
        t0 = time.perf_counter()
        groups = []
        for grp in (np.flatnonzero(dual["is_cons1"]), np.flatnonzero(~dual["is_cons1"])):
            if len(grp):
                groups.append(ctx.consensus(D, cfg(dual=0), offsets=[off_d[i] for i in grp], read_idx=grp.astype(np.uint32)))
        t_grp = time.perf_counter() - t0
        ms_grp = ctx.profile_get("cons_steps")
    truth = {fx.dna_fwd(a) for (gg, _c, _d, a) in wl.consensus if gg == g}
    print(f"gene {fx.genes[g]}: {len(sel)} segments, hpc on host {t_hpc:.2f}s, dual(HPC, two pass) {t_dual*1e3:.1f} ms (kernels {ms_dual[0]:.1f} ms / {ms_dual[1]} passes), "
          f"groups {t_grp*1e3:.1f} ms (kernels {ms_grp[0]:.1f} ms); is_dual={dual['is_dual']} split_at={dual['split_at']} group sizes {int(dual['is_cons1'].sum())}/{int((~dual['is_cons1']).sum())}; "
          f"lens {[len(x['cons'][0]) for x in groups]}; consensus == truth allele (as substring): {[any(t in x['cons'][0] for t in truth) for x in groups]}")

# the whole sample through the gene driver (all genes batched)
for rep in range(3):
    ctx.profile_reset()
    t0 = time.perf_counter()
    calls, _ = db.diplotype_genes(list(range(len(fx.genes))), reads, out)
    dt = time.perf_counter() - t0
    prof = {k: ctx.profile_get(k) for k in ("hla_segments", "cons_steps", "k2_cells_cdna", "k2_cells_dna", "k2_scan")}
print(f"sp_hla_diplotype_genes (both genes, {n_reads} reads): {dt*1e3:.1f} ms; kernels:", {k: round(v[0], 1) for k, v in prof.items()},
      "calls:", [(c.allele1, c.allele2, c.is_dual, c.dual_passed) for c, _a, _b in calls], "truth:", [a for (_g, _c, _d, a) in wl.consensus])
