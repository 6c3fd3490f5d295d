"""Aligner audit, GPU half: what the library's alignment contract gives on configs[1] / configs[2], written to gpurun_out/audit/ for
profiles/scripts/audit_cpu.py (the minimap2 restatement of oracle/mm2.c run beside it on the CPU).

  k1.npz    per read of the 10,000 configs[1] reads: winner, its (nm, span), the first really different competitor (lowest ratio above the
            winner's) with (nm, span), how many alleles tie with the winner
  k2.npz    the consensuses the sample's call produced, per consensus the K2 stats of every allele (len, nm, unmapped) x (cDNA, DNA), the winner
  k3_<i>.npz  per configs[2] scenario: the region hits of the first N reads (sp_cyp_find_regions), the sample's consensuses and the (ed, overlap)
            of every (region segment, consensus) pair (sp_cyp_weight_segments)
Reads are regenerated on the CPU side from the same seeds."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out", "audit")
N_CYP_READS = int(os.environ.get("AUDIT_CYP_READS", "400"))


def main():
    os.makedirs(OUT, exist_ok=True)
    pkg = ge.load_package()
    from pb_starphase_amd import synth
    import cyp_cases_real as cr
    ctx = pkg.Context(0)
    fx = synth.HlaFixture()
    wl = synth.Config2Workload(fx, n_reads=10000, seed=1000)
    db = fx.make_db(pkg, ctx)
    R = ctx.upload(wl.reads)
    out, cell = db.realign_reads(R, cells=True)
    n = len(wl.reads)
    win = out["best_allele"].astype(np.int64)
    valid = cell != 0xFFFFFFFF
    nm = (cell >> 16).astype(np.float64)
    span = (cell & 0xFFFF).astype(np.float64)
    ratio = np.where(valid & (span > 0), np.maximum(nm, 0.1) / np.maximum(span, 1.0), np.inf)
    # acceptance of realign_record on the cells (src/hla/realigner.rs:137-141): the cell's target length is the allele's
    alen = np.array([len(s) for s in fx.dna], np.float64)[None, :]
    acc = valid & ((nm + (alen - span)) / np.maximum(alen, 1) <= 0.5) & (ratio <= 0.03)
    ratio = np.where(acc, ratio, np.inf)
    rows = np.arange(n)
    wr = np.where(win >= 0, ratio[rows, np.maximum(win, 0)], np.inf)
    ties = (ratio == wr[:, None]).sum(1)
    above = np.where(ratio > wr[:, None], ratio, np.inf)
    ru = above.argmin(1)
    ru = np.where(np.isfinite(above[rows, ru]), ru, -1)
    g = lambda idx, arr: np.where(idx >= 0, arr[rows, np.maximum(idx, 0)], -1)
    np.savez_compressed(os.path.join(OUT, "k1.npz"), status=out["status"], winner=win, win_nm=out["nm"], win_unmapped=out["unmapped"], win_tlen=out["target_len"],
                        win_cell_nm=g(win, nm), win_cell_span=g(win, span), runner=ru, run_nm=g(ru, nm), run_span=g(ru, span), ties=ties,
                        gene=out["gene"], n_accepted=acc.sum(1))
    del cell, valid, nm, span, ratio, acc, above
    # K2: the sample's own consensuses, typed against every allele
    genes = list(range(len(fx.genes)))
    calls, _is1 = db.diplotype_genes(genes, R, out)
    cons, stats, best, gene_of_cons, cdna = [], [], [], [], []
    for gi, (call, c1, c2) in enumerate(calls):
        for c in (c1, c2):
            if not c:
                continue
            b, _ns, st, spliced = db.type_consensus(gi, c)
            cons.append(c); stats.append(st); best.append(b); gene_of_cons.append(gi); cdna.append(spliced)
    np.savez_compressed(os.path.join(OUT, "k2.npz"), cons=np.array(cons), cdna=np.array(cdna), stats=np.array(stats), best=np.array(best), gene=np.array(gene_of_cons),
                        call=np.array([[c.allele1, c.allele2] for c, _a, _b in calls]))
    # K3 / K4 on the six configs[2] samples
    cfg, gene_def = cr.load_db()
    locus = synth.Chr22Locus(cfg, gene_def, seed=3)
    cdb = pkg.ffi.CypDb(ctx, cfg, gene_def, locus.sequence, locus.start)
    tm = cdb.templates()
    tset = ctx.upload([t[3] for t in tm])
    ttype = np.array([t[0] for t in tm], np.int32)
    for i, (name, haps, expected) in enumerate(cr.scenarios(locus)):
        reads = locus.sample(np.random.default_rng(7), haps, 2000)
        Rc = ctx.upload(reads)
        call, ccons, labels = cdb.diplotype(Rc)
        sub = reads[:N_CYP_READS]
        hits = ctx.cyp_find_regions(tset, ttype, ctx.upload(sub), 0.5)
        segs = [sub[int(h["read"])][int(h["start"]):int(h["end"])] for h in hits]
        keep = [k for k, s in enumerate(segs) if len(s) >= 16][:1500]
        ed, ov, kept = ctx.cyp_weight_segments(ctx.upload(ccons), np.ones(len(ccons), np.uint8), ctx.upload([segs[k] for k in keep]))
        np.savez_compressed(os.path.join(OUT, f"k3_{i}.npz"), name=name, hits=hits, cons=np.array(ccons), labels=np.array([f"{t}:{s}" for t, s in labels]),
                            seg_of=np.array(keep), ed=ed, ov=ov, kept=kept, call=np.array([call.hap1.decode(), call.hap2.decode()]), expected=np.array(expected))
        print(name, "hits", len(hits), "consensuses", len(ccons), "call", call.hap1.decode(), "/", call.hap2.decode(), flush=True)
    print("audit dump complete:", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
