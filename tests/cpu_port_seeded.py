"""The reference's CPU path for one HLA sample in its own CALL PATTERN (test infrastructure: bench.py's cpu_baseline leg and the tests).

What the reference does per sample (src/hla/caller.rs:510-1083) and what stands in for it here:
  * per read ONE seeded map against the index of all DNA alleles, base-level alignment of the best chains only (`best_n 5`), the acceptance
    of realign_record (src/hla/realigner.rs:98-146): oracle/mm2.c (minimap2's published algorithm restated; minimap2 itself is not on disk);
    then the segment / offset bookkeeping of realign_record (:226-325) with the oracle's routines
  * per gene the dual consensus on the compressed reads, the group consensuses (waffle_con: oracle/consensus.c)
  * per consensus score_consensus + score_read: the consensus placed on the gene reference, spliced, and EVERY allele of the gene mapped to it at
    cDNA and DNA level with a = 5, running best by HlaProcessedMatch (src/hla/caller.rs:1258-1511): oracle/mm2.c + oracle/hla.c
It is a scalar port (no SSE in the DP, where minimap2 has it): a reported baseline, see DESIGN.md section 11."""
import ctypes as C
import multiprocessing as mp
import os
import time

import numpy as np

G = {}
OPMAP = {"=": 7, "X": 8, "I": 1, "D": 2}


def realign_pick(hits):
    """src/hla/realigner.rs:124-146"""
    best, bs = None, None
    for h in hits:
        tl = h["t_len"]; um = tl - (h["t_end"] - h["t_start"]); nm = h["nm"]
        if (nm + um) / tl <= 0.5 and max(nm, 0.1) / (tl - um) <= 0.03:
            s = max(nm, 0.1) / (tl - um)
            if bs is None or s < bs:
                best, bs = h, s
    return best


def select_best_mapping(hits, unmapped_from_target, penalize=True):
    """src/util/mapping.rs:22-57: the default is a 100 % mismatch rate (score 1.0), strict <: the first of equals stays"""
    best, bs = None, 1.0
    for h in hits:
        bl = h["t_len"] if unmapped_from_target else h["q_len"]
        um = bl - ((h["t_end"] - h["t_start"]) if unmapped_from_target else (h["q_end"] - h["q_start"]))
        s = max(float(h["nm"] + um), 0.1) / float(bl) if penalize else max(float(h["nm"]), 0.1) / float(bl - um)
        if s < bs:
            best, bs = h, s
    return best


def record_mm2(mm, o, fx, read, a, m, rev, am_cache):
    """realign_record behind the accepted mapping (src/hla/realigner.rs:214-337) IN THE REFERENCE'S CALL PATTERN: the accepted mapping's read segment +- 1,000 bases is mapped
    to the gene's reference (gene_aligner.map, :231) and the best mapping picked by select_best_mapping (unmapped from the target, penalised, :238-242); where the reference
    mapping does not start in front of the allele mapping the ALLELE is mapped to the reference as well (:290, Forward only, unmapped from the query) and the two offsets add
    up (:307-317).  m = (nm, t_start, t_end, q_start, q_end, t_len, q_len) of the accepted mapping, a = its allele (-1: none; rev: a best mapping on the reverse strand)"""
    res = dict(status=1, best_allele=-1, gene=-1)
    if a < 0:
        if rev:
            res["status"] = 2
        return res
    nm, ts, te, qs, qe, tl, _ql = m
    g = int(fx.gene_of[a])
    res.update(status=3, best_allele=a, gene=g, nm=int(nm), target_len=int(tl), unmapped=int(tl - (te - ts)))
    ref = fx.gene_ref[g]
    db_s, db_e = int(qs), int(qe)
    buf_s, buf_e = max(db_s - 1000, 0), min(db_e + 1000, len(read))
    rm = select_best_mapping(mm.map_pair(ref, read[buf_s:buf_e]), unmapped_from_target=True)
    if rm is None or rm["rev"]:                                               # "Remapping ... failed" / "was to Reverse strand, ignoring" (:326-337)
        return res
    adj_s, adj_e = buf_s + rm["q_start"], buf_s + rm["q_end"]
    if adj_s < db_s:
        d = rm["t_start"]
        h = o.hpc_pos(ref, d)
    else:
        if a not in am_cache:
            am_cache[a] = select_best_mapping([x for x in mm.map_pair(ref, fx.dna_fwd(a)) if not x["rev"]], unmapped_from_target=False)
        am = am_cache[a]
        if am is not None:
            added = max(am["t_start"] - am["q_start"], 0)
            d = added + int(ts)
            h = o.hpc_pos(ref, added) + o.hpc_pos(fx.dna[a], int(ts))
        else:
            d = rm["t_start"]
            h = o.hpc_pos(ref, d)
    res.update(status=0, seg_start=min(db_s, adj_s), seg_end=max(db_e, adj_e), dna_offset=int(d), hpc_offset=int(h))
    return res


def _k1_worker(args):
    """realign_record for a share of the reads: the seeded map + acceptance (src/hla/realigner.rs:98-146), then the segment / offset bookkeeping behind the best
    mapping (:226-325)"""
    import hla_expected as hx
    lo, hi, budget_s = args
    idx, reads, dna_ids, o = G["idx"], G["reads"], G["dna_ids"], G["o"]
    contract = os.environ.get("SP_PORT_K1_SECOND_STAGE", "mm2") == "contract"        # (the second stage on the library's own alignment contract, as until round 5)
    tb = None
    if contract:
        tb = G.get("tb") or hx.K1Tables(o, G["fx"], G["off"])
        G["tb"] = tb
    am_cache = G.setdefault("am_cache", {})
    out, t0, t_map = [], time.perf_counter(), 0.0
    for r in range(lo, hi):
        t1 = time.perf_counter()
        h = realign_pick(idx.map(reads[r]))
        t_map += time.perf_counter() - t1
        # a best mapping on the reverse strand is dropped (src/hla/realigner.rs:178-193)
        a, m = (-1, None) if (h is None or h["rev"]) else (dna_ids[h["rid"]], (h["nm"], h["t_start"], h["t_end"], h["q_start"], h["q_end"], h["t_len"], h["q_len"]))
        if contract:
            re = o.encode(reads[r])
            bm = None
            if a >= 0:
                nm, ts, te, qs, qe, tl, ql = m
                bm = np.zeros(1, hx.oracle_aln_dtype())[0]
                bm["ok"], bm["nm"], bm["a_start"], bm["a_end"], bm["b_start"], bm["b_end"], bm["a_len"], bm["b_len"] = 1, nm, ts, te, qs, qe, tl, ql
            rec = tb.record(reads[r], re, tb.anchors(re), a, bm)
            rec.pop("aln", None)
        else:
            rec = record_mm2(G["mm"], o, G["fx"], reads[r], a, m, bool(h is not None and h["rev"]), am_cache)
        out.append((r, a, m, rec))
        if time.perf_counter() - t0 > budget_s:
            break
    return out, time.perf_counter() - t0, t_map


def k2_scan_mm2(mm, o, alleles, cons_dna, cons_cdna, stats_out=None):
    """score_read (src/hla/caller.rs:1411-1510) with the restatement's mappings, in C (omm_hla_score_read): alleles = [(index, cdna, dna)] in
    database order.  stats_out (a list): receives the per-allele HlaMappingStats the reference prints into hla_debug.json -- an int64 array
    [allele][level: cDNA, DNA][len, nm, unmapped], -1 where a level has no mapping"""
    n = len(alleles)
    enc = [(o.encode(cd) if cd else None, o.encode(dn) if dn else None) for _a, cd, dn in alleles]
    cp, dp = (C.c_void_p * max(1, n))(), (C.c_void_p * max(1, n))()
    cl, dl = np.zeros(max(1, n), np.int32), np.zeros(max(1, n), np.int32)
    for i, (c, d) in enumerate(enc):
        if c is not None:
            cp[i], cl[i] = c.ctypes.data, len(c)
        if d is not None:
            dp[i], dl[i] = d.ctypes.data, len(d)
    cc, cd_ = o.encode(cons_cdna), o.encode(cons_dna)
    opts5 = mm.opts(a=5)
    L = o.L
    L.omm_hla_score_read.restype = C.c_int32
    L.omm_hla_score_read.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    st = np.zeros((max(1, n), 2, 3), np.int64) if stats_out is not None else None
    b = L.omm_hla_score_read(cc.ctypes.data if len(cc) else None, len(cc), cd_.ctypes.data if len(cd_) else None, len(cd_), n, cp, cl.ctypes.data, dp, dl.ctypes.data,
                             C.byref(opts5), st.ctypes.data if st is not None else None)
    if stats_out is not None:
        stats_out.append(st[:n])
    return alleles[b][0] if b >= 0 else -1


def type_consensus_mm2(o, fx, g, cons, synth, stats_out=None):
    """score_consensus + splice_read + score_read (src/hla/caller.rs:1258-1319,1332-1511,1518-1576) on the restatement's mappings.
    stats_out (a list): receives ([allele indices in database order], the per-allele stats of k2_scan_mm2)"""
    import mm2_ffi
    import oracle_ffi
    if not cons:
        return -1
    mm = G.get("mm") or mm2_ffi.Mm2(o)
    ref = fx.gene_ref[g][fx.buffer:len(fx.gene_ref[g]) - fx.buffer]
    best, bs = None, 1.0                                        # select_best_mapping(unmapped from the target, penalised): caller.rs:1290-1294
    for h in mm.map_pair(ref, cons, max_hits=8):
        s = max(float(h["nm"] + (h["t_len"] - (h["t_end"] - h["t_start"]))), 0.1) / float(h["t_len"])
        if s < bs:
            best, bs = h, s
    if best is None or best["rev"]:
        return -1
    bam = [(ln, {"=": 0, "X": 0, "I": 1, "D": 2}[op]) for ln, op in best["cigar"]]
    if best["q_start"]:
        bam.insert(0, (best["q_start"], 4))
    exons = [(e0 - fx.buffer, e1 - fx.buffer) for e0, e1 in fx.exons[g]]
    segs, _ = o.splice_read(best["t_start"], bam, exons)
    spliced = "".join(cons[x:y] for x, y in segs)
    fwd = bool(fx.gene_fwd[g])
    e_dna = cons if fwd else synth.revcomp(cons)
    e_cdna = spliced if fwd else synth.revcomp(spliced)
    alleles = [(a, fx.cdna[a], fx.dna[a]) for a in range(len(fx.ids)) if fx.gene_of[a] == g]
    if stats_out is None:
        return k2_scan_mm2(mm, o, alleles, e_dna, e_cdna)
    st = []
    best = k2_scan_mm2(mm, o, alleles, e_dna, e_cdna, stats_out=st)
    stats_out.append(([a for a, _c, _d in alleles], st[0]))
    return best


def _gene_worker(args):
    g, sample = args
    import hla_expected as hx
    import hla_pipeline as hp
    from pb_starphase_amd import synth
    o, fx = G["o"], G["fx"]
    reads = [G["reads"][r] for r in sample]
    t0 = time.perf_counter()
    k1 = [G["records"][r] for r in sample]                     # (realign_record's bookkeeping ran with the maps, spread over the workers)
    t1 = time.perf_counter()
    typed = []

    def typer(oracle, fx_, g_, cons, synth_):
        t = time.perf_counter()
        out = type_consensus_mm2(oracle, fx_, g_, cons, synth_)
        typed.append((t, time.perf_counter()))
        return out
    typer.threads = True                                       # the two consensuses of a gene are typed side by side

    res = hp.diplotype_gene(o, fx, g, reads, k1, synth, type_fn=typer)
    t2 = time.perf_counter()
    t_type = sum(b - a for a, b in typed)                                  # CPU seconds of the typing; its wall clock (two threads) is the span they cover
    t_type_wall = (max(b for _a, b in typed) - min(a for a, _b in typed)) if typed else 0.0
    return g, t1 - t0, (t2 - t1) - t_type_wall, t_type, (res["allele1"], res["allele2"]), (res["cons1"], res["cons2"])


def run(o, fx, reads, n_sample=2000, budget_s=12.0, cores=None, seed=0):
    """-> dict for bench.py's cpu_baseline block, {read: best allele} and {gene: (allele1, allele2)} of the sample"""
    import mm2_ffi
    mm = mm2_ffi.Mm2(o)
    cores = cores or max(1, min(len(os.sched_getaffinity(0)), 128))
    rng = np.random.default_rng(seed)
    sample = sorted(rng.choice(len(reads), min(n_sample, len(reads)), replace=False).tolist())
    t0 = time.perf_counter()
    dna_ids = [a for a in range(len(fx.ids)) if fx.dna[a]]
    idx = mm2_ffi.Index(mm, [fx.dna_fwd(a) for a in dna_ids])
    t_index = time.perf_counter() - t0
    off = np.full(len(fx.ids), -2 ** 31, np.int32)              # frame offsets of the alleles on their gene reference (HlaRealigner::new; untimed set-up)
    refs = [o.encode(s) for s in fx.gene_ref]
    for a in dna_ids:
        d, v = o.anchor(refs[int(fx.gene_of[a])], o.encode(fx.dna_fwd(a)))
        if v >= 16:
            off[a] = d
    G.update(o=o, mm=mm, fx=fx, reads=reads, idx=idx, dna_ids=dna_ids, off=off)
    sub = [reads[r] for r in sample]
    G["reads"] = reads
    # K1, one thread: the reference's own concurrency (src/cli/diplotype.rs:185-191)
    t1 = time.perf_counter()
    single = []
    for r in sample[:64]:
        single.append(realign_pick(idx.map(reads[r])))
    t_single = (time.perf_counter() - t1) / max(1, len(single))
    # K1, every core
    per = (len(sample) + cores - 1) // cores
    G["reads"] = sub
    t1 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        parts = pool.map(_k1_worker, [(w * per, min(len(sub), (w + 1) * per), budget_s) for w in range(cores) if w * per < len(sub)])
    t_k1 = time.perf_counter() - t1
    best, records = {}, {}
    for out, _dt, _tm in parts:
        for r, a, m, rec in out:
            best[sample[r]] = (a, m); records[sample[r]] = rec
    done = sorted(best)
    G["reads"], G["best"], G["records"] = reads, best, records
    map_cpu_s, k1_cpu_s = sum(p[2] for p in parts), sum(p[1] for p in parts)
    # per gene: consensus + typing (one worker per gene; inside a gene the reference is sequential)
    t1 = time.perf_counter()
    with mp.get_context("fork").Pool(len(fx.genes)) as pool:
        gres = pool.map(_gene_worker, [(g, done) for g in range(len(fx.genes))])
    t_genes = time.perf_counter() - t1
    idx.close()
    calls = {g: c for g, _a, _b, _c, c, _s in gres}
    cons = {g: s for g, _a, _b, _c, _d, s in gres}
    cons_s = sum(x[2] for x in gres); typing_s = sum(x[3] for x in gres)
    bookkeeping_s = k1_cpu_s - map_cpu_s
    n = len(done)
    one_thread_s = n * t_single + bookkeeping_s + cons_s + typing_s
    k1_workers = min(cores, (len(sub) + per - 1) // per)
    return {"value": n / (t_k1 + t_genes), "unit": "reads/s", "cores": cores, "kind": "port", "reads": n,
            "wall_s": t_k1 + t_genes, "single_thread_value": n / one_thread_s, "one_thread_s": one_thread_s,
            "workers_per_stage": {"realign_record (seeded map + segment / offsets)": k1_workers, "consensus (per gene)": len(fx.genes),
                                  "typing every allele (two consensuses per gene side by side)": 2 * len(fx.genes)},
            "k1_seeded_ms_per_read_one_thread": 1e3 * t_single, "k1_all_cores_s": t_k1, "genes_wall_s": t_genes,
            "cpu_s": {"seeded_maps": map_cpu_s, "realign_bookkeeping": bookkeeping_s, "consensus": cons_s, "typing_every_allele": typing_s},
            "index_build_s": t_index,
            "sample": f"{n} of the sample's {len(reads)} HLA reads: realign_record (one seeded map per read against the index of {len(dna_ids)} DNA alleles, best_n 5, + segment / "
                      f"offset bookkeeping) over {k1_workers} forked workers in {t_k1:.1f} s ({1e3 * t_single:.1f} ms per map on one thread); then per gene (one worker each, the two "
                      f"consensuses of a gene typed on two threads) dual + group consensus and the typing of the consensuses against every allele in {t_genes:.1f} s wall",
            "note": "the reference's call pattern on minimap2's published algorithm restated in scalar C (oracle/mm2.c; minimap2 itself and its SSE kernels are not on disk); "
                    "consensus = oracle/consensus.c.  single_thread_value is what one thread needs for the same sample (the reference is single-threaded)"}, best, calls, cons, done
