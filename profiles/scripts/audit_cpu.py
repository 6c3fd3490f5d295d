"""Aligner audit, CPU half (no GPU needed): the minimap2 restatement of the oracle (oracle/mm2.c) run beside what the library's alignment contract
gave on the GPU (gpurun_out/audit/*.npz, written by profiles/scripts/audit_gpu.py).  VERDICT round 2, item 1(a).

Writes profiles/r03/aligner_divergence.json and .md: for K1 (read -> allele), K2 (allele -> consensus, a = 5), K3 (template -> read) and K4
(consensus -> segment) the fraction of pairs on which nm, the unmapped bases or the WINNER differ, split by the divergence classes of DESIGN.md 3.4."""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402

AUD = os.path.join(ROOT, "gpurun_out", "audit")
OUT = os.path.join(ROOT, "profiles", "r03")
G = {}


def score(nm, span):
    return max(float(nm), 0.1) / float(span) if span > 0 else 1.0


def realign_pick(hits, forward_only=True):
    """the mapping realign_record keeps (src/hla/realigner.rs:124-146); strand handled as :180-193 (best mapping on the reverse strand => dropped)"""
    best, bs = None, None
    for h in hits:
        tl = h["t_len"]; um = tl - (h["t_end"] - h["t_start"]); nm = h["nm"]
        if (nm + um) / tl <= 0.5 and max(nm, 0.1) / (tl - um) <= 0.03:
            s = max(nm, 0.1) / (tl - um)
            if bs is None or s < bs:
                best, bs = h, s
    return best


def _init():
    import mm2_ffi
    import oracle_ffi
    G["o"] = oracle_ffi.load()
    G["mm"] = mm2_ffi.Mm2(G["o"])
    G["mm2_ffi"] = mm2_ffi


# ------------------------------------------------------------------------------------------------------------------------ K1
def _k1_pairs(args):
    lo, hi = args
    mm, fwd, reads, k1 = G["mm"], G["fwd"], G["reads"], G["k1"]
    out = []
    for r in range(lo, hi):
        rec = {"r": r}
        for tag, a in (("win", int(k1["winner"][r])), ("run", int(k1["runner"][r]))):
            if a < 0 or k1["status"][r] != 0:
                continue
            h = realign_pick(mm.map_pair(fwd[a], reads[r], G["opts_fwd"]))
            rec[tag] = (a, h["nm"], h["t_end"] - h["t_start"], h["t_start"], h["t_len"] - h["t_end"]) if h else (a, -1, 0, 0, 0)
        out.append(rec)
    return out


def _k1_seeded(args):
    lo, hi = args
    idx, reads, dna_ids = G["idx"], G["reads"], G["dna_ids"]
    out = []
    for r in range(lo, hi):
        t0 = time.perf_counter()
        hits = idx.map(reads[r])
        dt = time.perf_counter() - t0
        h = realign_pick(hits)
        best_any = min(hits, key=lambda x: (score(x["nm"], x["t_end"] - x["t_start"]), )) if hits else None
        out.append((r, dna_ids[h["rid"]] if h else -1, h["nm"] if h else -1, (h["t_end"] - h["t_start"]) if h else 0, h["rev"] if h else 0,
                    [dna_ids[x["rid"]] for x in hits], dt, best_any["rev"] if best_any else 0))
    return out


def audit_k1(pkg, synth, pool, n_reads=None):
    import hla_expected as hx
    fx = synth.HlaFixture()
    wl = synth.Config2Workload(fx, n_reads=10000, seed=1000)
    k1 = dict(np.load(os.path.join(AUD, "k1.npz")))
    n = len(wl.reads) if n_reads is None else n_reads
    G.update(reads=wl.reads, k1=k1, fwd=[fx.dna_fwd(a) if fx.dna[a] else "" for a in range(len(fx.ids))])
    G["opts_fwd"] = G["mm"].opts()
    chunks = [(i, min(n, i + 50)) for i in range(0, n, 50)]
    t0 = time.time()
    recs = [x for part in pool.map(_k1_pairs, chunks) for x in part]
    t_pairs = time.time() - t0
    res = {"reads": n, "reads_realigned_by_k1": int((k1["status"][:n] == 0).sum()), "pairs": 0, "identical_nm_and_span": 0, "nm_equal_span_differs": 0,
           "nm_differs_span_equal": 0, "both_differ": 0, "mm2_no_accepted_mapping": 0, "nm_delta_hist": {}, "span_delta_hist": {},
           "order_pairs": 0, "order_preserved": 0, "order_tied_under_mm2": 0, "order_flipped": 0, "flipped_examples": []}
    for rec in recs:
        r = rec["r"]
        sc = {}
        for tag, nmk, spk in (("win", "win_cell_nm", "win_cell_span"), ("run", "run_nm", "run_span")):
            if tag not in rec:
                continue
            a, nm2, sp2, cs, ce = rec[tag]
            nm1, sp1 = int(k1[nmk][r]), int(k1[spk][r])
            res["pairs"] += 1
            if nm2 < 0:
                res["mm2_no_accepted_mapping"] += 1
                continue
            sc[tag] = (score(nm1, sp1), score(nm2, sp2))
            dn, ds = nm2 - nm1, sp2 - sp1
            key = "identical_nm_and_span" if (dn == 0 and ds == 0) else "nm_equal_span_differs" if dn == 0 else "nm_differs_span_equal" if ds == 0 else "both_differ"
            res[key] += 1
            res["nm_delta_hist"][str(dn)] = res["nm_delta_hist"].get(str(dn), 0) + 1
            b = str(ds) if abs(ds) <= 8 else ("<-8" if ds < 0 else ">8")
            res["span_delta_hist"][b] = res["span_delta_hist"].get(b, 0) + 1
        if "win" in sc and "run" in sc:
            res["order_pairs"] += 1
            w2, r2 = sc["win"][1], sc["run"][1]
            if w2 < r2:
                res["order_preserved"] += 1
            elif w2 == r2:
                res["order_tied_under_mm2"] += 1
            else:
                res["order_flipped"] += 1
                if len(res["flipped_examples"]) < 20:
                    res["flipped_examples"].append({"read": r, "winner": rec["win"][:3], "runner": rec["run"][:3],
                                                    "k1": [int(k1["win_cell_nm"][r]), int(k1["win_cell_span"][r]), int(k1["run_nm"][r]), int(k1["run_span"][r])]})
    res["seconds_pairs"] = t_pairs
    # the reference's call pattern: one seeded map of the read against the index of all DNA alleles, base-level alignment of the best chains only
    mm2_ffi = G["mm2_ffi"]
    dna_ids = [a for a in range(len(fx.ids)) if fx.dna[a]]
    G["dna_ids"] = dna_ids
    G["idx"] = mm2_ffi.Index(G["mm"], [fx.dna_fwd(a) for a in dna_ids])
    with mp.get_context("fork").Pool(pool._processes) as p2:            # (forked after the index exists)
        t0 = time.time()
        seeded = [x for part in p2.map(_k1_seeded, chunks) for x in part]
        t_seed = time.time() - t0
    o = G["o"]
    tables = hx.K1Tables(o, fx)
    s = {"reads": n, "same_winner": 0, "other_allele_same_ratio_under_contract": 0, "other_allele_worse_under_contract": 0, "other_allele_better_under_contract": 0,
         "seeded_none_k1_some": 0, "k1_none_seeded_some": 0, "both_none": 0, "k1_winner_among_aligned_chains": 0, "worse_examples": [],
         "same_gene": 0, "both_found": 0, "k1_winner_shorter_than_seeded_winner": 0, "seeded_winner_is_truth_allele": 0, "k1_winner_is_truth_allele": 0,
         "cpu_seconds_per_read_single_thread": float(np.mean([x[6] for x in seeded])), "wall_seconds": t_seed, "index_mid_occ": G["idx"].mid_occ}
    for (r, a2, nm2, sp2, rev, cand, _dt, _brev) in seeded:
        a1 = int(k1["winner"][r]) if k1["status"][r] == 0 else -1
        if a1 >= 0 and a2 >= 0:
            s["both_found"] += 1
            s["same_gene"] += int(fx.gene_of[a1] == fx.gene_of[a2])
            s["k1_winner_shorter_than_seeded_winner"] += int(len(fx.dna[a1]) < len(fx.dna[a2]))
        s["seeded_winner_is_truth_allele"] += int(a2 == wl.read_truth[r][1])
        s["k1_winner_is_truth_allele"] += int(a1 == wl.read_truth[r][1])
        if a1 < 0 and a2 < 0:
            s["both_none"] += 1
        elif a2 < 0:
            s["seeded_none_k1_some"] += 1
        elif a1 < 0:
            s["k1_none_seeded_some"] += 1
        elif a1 == a2:
            s["same_winner"] += 1
        else:
            re_ = o.encode(wl.reads[r])
            anch = tables.anchors(re_)
            c2 = tables.cell(a2, re_, anch)                                        # the seeded winner under the library's contract
            sc1 = score(int(k1["win_cell_nm"][r]), int(k1["win_cell_span"][r]))
            sc2 = score(c2.nm, c2.a_end - c2.a_start) if c2 is not None and c2.ok else 1.0
            key = "other_allele_same_ratio_under_contract" if sc2 == sc1 else "other_allele_worse_under_contract" if sc2 > sc1 else "other_allele_better_under_contract"
            s[key] += 1
            if key != "other_allele_same_ratio_under_contract" and len(s["worse_examples"]) < 20:
                s["worse_examples"].append({"read": r, "k1": [a1, int(k1["win_cell_nm"][r]), int(k1["win_cell_span"][r])], "seeded": [a2, nm2, sp2],
                                            "seeded_under_contract": [int(c2.nm), int(c2.a_end - c2.a_start)] if c2 is not None and c2.ok else None})
        if a1 >= 0 and a1 in cand:
            s["k1_winner_among_aligned_chains"] += 1
    G["idx"].close()
    return res, s


# ------------------------------------------------------------------------------------------------------------------------ K2
def _k2_alleles(args):
    ci, lo, hi = args
    mm, o = G["mm"], G["o"]
    idxs = G["k2_idx"][ci]
    al = G["k2_alleles"][ci]
    out = []
    for j in range(lo, hi):
        a, cd, dn = al[j]
        rec = [a]
        for lv, seq in ((0, cd), (1, dn)):
            if not seq or idxs[lv] is None:
                rec.append(None); continue
            hits = [h for h in idxs[lv].map(seq, want_cigar=True) if h["rev"] == 0]       # caller.rs:1443-1445: Forward only
            best, bs = None, 1.0                                                         # select_best_mapping, query based, penalised
            for h in hits:
                s = max(float(h["nm"] + (h["q_len"] - (h["q_end"] - h["q_start"]))), 0.1) / float(h["q_len"])
                if s < bs:
                    best, bs = h, s
            if best is None:
                rec.append(None); continue
            rec.append((best["q_len"], best["nm"], best["q_len"] - (best["q_end"] - best["q_start"]), best["t_start"], best["t_end"], best["q_start"],
                        best["q_len"] - best["q_end"], [(ln, {"=": 7, "X": 8, "I": 1, "D": 2}[op]) for ln, op in best["cigar"]]))
        out.append(rec)
    return out


def audit_k2(pkg, synth, pool):
    import ctypes as C
    import oracle_ffi
    fx = synth.HlaFixture()
    k2 = dict(np.load(os.path.join(AUD, "k2.npz")))
    mm, o, mm2_ffi = G["mm"], G["o"], G["mm2_ffi"]
    res = {"consensuses": int(len(k2["cons"])), "per_consensus": []}
    opts5 = mm.opts(a=5)
    G["k2_idx"], G["k2_alleles"] = {}, {}
    for ci in range(len(k2["cons"])):
        g = int(k2["gene"][ci])
        cons_fwd, cdna = str(k2["cons"][ci]), str(k2["cdna"][ci])
        cons_gene = cons_fwd if fx.gene_fwd[g] else synth.revcomp(cons_fwd)
        G["k2_idx"][ci] = (mm2_ffi.Index(mm, [cdna], opts5) if cdna else None, mm2_ffi.Index(mm, [cons_gene], opts5))
        G["k2_alleles"][ci] = [(a, fx.cdna[a], fx.dna[a]) for a in range(len(fx.ids)) if fx.gene_of[a] == g]
    with mp.get_context("fork").Pool(pool._processes) as p2:
        for ci in range(len(k2["cons"])):
            al = G["k2_alleles"][ci]
            t0 = time.time()
            recs = [x for part in p2.map(_k2_alleles, [(ci, i, min(len(al), i + 200)) for i in range(0, len(al), 200)]) for x in part]
            dt = time.time() - t0
            st = k2["stats"][ci]
            clen = [len(str(k2["cdna"][ci])), len(str(k2["cons"][ci]))]
            lvl = [{"pairs": 0, "identical": 0, "nm_differs": 0, "unmapped_differs": 0, "one_side_missing": 0, "close_pairs": 0, "close_identical": 0, "max_abs_nm_delta": 0,
                    "nm_delta_hist": {}} for _ in range(2)]
            # running best over the alleles in database order with the minimap2 restatement's mappings (score_read's own scan)
            best_levels, best_keep, best_a = (oracle_ffi.HlaLevel * 2)(), [None, None], -1
            for rec in recs:
                a = rec[0]
                cur, keep = (oracle_ffi.HlaLevel * 2)(), [None, None]
                for lv in range(2):
                    m = rec[1 + lv]
                    s1 = st[a][3 * lv:3 * lv + 3]
                    have1 = s1[0] >= 0
                    if m is None and not have1:
                        continue
                    lvl[lv]["pairs"] += 1
                    close = have1 and int(s1[1]) + int(s1[2]) <= 30                 # within 30 edits of the consensus: the alleles that compete for the call
                    lvl[lv]["close_pairs"] += int(close)
                    if m is not None and have1:
                        dnm = m[1] - int(s1[1])
                        lvl[lv]["max_abs_nm_delta"] = max(lvl[lv]["max_abs_nm_delta"], abs(dnm))
                        lvl[lv]["nm_delta_hist"][str(dnm)] = lvl[lv]["nm_delta_hist"].get(str(dnm), 0) + 1
                        lvl[lv]["close_identical"] += int(close and (m[1], m[2]) == (int(s1[1]), int(s1[2])))
                    if (m is None) != (not have1):
                        lvl[lv]["one_side_missing"] += 1
                    elif (m[1], m[2]) == (int(s1[1]), int(s1[2])):
                        lvl[lv]["identical"] += 1
                    else:
                        lvl[lv]["nm_differs"] += int(m[1] != int(s1[1]))
                        lvl[lv]["unmapped_differs"] += int(m[2] != int(s1[2]))
                    if m is not None:
                        qlen, nm, um, ts, te, cs, ce, cg = m
                        pc = np.array(o.process_mm_cigar(cg, ts, clen[lv], cs, ce), np.uint64)
                        keep[lv] = pc
                        cur[lv].present = 1
                        cur[lv].range_start = max(ts - cs, 0)
                        cur[lv].range_end = te + min(ce, clen[lv] - te)
                        cur[lv].len, cur[lv].nm, cur[lv].unmapped = qlen, nm, um
                        cur[lv].pc = pc.ctypes.data_as(C.POINTER(C.c_uint64))
                if o.L.osp_is_better_match(C.byref(cur), C.byref(best_levels)):
                    best_levels, best_keep, best_a = cur, keep, a
            b1 = int(k2["best"][ci])
            same_seq = b1 == best_a or (b1 >= 0 and best_a >= 0 and fx.cdna[b1] == fx.cdna[best_a] and fx.dna[b1] == fx.dna[best_a])
            res["per_consensus"].append({"gene": fx.genes[int(k2["gene"][ci])], "alleles": len(recs), "cdna": lvl[0], "dna": lvl[1], "winner_contract": b1,
                                         "winner_mm2": best_a, "winner_identical": bool(b1 == best_a), "winner_same_sequences": bool(same_seq), "seconds": dt})
    for v in G["k2_idx"].values():
        for ix in v:
            if ix is not None:
                ix.close()
    return res


# ------------------------------------------------------------------------------------------------------------------------ K3 / K4
PENALISED_TYPES = None


def _k3_reads(args):
    si, lo, hi = args
    mm, mm2_ffi = G["mm"], G["mm2_ffi"]
    reads, tm = G["cyp_reads"][si], G["templates"]
    out = []
    for r in range(lo, hi):
        idx = mm2_ffi.Index(mm, [reads[r]])
        hits = []
        for t, (typ, sub, full, seq, deep) in enumerate(tm):
            for h in idx.map(seq):
                if h["rev"]:
                    continue
                hits.append((t, h["t_start"], h["t_end"], h["nm"], h["q_len"] - (h["q_end"] - h["q_start"]), h["q_start"], h["q_len"] - h["q_end"]))
        idx.close()
        out.append((r, hits))
    return out


def _k4_segments(args):
    si, lo, hi = args
    mm, mm2_ffi = G["mm"], G["mm2_ffi"]
    segs, cons = G["cyp_segs"][si], G["cyp_cons"][si]
    out = []
    for s in range(lo, hi):
        idx = mm2_ffi.Index(mm, [segs[s]])
        row = []
        for c in cons:
            best = (len(segs[s]), 0.0)
            for h in idx.map(c):
                ed = h["nm"] + (len(segs[s]) - (h["t_end"] - h["t_start"]))
                ov = 1.0 - (h["q_start"] + (h["q_len"] - h["q_end"])) / float(h["q_len"])
                if ed < best[0] or (ed == best[0] and ov > best[1]):
                    best = (ed, ov)
            row.append(best)
        idx.close()
        out.append((s, row))
    return out


def audit_cyp(pkg, synth, pool):
    import cyp_cases_real as cr
    cfg, gene_def = cr.load_db()
    locus = synth.Chr22Locus(cfg, gene_def, seed=3)
    import oracle_ffi  # noqa: F401
    # the 39 templates as the library builds them come from the GPU dump's hits' template_idx; rebuild them with the independent Python statement
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cyp_db
    hyb = cyp_db.generate_cyp_hybrids(locus.slice, cfg)
    tm = [(k[0], k[1], cyp_db.full_allele(*k), hyb[k], False) for k in cyp_db.template_order(hyb)]      # the library's visiting order (tests/test_cyp_db.py checks it)
    G["templates"] = tm
    G["cyp_reads"], G["cyp_segs"], G["cyp_cons"] = {}, {}, {}
    dumps = {}
    for si, (name, haps, expected) in enumerate(cr.scenarios(locus)):
        d = dict(np.load(os.path.join(AUD, f"k3_{si}.npz")))
        reads = locus.sample(np.random.default_rng(7), haps, 2000)
        nsub = int(d["hits"]["read"].max()) + 1 if len(d["hits"]) else 0
        G["cyp_reads"][si] = reads[:nsub]
        hits = d["hits"]
        segs_all = [reads[int(h["read"])][int(h["start"]):int(h["end"])] for h in hits]
        G["cyp_segs"][si] = [segs_all[k] for k in d["seg_of"]]
        G["cyp_cons"][si] = [str(c) for c in d["cons"]]
        dumps[si] = (name, d, nsub)
    k3 = {"scenarios": {}}
    k4 = {"scenarios": {}}
    with mp.get_context("fork").Pool(pool._processes) as p2:
        for si, (name, d, nsub) in dumps.items():
            t0 = time.time()
            per_read = dict(x for part in p2.map(_k3_reads, [(si, i, min(nsub, i + 10)) for i in range(0, nsub, 10)]) for x in part)
            s = {"reads": nsub, "library_hits": int(len(d["hits"])), "found_by_mm2": 0, "same_start_end": 0, "same_nm": 0, "same_unmapped": 0, "same_all": 0,
                 "not_found_by_mm2": 0, "abs_nm_delta_sum": 0, "missing_examples": []}
            for h in d["hits"]:
                r, t = int(h["read"]), int(h["template_idx"])
                best, bov = None, 0
                for (t2, ts, te, nm, um, cs, ce) in per_read.get(r, []):
                    if t2 != t:
                        continue
                    ov = min(te, int(h["end"])) - max(ts, int(h["start"]))
                    if ov > bov:
                        best, bov = (ts, te, nm, um, cs, ce), ov
                if best is None:
                    s["not_found_by_mm2"] += 1
                    if len(s["missing_examples"]) < 5:
                        s["missing_examples"].append({k: int(h[k]) for k in h.dtype.names})
                    continue
                s["found_by_mm2"] += 1
                se = (best[0], best[1]) == (int(h["start"]), int(h["end"]))
                sn, su = best[2] == int(h["nm"]), best[3] == int(h["unmapped"])
                s["same_start_end"] += se; s["same_nm"] += sn; s["same_unmapped"] += su; s["same_all"] += (se and sn and su)
                s["abs_nm_delta_sum"] += abs(best[2] - int(h["nm"]))
            s["seconds"] = time.time() - t0
            k3["scenarios"][name] = s
            # K4
            t0 = time.time()
            ns = len(G["cyp_segs"][si])
            rows = dict(x for part in p2.map(_k4_segments, [(si, i, min(ns, i + 20)) for i in range(0, ns, 20)]) for x in part)
            w = {"segments": ns, "consensuses": len(G["cyp_cons"][si]), "pairs": 0, "same_ed": 0, "same_ed_and_overlap": 0, "same_argmin_set": 0, "abs_ed_delta_sum": 0, "library_default_mm2_mapped": 0, "mm2_default_library_mapped": 0,
                 "abs_ed_delta_sum_both_mapped": 0, "min_ed_pairs": 0, "min_ed_pairs_same_ed": 0}
            for sidx in range(ns):
                lib_ed = d["ed"][sidx]; lib_ov = d["ov"][sidx]
                m = rows[sidx]
                for c in range(len(m)):
                    w["pairs"] += 1
                    w["same_ed"] += int(m[c][0] == int(lib_ed[c]))
                    w["same_ed_and_overlap"] += int(m[c][0] == int(lib_ed[c]) and abs(m[c][1] - float(lib_ov[c])) < 1e-12)
                    w["abs_ed_delta_sum"] += abs(m[c][0] - int(lib_ed[c]))
                    ldef = int(lib_ed[c]) == len(G["cyp_segs"][si][sidx]) and float(lib_ov[c]) == 0.0
                    mdef = m[c][0] == len(G["cyp_segs"][si][sidx]) and m[c][1] == 0.0
                    w["library_default_mm2_mapped"] += int(ldef and not mdef)
                    w["mm2_default_library_mapped"] += int(mdef and not ldef)
                    if not ldef and not mdef:
                        w["abs_ed_delta_sum_both_mapped"] += abs(m[c][0] - int(lib_ed[c]))
                    if int(lib_ed[c]) == min(int(x) for x in lib_ed):
                        w["min_ed_pairs"] += 1
                        w["min_ed_pairs_same_ed"] += int(m[c][0] == int(lib_ed[c]))
                mn1 = min(int(x) for x in lib_ed); mn2 = min(x[0] for x in m)
                w["same_argmin_set"] += int([c for c in range(len(m)) if int(lib_ed[c]) == mn1] == [c for c in range(len(m)) if m[c][0] == mn2])
            w["seconds"] = time.time() - t0
            k4["scenarios"][name] = w
    return k3, k4


def main():
    os.makedirs(OUT, exist_ok=True)
    ge.build()
    pkg = ge.load_package()
    from pb_starphase_amd import synth
    _init()
    what = sys.argv[1:] or ["k1", "k2", "cyp"]
    workers = int(os.environ.get("AUDIT_WORKERS", str(len(os.sched_getaffinity(0)))))
    n_reads = int(os.environ["AUDIT_K1_READS"]) if "AUDIT_K1_READS" in os.environ else None
    path = os.path.join(OUT, "aligner_divergence.json")
    res = json.load(open(path)) if os.path.exists(path) else {}
    if "k1" in what:
        G["k1"] = None
        # the pool is forked inside audit_k1 once the globals are in place
        fx_ready = True
        import hla_expected  # noqa: F401
        from pb_starphase_amd import synth as _s  # noqa: F401
        # globals first, then fork
        fx = synth.HlaFixture()
        wl = synth.Config2Workload(fx, n_reads=10000, seed=1000)
        G.update(reads=wl.reads, k1=dict(np.load(os.path.join(AUD, "k1.npz"))), fwd=[fx.dna_fwd(a) if fx.dna[a] else "" for a in range(len(fx.ids))], opts_fwd=G["mm"].opts())
        with mp.get_context("fork").Pool(workers) as pool:
            res["k1_pairs"], res["k1_seeded_call_pattern"] = audit_k1(pkg, synth, pool, n_reads)
    if "k2" in what:
        with mp.get_context("fork").Pool(1) as pool:
            pool._processes = workers
            res["k2"] = audit_k2(pkg, synth, pool)
    if "cyp" in what:
        with mp.get_context("fork").Pool(1) as pool:
            pool._processes = workers
            res["k3"], res["k4"] = audit_cyp(pkg, synth, pool)
    json.dump(res, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps({k: (v if k not in ("k2",) else "...") for k, v in res.items()}, indent=1)[:6000])


if __name__ == "__main__":
    main()
