"""Expected K1 / K2 results computed with the CPU oracle primitives (test infrastructure).
Each function mirrors the reference routine named in its docstring on top of the alignment contract."""
import numpy as np

K1_MIN_VOTES = 16
K2_MIN_VOTES = 2
NONE = 0xFFFFFFFF


def score_value(length, nm, unmapped):
    """src/data_types/mapping.rs:191-195"""
    return max(float(nm + unmapped), 0.1) / float(length)


def k2_expected(oracle, fx, gene, cons_dna, cons_cdna, require_dna=False, disable_cdna=False):
    """score_read allele loop (src/hla/caller.rs:1411-1510). Returns (best allele index or -1, stats dict)."""
    idx = [a for a in range(len(fx.ids)) if fx.gene_of[a] == gene and (fx.dna[a] or not require_dna)]
    cdna_l, dna_l, dg_c, dg_d = [], [], [], []
    ccons = "" if disable_cdna else cons_cdna
    for a in idx:
        c = None if disable_cdna else fx.cdna[a]
        d = fx.dna[a] or None
        cdna_l.append(c)
        dna_l.append(d)
        for seq, cons, out in ((c, ccons, dg_c), (d, cons_dna, dg_d)):
            if seq and cons:
                dd, v = oracle.anchor(cons, seq)              # allele_pos - cons_pos
                out.append(-dd if v >= K2_MIN_VOTES else None)
            else:
                out.append(None)
    best, stats, alns = oracle.hla_score_read(ccons, cons_dna, cdna_l, dna_l, dg_c, dg_d, 255)
    return (idx[best] if best >= 0 else -1), {a: stats[i].reshape(6).tolist() for i, a in enumerate(idx)}


def k1_expected(oracle, fx, reads):
    """HlaRealigner::realign_record (src/hla/realigner.rs:98-350) for every read; returns list of dicts + cell matrix"""
    G = len(fx.genes)
    n_all = len(fx.ids)
    refs = [oracle.encode(s) for s in fx.gene_ref]
    fwd = [fx.dna_fwd(a) if fx.dna[a] else "" for a in range(n_all)]
    fwd_e = [oracle.encode(s) if s else None for s in fwd]
    off = []
    am = []
    for a in range(n_all):
        if not fwd[a]:
            off.append(None)
            am.append(None)
            continue
        g = int(fx.gene_of[a])
        d, v = oracle.anchor(refs[g], fwd_e[a])             # allele_pos - ref_pos
        off.append(d if v >= K1_MIN_VOTES else None)
        m = None
        if v >= K1_MIN_VOTES:
            al, _ = oracle.wfa(fwd_e[a], refs[g], -d, 255, events=False)
            if al.ok and score_value(al.a_len, al.nm, al.a_len - (al.a_end - al.a_start)) < 1.0:
                m = (al.a_start, al.b_start)
        am.append(m)
    results = []
    cells = np.full((len(reads), n_all), NONE, np.uint32)
    for r, read in enumerate(reads):
        re = oracle.encode(read)
        anch = [oracle.anchor(refs[g], re) for g in range(G)]       # read_pos - ref_pos
        vmax = max(v for _, v in anch)
        vmin = max(vmax // 10, K1_MIN_VOTES)
        alns = np.zeros(n_all, oracle_aln_dtype())
        for a in range(n_all):
            if not fwd[a] or off[a] is None:
                continue
            g = int(fx.gene_of[a])
            if anch[g][1] < vmin:
                continue
            cap = min(255, int(0.03 * len(fwd[a])) + 1)
            al, _ = oracle.wfa(fwd_e[a], re, anch[g][0] - off[a], cap, events=False)
            if al.ok:
                alns[a] = (1, al.nm, al.a_start, al.a_end, al.b_start, al.b_end, al.a_len, al.b_len)
                cells[r, a] = (al.nm << 16) | (al.a_end - al.a_start)
        best = oracle.pick_allele(alns, len(read))
        res = dict(status=1, best_allele=-1, gene=-1)
        if best >= 0:
            g = int(fx.gene_of[best])
            bm = alns[best]
            res.update(status=3, best_allele=best, gene=g, nm=int(bm["nm"]), target_len=int(bm["a_len"]),
                       unmapped=int(bm["a_len"] - (bm["a_end"] - bm["a_start"])),
                       aln=tuple(int(x) for x in bm.tolist()))
            db_s, db_e = int(bm["b_start"]), int(bm["b_end"])
            buf_s, buf_e = max(db_s - 1000, 0), min(db_e + 1000, len(read))
            seg = re[buf_s:buf_e]
            rm, _ = oracle.wfa(refs[g], seg, anch[g][0] - buf_s, 255, events=False)
            reflen = len(refs[g])
            if rm.ok and score_value(reflen, rm.nm, reflen - (rm.a_end - rm.a_start)) < 1.0:
                adj_s, adj_e = buf_s + rm.b_start, buf_s + rm.b_end
                if adj_s < db_s or am[best] is None:
                    d = rm.a_start
                    h = oracle.hpc_pos(fx.gene_ref[g], d)
                else:
                    added = max(am[best][1] - am[best][0], 0)
                    d = added + int(bm["a_start"])
                    h = oracle.hpc_pos(fx.gene_ref[g], added) + oracle.hpc_pos(fx.dna[best], int(bm["a_start"]))
                res.update(status=0, seg_start=min(db_s, adj_s), seg_end=max(db_e, adj_e), dna_offset=d, hpc_offset=h)
        results.append(res)
    return results, cells


def oracle_aln_dtype():
    import oracle_ffi
    return oracle_ffi.ALN_DTYPE
