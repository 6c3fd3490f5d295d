"""Expected K1 / K2 results computed with the CPU oracle primitives (test infrastructure).
Each function mirrors the reference routine named in its docstring on top of the alignment contract."""
import numpy as np

K1_MIN_VOTES = 16
K1_WEAK_VOTES = 512
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


class K1Tables:
    """what HlaRealigner::new prepares (src/hla/realigner.rs:42-91): hg38-strand alleles, their frame offsets on the gene reference
    and, lazily, their static mappings to it"""

    def __init__(self, oracle, fx, off=None):
        self.oracle, self.fx = oracle, fx
        n_all = len(fx.ids)
        self.refs = [oracle.encode(s) for s in fx.gene_ref]
        self.fwd = [fx.dna_fwd(a) if fx.dna[a] else "" for a in range(n_all)]
        self.fwd_e = [oracle.encode(s) if s else None for s in self.fwd]
        self.off, self._am = [], {}
        if off is not None:                                  # frame offsets worked out before (INT_MIN = none)
            self.off = [None if int(x) == -2 ** 31 else int(x) for x in off]
            return
        for a in range(n_all):
            if not self.fwd[a]:
                self.off.append(None)
                continue
            d, v = oracle.anchor(self.refs[int(fx.gene_of[a])], self.fwd_e[a])             # allele_pos - ref_pos
            self.off.append(d if v >= K1_MIN_VOTES else None)

    def am(self, a):
        """static allele -> gene reference mapping (a_start, b_start) or None"""
        if a not in self._am:
            m = None
            if self.off[a] is not None:
                g = int(self.fx.gene_of[a])
                al, _ = self.oracle.wfa(self.fwd_e[a], self.refs[g], -self.off[a], 511, events=False, retry=2)
                if al.ok and score_value(al.a_len, al.nm, al.a_len - (al.a_end - al.a_start)) < 1.0:
                    m = (al.a_start, al.b_start)
            self._am[a] = m
        return self._am[a]

    def anchors(self, re):
        return [self.oracle.anchor(self.refs[g], re) for g in range(len(self.fx.genes))]       # read_pos - ref_pos

    def is_reverse(self, re, anch):
        """src/hla/realigner.rs:178-193 at the level of the seeds (include/starphase_hip.h, sp_hla_realign.status 2): a read with a weak forward
        anchor (< 512 votes) whose best anchor on the reverse-complemented gene references has more votes (and >= 16) is dropped: 2 when the
        reverse anchor leads by a factor of two or more, 1 when it leads by less (then only a read that did not realign forwards is dropped)"""
        fwd = max((v for _d, v in anch), default=0)
        if fwd >= K1_WEAK_VOTES:
            return 0
        if not hasattr(self, "refs_rev"):
            comp = np.array([3, 2, 1, 0, 4], np.uint8)
            self.refs_rev = [np.ascontiguousarray(comp[r][::-1]) for r in self.refs]
        rev = max((self.oracle.anchor(rr, re)[1] for rr in self.refs_rev), default=0)
        return (2 if rev >= 2 * fwd else 1) if (rev >= K1_MIN_VOTES and rev > fwd) else 0

    def cell(self, a, re, anch):
        cap = min(511, int(0.03 * len(self.fwd[a])) + 1)
        al, _ = self.oracle.wfa(self.fwd_e[a], re, anch[int(self.fx.gene_of[a])][0] - self.off[a], cap, events=False)
        return al

    def record(self, read, re, anch, best, bm, seeded=None):
        """everything realign_record does once the best allele is known (src/hla/realigner.rs:149-350); bm = its alignment as an
        ALN_DTYPE row.  seeded: None = the exhaustive mode (strand decided at the seeds, mm2 = the re-score of bm), else dict(rev, mm2, chains, mappings,
        chain_score) of the seeded map (omm_hla_k1_seeded): a best mapping on the reverse strand drops the read, mm2 = the accepted mapping's re-score"""
        oracle, fx = self.oracle, self.fx
        res = dict(status=1, best_allele=-1, gene=-1)
        if seeded is not None:
            res.update(k1_chains=seeded["chains"], k1_mappings=seeded["mappings"], k1_chain_score=seeded["chain_score"])
        if best < 0:
            if (seeded["rev"] if seeded is not None else self.is_reverse(re, anch)):
                res["status"] = 2
            return res
        g = int(fx.gene_of[best])
        res.update(status=3, best_allele=best, gene=g, nm=int(bm["nm"]), target_len=int(bm["a_len"]),
                   unmapped=int(bm["a_len"] - (bm["a_end"] - bm["a_start"])), aln=tuple(int(x) for x in bm.tolist()))
        # the same mapping re-scored the reference's way (sp_hla_realign.mm2_*: oracle/affine.c on the 64 diagonals around the alignment)
        import oracle_ffi
        twice = (int(bm["a_start"]) - int(bm["b_start"])) + (int(bm["a_end"]) - int(bm["b_end"]))
        diag = int(twice / 2)                                                  # (C's division: towards zero)
        res["mm2"] = seeded["mm2"] if seeded is not None else oracle_ffi.oracle_affine(oracle, self.fwd_e[best], re, -diag, 64, 1)
        # what realign_record takes from the accepted mapping (bm.query_start / query_end / target_start, realigner.rs:219-221,307): in seeded mode the numbers of the
        # re-scored mapping (minimap2's end-clipped extent), in the exhaustive mode the cell's
        db_s, db_e, t_s = int(bm["b_start"]), int(bm["b_end"]), int(bm["a_start"])
        if seeded is not None and res["mm2"][0] > 0:
            _sc, _nm, t_s, _te, db_s, db_e = (int(x) for x in res["mm2"])
        buf_s, buf_e = max(db_s - 1000, 0), min(db_e + 1000, len(read))
        seg = re[buf_s:buf_e]
        # (seeded mode: a segment the 64-diagonal cell loses -- a long insertion / deletion against the reference -- runs again on the wide band, as the chains' cells do)
        rm, _ = oracle.wfa(self.refs[g], seg, anch[g][0] - buf_s, 511, events=False, retry=1 if seeded is not None else 0)
        reflen = len(self.refs[g])
        rm_nm, rm_a_start, rm_a_end, rm_b_start, rm_b_end = rm.nm, rm.a_start, rm.a_end, rm.b_start, rm.b_end
        if seeded is not None and rm.ok:
            # round 6: the second stage in the reference's numbers -- the segment's placement on the reference re-scored as minimap2 reports a mapping (two-piece affine gaps,
            # end clipping; segment = query, reference = target) on the 256 diagonals around it: its query span and target start are what realign_record takes (realigner.rs:262-283)
            twice2 = (rm.a_start - rm.b_start) + (rm.a_end - rm.b_end)
            af = oracle_ffi.oracle_affine(oracle, self.refs[g], seg, -int(twice2 / 2), 256, 1)      # (score, nm, t_start, t_end, q_start, q_end)
            if af[0] > 0:
                _sc2, rm_nm, rm_a_start, rm_a_end, rm_b_start, rm_b_end = (int(x) for x in af)
        if rm.ok and score_value(reflen, rm_nm, reflen - (rm_a_end - rm_a_start)) < 1.0:
            adj_s, adj_e = buf_s + rm_b_start, buf_s + rm_b_end
            am = self.am(best)
            if adj_s < db_s or am is None:
                d = rm_a_start
                h = oracle.hpc_pos(fx.gene_ref[g], d)
            else:
                added = max(am[1] - am[0], 0)
                d = added + t_s
                h = oracle.hpc_pos(fx.gene_ref[g], added) + oracle.hpc_pos(fx.dna[best], t_s)
            res.update(status=0, seg_start=min(db_s, adj_s), seg_end=max(db_e, adj_e), dna_offset=d, hpc_offset=h)
        if seeded is None:
            rv = self.is_reverse(re, anch)
            if rv == 2 or (rv == 1 and res["status"] != 0):
                res["status"] = 2                                  # (the other fields keep what the forward search found)
        return res


def k1_records_for(oracle, fx, reads, best_alleles, tables=None):
    """realign_record's result for reads whose best allele is already known (bench.py's CPU leg: the cells ran in C)"""
    tb = tables or K1Tables(oracle, fx)
    out = []
    for read, best in zip(reads, best_alleles):
        re = oracle.encode(read)
        anch = tb.anchors(re)
        bm = None
        if best >= 0:
            al = tb.cell(best, re, anch)
            bm = np.zeros(1, oracle_aln_dtype())[0]
            bm["ok"], bm["nm"], bm["a_start"], bm["a_end"], bm["b_start"], bm["b_end"], bm["a_len"], bm["b_len"] = (
                1, al.nm, al.a_start, al.a_end, al.b_start, al.b_end, al.a_len, al.b_len)
        out.append(tb.record(read, re, anch, best, bm))
    return out


def k1_expected(oracle, fx, reads):
    """HlaRealigner::realign_record (src/hla/realigner.rs:98-350) for every read; returns list of dicts + cell matrix"""
    n_all = len(fx.ids)
    tb = K1Tables(oracle, fx)
    results = []
    cells = np.full((len(reads), n_all), NONE, np.uint32)
    for r, read in enumerate(reads):
        re = oracle.encode(read)
        anch = tb.anchors(re)
        vmax = max(v for _, v in anch)
        vmin = max(vmax // 10, K1_MIN_VOTES)
        alns = np.zeros(n_all, oracle_aln_dtype())
        for a in range(n_all):
            if not tb.fwd[a] or tb.off[a] is None:
                continue
            if anch[int(fx.gene_of[a])][1] < vmin:
                continue
            al = tb.cell(a, re, anch)
            if al.ok:
                alns[a] = (1, al.nm, al.a_start, al.a_end, al.b_start, al.b_end, al.a_len, al.b_len)
                cells[r, a] = (al.nm << 16) | (al.a_end - al.a_start)
        best = oracle.pick_allele(alns, len(read))
        results.append(tb.record(read, re, anch, best, alns[best] if best >= 0 else None))
    return results, cells


_SEED_INDEX = {}


def seed_index(oracle, fx):
    """the minimizer index of the fixture's DNA alleles in hg38 orientation (aligner.with_index of HlaRealigner::new), cached per fixture"""
    import mm2_ffi
    key = id(fx)
    if key not in _SEED_INDEX:
        mm = mm2_ffi.Mm2(oracle)
        dna_ids = [a for a in range(len(fx.ids)) if fx.dna[a]]
        _SEED_INDEX[key] = (mm2_ffi.Index(mm, [fx.dna_fwd(a) for a in dna_ids]), dna_ids, fx)
    return _SEED_INDEX[key][0], _SEED_INDEX[key][1]


def k1_expected_seeded(oracle, fx, reads, tables=None):
    """HlaRealigner::realign_record in the reference's call pattern (sp_hla_realign_reads with k1_best_n > 0): the seeded map of oracle/mm2.c
    (omm_hla_k1_seeded: minimap2's seeding, chaining and selection, the library's cell + re-score for the selected chains, the acceptance loop over the
    mappings in output order), then the bookkeeping behind the accepted mapping.  Returns the records and the per-read (pick, hits)."""
    idx, dna_ids = seed_index(oracle, fx)
    tb = tables or K1Tables(oracle, fx)
    out, audits = [], []
    for read in reads:
        re = oracle.encode(read)
        pick, hits, n_chains = idx.k1_seeded(read) if len(read) else (-1, [], 0)
        best, bm, info = -1, None, dict(rev=False, mm2=(0, 0, 0, 0, 0, 0), chains=n_chains, mappings=len(hits), chain_score=0)
        if pick >= 0:
            h = hits[pick]
            info["chain_score"] = int(h["chain_score"])
            info["rev"] = bool(h["rev"])
            if not h["rev"]:
                best = dna_ids[int(h["rid"])]
                bm = np.zeros(1, oracle_aln_dtype())[0]
                bm["ok"], bm["nm"], bm["a_start"], bm["a_end"], bm["b_start"], bm["b_end"], bm["a_len"], bm["b_len"] = (
                    1, h["cell_nm"], h["a_start"], h["a_end"], h["b_start"], h["b_end"], h["t_len"], len(read))
                info["mm2"] = (int(h["dp_max"]), int(h["nm"]), int(h["t_start"]), int(h["t_end"]), int(h["q_start"]), int(h["q_end"]))
        anch = tb.anchors(re) if len(read) else [(0, 0)] * len(fx.genes)
        out.append(tb.record(read, re, anch, best, bm, seeded=info))
        audits.append((pick, hits))
    return out, audits


def oracle_aln_dtype():
    import oracle_ffi
    return oracle_ffi.ALN_DTYPE
