"""Pins the CPU oracle to the reference's own known-answer tests for the HLA path (no GPU needed).
Every test names the reference test it ports (/root/reference paths)."""
import ctypes as C
import gzip
import json
import math
import os

import numpy as np
import pytest

import oracle_ffi

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_process_mm_cigar_vectors(oracle):
    """src/hla/processed_match.rs:270-302 test_process_mm_cigar"""
    cigar = [(2, 7), (1, 8), (2, 7), (1, 1), (2, 7), (1, 2), (2, 7)]
    assert oracle.process_mm_cigar(cigar, 0, 10, 0, 0) == [0, 0, 0, 1, 1, 1, 2, 2, 3, 3, 3]
    assert oracle.process_mm_cigar(cigar, 3, 18, 2, 3) == [0, 0, 1, 2, 2, 2, 3, 3, 3, 4, 4, 5, 5, 5, 6, 7, 8, 8, 8]


def test_process_mm_cigar_large_unmapped(oracle):
    """src/hla/processed_match.rs:304-328 test_large_unmapped"""
    assert oracle.process_mm_cigar([(2, 7)], 2, 4, 100, 0) == [0, 1, 2, 2, 2]
    assert oracle.process_mm_cigar([(2, 7)], 0, 4, 0, 100) == [0, 0, 0, 1, 2]


def test_process_mm_cigar_bad_op(oracle):
    """unexpected cigar op is an error (processed_match.rs:243)"""
    with pytest.raises(ValueError):
        oracle.process_mm_cigar([(2, 0)], 0, 4, 0, 0)


def test_mapping_stats_scores(oracle):
    """src/data_types/mapping.rs:216-222 test_mapping_stats ; src/hla/mapping.rs:185-191 test_mapping_stats"""
    L = oracle.L
    assert L.osp_custom_score(10, 1, 0, 1) == 0.1
    assert L.osp_custom_score(10, 1, 0, 1) == L.osp_score_value(10, 1, 0)
    assert (L.osp_custom_score(10, 1, 0, 1), L.osp_custom_score(20, 0, 1, 1)) == (0.1, 0.05)
    # nm = 0 is floored at 0.1 (mapping.rs:191-195); un-penalised form drops unmapped from both sides (:60-84)
    assert L.osp_score_value(100, 0, 0) == 0.1 / 100.0
    assert L.osp_custom_score(100, 3, 20, 0) == 3.0 / 80.0
    assert L.osp_custom_score(100, 3, 20, 1) == 23.0 / 100.0


def test_select_best_mapping(oracle):
    """src/util/mapping.rs:22-57: default is the 100 % mismatch (1,1,0); strict < keeps the first of equals"""
    M = oracle_ffi.Mapping
    maps = (M * 3)(M(100, 0, 100, 200, 10, 110, 5, 1), M(100, 0, 100, 200, 10, 110, 3, 1), M(100, 0, 100, 200, 20, 120, 3, 1))
    st = (C.c_uint64 * 3)()
    assert oracle.L.osp_select_best_mapping(maps, 3, 0, 1, -1, C.byref(st)) == 1
    assert list(st) == [100, 3, 0]
    assert oracle.L.osp_select_best_mapping(maps, 3, 1, 1, -1, C.byref(st)) == 1
    assert list(st) == [200, 3, 100]
    # nothing beats the default when nm + unmapped >= len
    bad = (M * 1)(M(10, 0, 2, 10, 0, 2, 2, 1))
    assert oracle.L.osp_select_best_mapping(bad, 1, 0, 1, -1, C.byref(st)) == -1
    assert list(st) == [1, 1, 0]
    assert oracle.L.osp_select_best_mapping(maps, 0, 0, 1, 50, C.byref(st)) == -1
    assert list(st) == [50, 50, 0]


def _level(present, rs=0, re=0, length=0, nm=0, um=0, pc=None):
    lv = oracle_ffi.HlaLevel()
    lv.present, lv.range_start, lv.range_end, lv.len, lv.nm, lv.unmapped = present, rs, re, length, nm, um
    if pc is not None:
        arr = np.array(pc, np.uint64)
        lv._keep = arr
        lv.pc = arr.ctypes.data_as(C.POINTER(C.c_uint64))
    return lv


def _better(oracle, lhs, rhs):
    a = (oracle_ffi.HlaLevel * 2)(*lhs)
    b = (oracle_ffi.HlaLevel * 2)(*rhs)
    return bool(oracle.L.osp_is_better_match(C.byref(a), C.byref(b)))


def test_is_better_match_rules(oracle):
    """src/hla/processed_match.rs:103-184"""
    worst = [_level(0), _level(0)]
    pc_clean = [0] * 11
    pc_one = [0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1]
    a = [_level(1, 0, 10, 10, 0, 0, pc_clean), _level(0)]
    b = [_level(1, 0, 10, 10, 1, 0, pc_one), _level(0)]
    assert _better(oracle, a, worst) and not _better(oracle, worst, a)          # any mapping beats the worst match
    assert not _better(oracle, worst, worst)                                    # (1.0,1.0) is not < (1.0,1.0)
    assert _better(oracle, a, b) and not _better(oracle, b, a)                  # fewer edits in the overlap
    # overlap restricted to a clean stretch: tie on the level, tie-break by the whole-mapping score
    b2 = [_level(1, 5, 10, 10, 1, 0, pc_one), _level(0)]
    a2 = [_level(1, 5, 10, 10, 0, 0, pc_clean), _level(0)]
    assert _better(oracle, a2, b2) and not _better(oracle, b2, a2)
    # a present DNA level beats an absent one when the cDNA ties
    c = [_level(1, 0, 10, 10, 0, 0, pc_clean), _level(1, 0, 10, 10, 5, 0, list(range(11)))]
    assert _better(oracle, c, a) and not _better(oracle, a, c)
    # no overlap on a level counts as equal there (processed_match.rs:127-131)
    d = [_level(1, 0, 4, 10, 3, 0, [0, 1, 2, 3, 3, 3, 3, 3, 3, 3, 3]), _level(0)]
    e = [_level(1, 6, 10, 10, 0, 4, pc_clean), _level(0)]
    assert _better(oracle, d, e) == ((3 + 0) / 10 < (0 + 4) / 10)


def test_score_min(oracle):
    """test_score_min (src/hla/mapping.rs:220-229): HlaMappingScore orders by (cDNA score, DNA score).  The levels do not overlap, so
    is_better_match falls through to the score comparison (processed_match.rs:127-131, 180-183).  The three constructor panics of the
    same file (:193-217: a level with some but not all of its fields) have no counterpart: a flattened level is present or absent."""
    pc = [0] * 11
    def score(c, d, lo):                      # (nm + unmapped) / len on each level; lo picks a disjoint range
        return [_level(1, lo, lo + 5, 10, int(round(c * 10)), 0, pc), _level(1, lo, lo + 5, 10, int(round(d * 10)), 0, pc)]
    s1, s2, s3 = score(1.0, 0.5, 0), score(0.9, 1.0, 5), score(1.0, 0.2, 5)
    assert _better(oracle, s2, s1) and not _better(oracle, s1, s2)              # s1.min(s2) == s2
    assert _better(oracle, s3, s1) and not _better(oracle, s1, s3)              # s1.min(s3) == s3
    s2b = score(0.9, 1.0, 0)
    assert _better(oracle, s2b, s3) and not _better(oracle, s3, s2b)            # s2.min(s3) == s2


def test_is_passing_dual(oracle):
    """src/hla/caller.rs:1837-1845 test_is_passing_dual (min_cdf 0.001, min fraction 0.10, expected maf 0.5)"""
    f = lambda c1, c2: bool(oracle.L.osp_is_passing_dual(c1, c2, 0.10, 0.5, 0.001, None, None))
    assert not f(3, 20) and not f(20, 3)
    assert f(10, 20) and f(20, 10)
    cdf = C.c_double()
    oracle.L.osp_is_passing_dual(3, 20, 0.10, 0.5, 0.001, None, C.byref(cdf))
    assert abs(cdf.value - 2.4414e-4) < 1e-7                                     # binom.cdf(3,23,.5)
    oracle.L.osp_is_passing_dual(10, 20, 0.10, 0.5, 0.001, None, C.byref(cdf))
    assert abs(cdf.value - 0.049368) < 1e-5


def _hemi(oracle, c1, c2, norm, delta):
    n = c1 + c2
    is_c1 = np.array([1] * c1 + [0] * c2, np.uint8)
    s1 = np.array([0] * c1 + [delta] * c2, np.int64)
    s2 = np.array([delta] * c1 + [0] * c2, np.int64)
    h, d = C.c_double(), C.c_double()
    r = oracle.L.osp_is_hemizygous_better(s1.ctypes.data_as(C.c_void_p), s2.ctypes.data_as(C.c_void_p), is_c1.ctypes.data_as(C.c_void_p),
                                          n, 1 if c2 else 0, 20, 1, norm, C.byref(h), C.byref(d))
    return bool(r), h.value, d.value


def test_is_hemizygous_better(oracle):
    """src/hla/caller.rs:1884-1898 test_is_hemizygous_better + the five cost pairs reproduced in SURVEY 8(c)"""
    cases = [((20, 0, 20.0, 1), True, 1.61, 51.6), ((40, 0, 20.0, 1), False, 51.6, 1.61), ((18, 2, 20.0, 1), True, 5.61, 68.8),
             ((18, 17, 20.0, 1), False, 63.7, 8.79), ((15, 6, 20.0, 20), False, 241.7, 54.0)]
    for args, want, hc, dc in cases:
        got, h, d = _hemi(oracle, *args)
        assert got == want, args
        assert abs(h - hc) < 0.06 and abs(d - dc) < 0.06, (args, h, d)


def test_realign_filter(oracle):
    """src/hla/realigner.rs:124-146: <= 0.5 unmapped+nm, <= 0.03 edits in the mapped part, strictly better wins, first of equals kept"""
    al = np.zeros(5, oracle_ffi.ALN_DTYPE)
    al[0] = (1, 200, 0, 3000, 0, 3000, 3000, 9000)      # 6.7 % edits -> rejected
    al[1] = (1, 3, 0, 1000, 0, 1000, 3000, 9000)        # only a third of the allele mapped -> rejected
    al[2] = (1, 6, 0, 3000, 0, 3000, 3000, 9000)        # 0.002
    al[3] = (1, 4, 0, 2000, 0, 2000, 3000, 9000)        # 0.002 exactly equal -> the earlier one stays
    al[4] = (0, 0, 0, 0, 0, 0, 3000, 9000)
    assert oracle.pick_allele(al, 9000) == 2
    al[3]["nm"] = 3
    assert oracle.pick_allele(al, 9000) == 3
    assert oracle.pick_allele(al[:2], 9000) == -1


def _faux():
    return json.load(open(os.path.join(GOLDEN, "hla_faux_database.json")))


def _cfg():
    return json.load(gzip.open(os.path.join(GOLDEN, "hla_db_v0.14.1.json.gz")))["hla_config"]


def test_reference_alleles(oracle):
    """src/hla/caller.rs:1710-1773 test_reference_alleles: the reference allele as a read (CIGAR all-M at the gene start)
    must type as itself with (cdna_len,0,0,dna_len,0,0); HLA-B goes through the reverse strand."""
    db, cfg = _faux(), _cfg()
    ids = sorted(db["hla_sequences"])
    cases = [("HLA-A", "HLA:HLA00037", "03:01:01:01", 29942254, False), ("HLA-B", "HLA:HLA00132", "07:02:01:01", 31353362, True)]
    for gene, key, star, pos1, is_rev in cases:
        seq = db["hla_sequences"][key]["dna_sequence"]
        read = oracle.revcomp(seq) if is_rev else seq                  # hg38-forward read
        exons = [(e["start"], e["end"]) for e in cfg["hla_exons"][gene]]
        segs, _off = oracle.splice_read(pos1 - 1, [(len(read), 0)], exons)
        spliced = "".join(read[s:e] for s, e in segs)
        cons_dna = oracle.revcomp(read) if is_rev else read            # score_read puts the read on the gene strand
        cons_cdna = oracle.revcomp(spliced) if is_rev else spliced
        alle = [k for k in ids if db["hla_sequences"][k]["gene_name"] == gene]
        cd = [db["hla_sequences"][k]["cdna_sequence"] for k in alle]
        dn = [db["hla_sequences"][k]["dna_sequence"] for k in alle]
        dg_c = [(-oracle.anchor(cons_cdna, c)[0]) for c in cd]
        dg_d = [(-oracle.anchor(cons_dna, d)[0]) for d in dn]
        best, stats, _ = oracle.hla_score_read(cons_cdna, cons_dna, cd, dn, dg_c, dg_d)
        assert alle[best] == key
        assert ":".join(db["hla_sequences"][alle[best]]["star_allele"]) == star
        cdna_len = len(db["hla_sequences"][key]["cdna_sequence"])
        assert stats[best].reshape(6).tolist() == [cdna_len, 0, 0, len(seq), 0, 0]


def test_score_bad_read(oracle):
    """src/hla/caller.rs:1784-1809 test_score_bad_read: a 4-bp read maps nowhere -> no best id, every score is the worst"""
    db = _faux()
    alle = [k for k in sorted(db["hla_sequences"]) if db["hla_sequences"][k]["gene_name"] == "HLA-A"]
    dn = [db["hla_sequences"][k]["dna_sequence"] for k in alle]
    dg = []
    for d in dn:
        dd, v = oracle.anchor("ACGT", d)
        dg.append(-dd if v >= 2 else None)
    best, stats, _ = oracle.hla_score_read("", "ACGT", [None] * len(alle), dn, [None] * len(alle), dg)
    assert best == -1
    assert (stats == -1).all()


def test_realigned_record_helpers(oracle):
    """src/hla/realigner.rs:534-556 test_realigned_record: segment 4..10 of AACCGGTTAACCGGTTAACCGGTT -> GGTTAA / GTA"""
    full = "AACCGGTTAACCGGTTAACCGGTT"
    assert full[4:10] == "GGTTAA" and oracle.hpc(full[4:10]) == "GTA"


def test_alignment_contract_pinned_cases(oracle):
    """What the reference's tests pin about minimap2 itself: identical sequence => nm 0 over the whole
    query; a single mismatch ranks strictly worse; 'N' mismatches everything; too-short input => no mapping."""
    rng = np.random.default_rng(0)
    s = "".join(rng.choice(list("ACGT"), 900))
    d, v = oracle.anchor(s, s)
    al, ev = oracle.wfa(s, s, -d)
    assert (al.ok, al.nm, al.a_start, al.a_end, al.b_start, al.b_end) == (1, 0, 0, 900, 0, 900) and len(ev) == 0
    t = s[:400] + ("A" if s[400] != "A" else "C") + s[401:]
    al1, ev1 = oracle.wfa(t, s, 0)
    assert al1.nm == 1 and [(int(e) >> 30, int(e) & 0x3FFFFFFF) for e in ev1] == [(0, 400)]
    n = s[:400] + "N" + s[401:]
    for other in (s, t, n):
        assert oracle.wfa(n, other, 0)[0].nm == 1
    assert oracle.anchor("ACGT", s) == (0, 0)
