"""The host-side scalar routines of the library (no GPU needed: they do no device work) against the reference's own tests."""
import ctypes as C

import numpy as np

import oracle_ffi as of


def test_is_passing_dual(pkg):
    """src/hla/caller.rs:1837-1845"""
    L = pkg.ffi.lib()
    f = lambda a, b: L.sp_hla_is_passing_dual(a, b, 0.10, 0.5, 0.001, None, None)
    assert (f(3, 20), f(20, 3), f(10, 20), f(20, 10)) == (0, 0, 1, 1)


def test_is_hemizygous_better(pkg, oracle):
    """src/hla/caller.rs:1884-1898 ; identical costs to the oracle restatement"""
    L = pkg.ffi.lib()
    for (c1, c2, norm, delta), want in (((20, 0, 20.0, 1), 1), ((40, 0, 20.0, 1), 0), ((18, 2, 20.0, 1), 1), ((18, 17, 20.0, 1), 0), ((15, 6, 20.0, 20), 0)):
        n = c1 + c2
        is_c1 = np.array([1] * c1 + [0] * c2, np.uint8)
        s1 = np.array([0] * c1 + [delta] * c2, np.int64)
        s2 = np.array([delta] * c1 + [0] * c2, np.int64)
        h, d, oh, od = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        got = L.sp_hla_is_hemizygous_better(s1.ctypes.data, s2.ctypes.data, is_c1.ctypes.data, n, 1 if c2 else 0, 20, norm, C.byref(h), C.byref(d))
        exp = oracle.L.osp_is_hemizygous_better(s1.ctypes.data_as(C.c_void_p), s2.ctypes.data_as(C.c_void_p), is_c1.ctypes.data_as(C.c_void_p),
                                                n, 1 if c2 else 0, 20, 1, norm, C.byref(oh), C.byref(od))
        assert got == want == exp
        assert abs(h.value - oh.value) <= 1e-9 and abs(d.value - od.value) <= 1e-9


def test_hpc(pkg):
    """src/util/homopolymers.rs:72-100"""
    L = pkg.ffi.lib()
    out = C.create_string_buffer(64)
    n = L.sp_hpc(b"AACAAAAAAGGGTAACAA", 18, out)
    assert out.raw[:n] == b"ACAGTACA"
    seq = b"AACCCGTTTT"
    assert [L.sp_hpc_pos(seq, len(seq), i) for i in range(len(seq))] == [0, 0, 1, 1, 1, 2, 3, 3, 3, 3]
    assert L.sp_hpc_pos(b"ATTGGGGGAACCCGTTTT", 18, 6) == 2


def test_chain_to_hap(pkg):
    """src/cyp2d6/caller.rs:972-1006"""
    L = pkg.ffi.lib()
    cfg = of.default_cyp_config()
    tk = (C.c_char_p * len(cfg["translate"]))(*[a.encode() for a, _ in cfg["translate"]])
    tv = (C.c_char_p * len(cfg["translate"]))(*[b.encode() for _, b in cfg["translate"]])
    types = np.array([6, 2, 2, 2, 2], np.int32)
    subs = (C.c_char_p * 5)(None, b"1.001", b"10", b"1.002", b"1.002")

    def hap(chain, detail):
        ch = np.array(chain, np.int32)
        out = C.create_string_buffer(256)
        L.sp_cyp_chain_to_hap(ch.ctypes.data, len(ch), types.ctypes.data, subs, len(cfg["translate"]), tk, tv, detail, out, 256)
        return out.value.decode()
    assert hap([2, 2, 1, 0], 1) == "*1.001 + *10x2"
    assert hap([3, 1, 0], 1) == "*1.001 + *1.002"
    assert hap([3, 1, 0], 0) == "*1x2"
    assert hap([3, 4], 1) == "*1.002x2"


def _py_build_chains(hap_type, read_seg_off, ed, kept):
    """independent pure-Python statement of src/cyp2d6/caller.rs:429-583 (third opinion for the two C versions)"""
    n_haps = len(hap_type)
    uniq = [0] * n_haps
    recorded = []
    for r in range(len(read_seg_off) - 1):
        segs = range(read_seg_off[r], read_seg_off[r + 1])
        if len(segs) == 0:
            continue
        chains, rows = [[]], []
        for sg in segs:
            if not kept[sg]:
                continue
            w = [int(x) for x in ed[sg]]
            mn = min(w)
            nmin = w.count(mn)
            nxt = []
            for pc in chains:
                for c in range(n_haps):
                    if w[c] == mn:
                        nxt.append(pc + [c])
                        if nmin == 1:
                            uniq[c] += 1
            chains = nxt
            rows.append(sg)
        if chains == [[]]:
            continue
        recorded.append((r, chains, rows))
    out = dict(read_index=[], chains=[], w_rows=[])
    for r, chains, rows in recorded:
        f = [c for c in chains if all(uniq[x] > 0 for x in c)]
        if not f:
            return None
        out["read_index"].append(r); out["chains"].append(f); out["w_rows"].append(rows)
    out["unique_counts"] = uniq
    out["false_allele"] = [int(uniq[h] == 0 and hap_type[h] not in (0, 9)) for h in range(n_haps)]
    return out


def _same_chains(a, b):
    assert a["read_index"] == b["read_index"] and a["chains"] == b["chains"] and a["w_rows"] == b["w_rows"]
    assert [int(x) for x in a["unique_counts"]] == [int(x) for x in b["unique_counts"]]
    assert [int(x) for x in a["false_allele"]] == [int(x) for x in b["false_allele"]]


def test_build_chains_handmade(pkg, oracle):
    """chain building, src/cyp2d6/caller.rs:429-583: ambiguity multiplies chains, a unique minimum is counted once per chain
    being extended, unsupported consensuses drop out of ambiguous chains and become FalseAllele, dropped segments/reads"""
    hap_type = [1, 2, 2, 3, 0]                     # REP6, D6, D6, link, Unknown
    ed = np.array([[0, 9, 9, 9, 9],                # read 0: REP6 unique
                   [9, 1, 1, 9, 9],                #          D6 ambiguous (haps 1 and 2)
                   [9, 9, 9, 0, 9],                #          link unique -> counted twice (two chains being extended)
                   [9, 0, 5, 9, 9],                # read 1: hap 1 unique
                   [7, 7, 7, 7, 7],                # read 2: dropped segment (kept = 0) only -> read not recorded
                   [9, 2, 9, 9, 9]], np.uint64)    # read 4: hap 1 unique
    kept = np.array([1, 1, 1, 1, 0, 1], np.uint8)
    seg_off = [0, 3, 4, 5, 5, 6]                   # read 3 has no segments at all
    got = pkg.ffi.build_chains(hap_type, seg_off, ed, kept)
    assert got["read_index"] == [0, 1, 4]
    assert got["chains"] == [[[0, 1, 3]], [[1]], [[1]]]          # [0,2,3] removed: hap 2 has no unique support
    assert got["w_rows"] == [[0, 1, 2], [3], [5]]
    assert [int(x) for x in got["unique_counts"]] == [1, 2, 0, 2, 0]
    assert [int(x) for x in got["false_allele"]] == [0, 0, 1, 0, 0]  # Unknown (hap 4) is never re-labelled
    _same_chains(got, of.oracle_build_chains(oracle, hap_type, seg_off, ed, kept))
    _same_chains(got, _py_build_chains(hap_type, seg_off, ed, kept))
    # a read whose only candidate has no unique support anywhere: the reference panics ("chain collapse")
    ed2 = np.array([[3, 3, 9, 9, 9]], np.uint64)
    try:
        pkg.ffi.build_chains(hap_type, [0, 1], ed2, np.ones(1, np.uint8))
        raise AssertionError("expected SP_ERR_CHAIN_COLLAPSE")
    except pkg.StarphaseError as e:
        assert e.code == 7
    assert of.oracle_build_chains(oracle, hap_type, [0, 1], ed2, np.ones(1, np.uint8)) is None


def test_build_chains_random(pkg, oracle):
    rng = np.random.default_rng(77)
    n_ok = 0
    for it in range(200):
        n_haps = int(rng.integers(1, 9))
        n_reads = int(rng.integers(0, 12))
        nseg = rng.integers(0, 5, n_reads)
        seg_off = np.concatenate([[0], np.cumsum(nseg)]).astype(np.uint32)
        S = int(seg_off[-1])
        ed = rng.integers(0, 4, (S, n_haps)).astype(np.uint64)
        kept = (rng.random(S) < 0.85).astype(np.uint8)
        hap_type = rng.integers(0, 10, n_haps).astype(np.int32)
        exp = _py_build_chains([int(x) for x in hap_type], [int(x) for x in seg_off], ed, kept)
        orc = of.oracle_build_chains(oracle, hap_type, seg_off, ed.reshape(-1), kept)
        if exp is None:
            assert orc is None
            try:
                pkg.ffi.build_chains(hap_type, seg_off, ed.reshape(-1), kept)
                raise AssertionError("expected chain collapse")
            except pkg.StarphaseError as e:
                assert e.code == 7
            continue
        n_ok += 1
        _same_chains(orc, exp)
        _same_chains(pkg.ffi.build_chains(hap_type, seg_off, ed.reshape(-1), kept), exp)
    assert n_ok > 50


# ------------------------------------------------------------------ variant normalisation (a24) in the product library
import pytest
import variant_glue as vg

GENOME = {"chr1": "AAAAAAAAAAACACACACAC", "chr2": "ACACACACACAGTAGTAGTA", "chr3": "ACGTACGTACGTACGTACGT"}


def test_product_normalize_vectors(pkg, oracle):
    """the same vectors tests/test_oracle_variant.py pins the oracle on (normalized_variant.rs:43-170,262-279), through sp_variant_normalize"""
    prod = vg.ProductNormalizer(pkg)
    n = lambda *a: vg.normalize(prod, *a)
    assert n("chr1", 10, "A", "C", None) == ("chr1", 10, "A", "C")
    assert n("chr1", 10, "AC", "AG", None) == ("chr1", 11, "C", "G")
    assert n("chr1", 10, "CA", "GA", None) == ("chr1", 10, "C", "G")
    assert n("chr1", 14, "AC", "del", GENOME) == ("chr1", 9, "AAC", "A")
    assert n("chr1", 14, "del", "insAC", GENOME) == ("chr1", 14, "A", "AAC")
    assert n("chr1", 12, "AC(2)", "AC(3)", GENOME) == ("chr1", 9, "A", "AAC")
    assert n("chr2", 10, "AGT", "delinsA", None) == ("chr2", 10, "AGT", "A")
    for bad in (("chr1", 10, "", "A", None), ("chr1", 10, "del", "A", None), ("chr1", 10, "C", "A", GENOME), ("chr1", 10, "A", "N", None),
                ("chr9", 10, "A", "C", GENOME), ("chr1", 0, "A", "del", GENOME), ("chr1", 10, "del", "insAC", None)):
        with pytest.raises(ValueError):
            n(*bad)
        with pytest.raises(ValueError):
            vg.normalize(oracle, *bad)
    assert vg.multi_new(prod, "chr1", 10, "A", "R", None) == [None, ("chr1", 10, "A", "G")]
    assert vg.multi_new(prod, "chr1", 10, "A", "Y", None) == [("chr1", 10, "A", "C"), ("chr1", 10, "A", "T")]
    assert vg.multi_new(prod, "chr1", 10, "A", "delinsCC; delinsCCC", None) == [("chr1", 10, "A", "CC"), ("chr1", 10, "A", "CCC")]


def test_product_normalize_fuzz(pkg, oracle):
    """random CPIC-syntax alleles over a repetitive contig: library == oracle, including which inputs are refused"""
    rng = np.random.default_rng(5)
    contig = "".join(rng.choice(list("AC"), 30)) + "ACACACACAC" + "GGGGGG" + "".join(rng.choice(list("ACGT"), 40)) + "TTTTTTAGAGAGAG"
    prod = vg.ProductNormalizer(pkg)
    n_ok = n_bad = 0
    for it in range(3000):
        pos = int(rng.integers(0, len(contig) - 6))
        rl = int(rng.integers(0, 5))
        ref_bases = contig[pos:pos + rl]
        kind = int(rng.integers(0, 7))
        rnd = lambda k: "".join(rng.choice(list("ACGT"), k))
        if kind == 0:
            ref, alt = (ref_bases or "del"), rnd(int(rng.integers(1, 4)))
        elif kind == 1:
            ref, alt = (ref_bases or "del"), "del"
        elif kind == 2:
            ref, alt = "del", "ins" + rnd(int(rng.integers(1, 4)))
        elif kind == 3:
            ref, alt = (ref_bases or "del"), "delins" + rnd(int(rng.integers(0, 4)))
        elif kind == 4:                                          # tandem-repeat syntax, genome-consistent or not
            unit = contig[pos:pos + int(rng.integers(1, 3))]
            ref, alt = f"{unit}({int(rng.integers(1, 4))})", f"{unit}({int(rng.integers(0, 5))})"
        elif kind == 5:                                          # duplicate a downstream copy: exercises the left shift
            ref, alt = (ref_bases or "del"), "ins" + contig[pos:pos + int(rng.integers(1, 4))] if not ref_bases else ref_bases + ref_bases
        else:
            ref, alt = rnd(max(1, rl)), rnd(int(rng.integers(1, 4)))   # usually disagrees with the genome
        for genome in ({"c": contig}, None):
            try:
                want = vg.normalize(oracle, "c", pos, ref, alt, genome)
            except ValueError:
                want = None
            try:
                got = vg.normalize(prod, "c", pos, ref, alt, genome)
            except ValueError:
                got = None
            assert got == want, (pos, ref, alt, genome is not None, got, want)
            n_ok += want is not None
            n_bad += want is None
    assert n_ok > 1500 and n_bad > 300


def test_product_loads_reference_databases(pkg, oracle):
    """load_database_haplotypes (src/diplotyper.rs:437-538) over every committed DB fixture: identical tables whether the strings are
    normalised by the oracle or by the library"""
    import json, os, glob
    ref = json.load(open(os.path.join(vg.GOLDEN, "test_reference.json"))) if os.path.exists(os.path.join(vg.GOLDEN, "test_reference.json")) else None
    prod = vg.ProductNormalizer(pkg)
    n = 0
    for path in sorted(glob.glob(os.path.join(vg.GOLDEN, "variant_dbs", "*.json"))):
        db = json.load(open(path))
        for gene, entry in db["gene_entries"].items():
            for genome in (None, ref):
                if genome is not None and entry["chromosome"] not in genome:
                    continue
                a = vg.load_database_haplotypes(oracle, entry, genome)
                b = vg.load_database_haplotypes(prod, entry, genome)
                assert a == b
                n += 1
    assert n >= 5


# ------------------------------------------------------------------ result strings (src/data_types/pgx_diplotype.rs:237-316, region_variants.rs:66-110)
REL = dict(Unknown=0, Match=1, Unexpected=2, Missing=3, AmbiguousUnexpected=4, AmbiguousMissing=5, UnknownUnexpected=6, UnknownMissing=7)


def _dip(pkg, oracle, h1, h2, pharmcat):
    out, oout = C.create_string_buffer(256), C.create_string_buffer(256)
    pkg.ffi.lib().sp_diplotype_string(h1.encode(), h2.encode(), int(pharmcat), out, 256)
    oracle.L.osp_diplotype_string(h1.encode(), h2.encode(), int(pharmcat), oout, C.c_size_t(256))
    assert out.value == oout.value
    return out.value.decode()


def _inexact(pkg, oracle, base, variants):
    """variants: [(label, is_vi, state name)] in any order for the library; the oracle gets them in BTreeSet order"""
    def call_lib(vs):
        labels = (C.c_char_p * max(1, len(vs)))(*[v[0].encode() for v in vs])
        vi = np.array([v[1] for v in vs], np.uint8)
        st = np.array([REL[v[2]] for v in vs], np.int32)
        mt, out = C.c_int32(0), C.create_string_buffer(512)
        pkg.ffi.lib().sp_inexact_haplotype(base.encode(), len(vs), labels, vi.ctypes.data, st.ctypes.data, C.byref(mt), out, 512)
        return mt.value, out.value.decode()
    ordered = sorted(set(variants), key=lambda v: (v[0], v[1], REL[v[2]]))
    labels = (C.c_char_p * max(1, len(ordered)))(*[v[0].encode() for v in ordered])
    vi = np.array([v[1] for v in ordered], np.uint8)
    st = np.array([REL[v[2]] for v in ordered], np.int32)
    oout = C.create_string_buffer(512)
    oracle.L.osp_inexact_haplotype.restype = C.c_int
    omt = oracle.L.osp_inexact_haplotype(base.encode(), len(ordered), labels, vi.ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p), oout, C.c_size_t(512))
    got = call_lib(list(reversed(variants)) + variants[:1])            # shuffled and with a duplicate: the library sorts and dedups
    assert got == (omt, oout.value.decode())
    return got


def test_diplotype_strings(pkg, oracle):
    """test_diplotype / test_pharmcat_diplotype (src/data_types/pgx_diplotype.rs:237-256)"""
    assert _dip(pkg, oracle, "B", "A", False) == "B/A"
    assert _dip(pkg, oracle, "*4", "*1", True) == "*4/*1"
    assert _dip(pkg, oracle, "*4x2", "*1", True) == "*4x2/*1"
    assert _dip(pkg, oracle, "*4 + *68", "*1", True) == "[*4 + *68]/*1"
    assert _dip(pkg, oracle, "*4 + *68", "*1", False) == "*4 + *68/*1"


def test_inexact_haplotype_strings(pkg, oracle):
    """test_inexact_haplotype (src/data_types/pgx_diplotype.rs:276-316) and RegionVariant's Display (region_variants.rs:66-110)"""
    SUB, CORE, NO = 3, 2, 1
    assert _inexact(pkg, oracle, "*1.001", [("rs123", 1, "Match"), ("rs456", 0, "Match")]) == (SUB, "*1.001")
    assert _inexact(pkg, oracle, "*1.001", [("rs123", 1, "Match"), ("rs456", 0, "Unexpected")]) == (CORE, "(*1.001 +rs456)")
    assert _inexact(pkg, oracle, "*1", [("rs123", 1, "Missing"), ("rs456", 0, "Unexpected")]) == (NO, "(*1 -rs123 +rs456)")
    assert _inexact(pkg, oracle, "*3", []) == (SUB, "*3")
    for state in ("AmbiguousUnexpected", "AmbiguousMissing", "UnknownUnexpected", "UnknownMissing", "Unknown"):
        assert _inexact(pkg, oracle, "*9", [("rs012", 0, state)]) == (CORE, "(*9 ?rs012)")
    # InexactDiplotype::new joins the two full haplotypes (:99-106)
    h1 = _inexact(pkg, oracle, "*1", [("v1", 1, "Match"), ("v2", 1, "Unexpected")])[1]
    h2 = _inexact(pkg, oracle, "*2", [("v2", 1, "Match")])[1]
    assert _dip(pkg, oracle, h1, h2, False) == "(*1 +v2)/*2"


# ------------------------------------------------------------------ is_deletion (src/diplotyper.rs:1020-1174)
def test_is_deletion_vectors(pkg, oracle):
    """test_deletion_search (src/diplotyper.rs:2169-2251) through sp_variant_is_deletion; reverse strand; undefined gene"""
    import pytest
    import variant_glue as vg
    from test_oracle_variant import DELETION_SEARCH, deletion_search_defs
    genes, svs, defs = deletion_search_defs(pkg)
    for start, end, want in DELETION_SEARCH:
        assert defs.is_deletion(start, end) == want, (start, end)
    genes["gene2"]["is_forward_strand"] = False
    rev = pkg.ffi.SvDefinitions(genes, svs)
    assert rev.is_deletion(100, 145) == "specific_partial" and rev.is_deletion(125, 200) == "generic_partial"
    svs["partial_gene_deletions"]["ghost"] = {"is_generic": True, "exons_deleted": {"gene3": {"start": 0, "end": 1}}}
    with pytest.raises(pkg.StarphaseError):
        pkg.ffi.SvDefinitions(genes, svs).is_deletion(0, 1)
    assert pkg.ffi.SvDefinitions(genes, {}).is_deletion(0, 1000) is None          # a gene entry without definitions


def test_is_deletion_random(pkg, oracle):
    """random gene collections / definitions / regions: the library and the oracle pick the same definition"""
    import variant_glue as vg
    rng = np.random.default_rng(77)
    for _trial in range(60):
        genes, pos = {}, 0
        for g in range(int(rng.integers(1, 5))):
            pos += int(rng.integers(5, 40))
            start, exons = pos, []
            for _x in range(int(rng.integers(1, 6))):
                e = pos + int(rng.integers(3, 20))
                exons.append({"start": pos, "end": e})
                pos = e + int(rng.integers(2, 20))
            genes[f"g{g}"] = {"coordinates": {"chrom": "c", "start": start, "end": exons[-1]["end"]}, "exons": exons,
                              "is_forward_strand": bool(rng.integers(0, 2))}
        names = sorted(genes)
        pick = lambda: [names[i] for i in sorted(rng.choice(len(names), size=int(rng.integers(1, len(names) + 1)), replace=False))]
        svs = {"full_gene_deletions": {}, "partial_gene_deletions": {}}
        for k in range(int(rng.integers(0, 4))):
            svs["full_gene_deletions"][f"f{k}"] = {"is_generic": bool(rng.integers(0, 2)), "full_genes_deleted": pick()}
        for k in range(int(rng.integers(0, 5))):
            ex = {}
            for g in pick():
                ne = len(genes[g]["exons"])
                a = int(rng.integers(0, ne)); b = int(rng.integers(a + 1, ne + 1))
                ex[g] = {"start": a, "end": b}
            svs["partial_gene_deletions"][f"p{k}"] = {"is_generic": bool(rng.integers(0, 2)), "exons_deleted": ex}
        defs = pkg.ffi.SvDefinitions(genes, svs)
        edges = sorted({0, pos + 10} | {x[k] for g in genes.values() for x in g["exons"] for k in ("start", "end")})
        for _q in range(40):
            a, b = sorted(int(v) for v in rng.choice(edges, size=2, replace=False))
            a, b = max(0, a - int(rng.integers(0, 2))), b + int(rng.integers(0, 2))
            assert defs.is_deletion(a, b) == vg.oracle_is_deletion(oracle, defs, a, b), (genes, svs, a, b)
