"""f4, the debug files (host only): `hla_debug.json` (HlaDebug, /root/reference/src/hla/debug.rs:7-221), `cyp2d6_alleles.json` (DeeplotypeDebug,
src/cyp2d6/debug.rs:10-70) and the CIGAR / MD strings DetailedMappingStats copies from minimap2 -- layout as serde_json's pretty printer
writes them (save_json, src/util/file_io.rs:37-52), float text as ryu prints f64."""
import ctypes as C
import gzip
import json
import re

import numpy as np
import pytest


@pytest.fixture(scope="module")
def D(pkg):
    return pkg.database


# ---------------------------------------------------------------- CIGAR / MD
def random_alignment(rng, n=400, rate=0.06):
    """target, query and the edit events between them, drawn column by column"""
    target = "".join(rng.choice(list("ACGT"), n))
    b_start, b_end = int(rng.integers(0, 20)), n - int(rng.integers(0, 20))
    cols, events, query = [], [], []
    j = b_start
    while j < b_end:
        u = rng.random()
        if u < rate / 3:
            alt = rng.choice([c for c in "ACGT" if c != target[j]])
            cols.append(("X", target[j])); events.append((0 << 30) | j); query.append(alt); j += 1
        elif u < 2 * rate / 3:
            cols.append(("D", target[j])); events.append((1 << 30) | j); j += 1
        elif u < rate and cols:
            cols.append(("I", None)); events.append((2 << 30) | j); query.append(rng.choice(list("ACGT")))
        else:
            cols.append(("=", target[j])); query.append(target[j]); j += 1
    return target, "".join(query), b_start, b_end, cols, np.array(events, np.uint32)


def expected_strings(cols):
    ops = "".join("M" if c in "=X" else c for c, _ in cols)
    cigar = "".join(f"{len(m.group(0))}{m.group(0)[0]}" for m in re.finditer(r"M+|I+|D+", ops))
    md, run, in_del = "", 0, False
    for c, base in cols:
        if c == "=":
            run += 1; in_del = False
        elif c == "X":
            md += f"{run}{base}"; run = 0; in_del = False
        elif c == "D":
            md += base if in_del else f"{run}^{base}"
            run = 0; in_del = True
    return cigar, md + str(run), sum(1 for c, _ in cols if c == "=")


def test_cigar_and_md_of_random_alignments(D):
    rng = np.random.default_rng(5)
    for _ in range(200):
        target, query, b_start, b_end, cols, events = random_alignment(rng)
        aln = dict(ok=1, nm=len(events), a_start=0, a_end=len(query), b_start=b_start, b_end=b_end, a_len=len(query), b_len=len(target))
        cigar, md, match_len = D.aln_strings(aln, events, target)
        assert (cigar, md, match_len) == expected_strings(cols)
        # the strings describe the alignment: CIGAR spans both sequences, MD rebuilds the target from the query
        ops = [(int(n), op) for n, op in re.findall(r"(\d+)([MID])", cigar)]
        assert sum(n for n, op in ops if op in "MD") == b_end - b_start and sum(n for n, op in ops if op in "MI") == len(query)
        aligned_query = []                                       # query bases of the M columns
        q = 0
        for n, op in ops:
            if op == "M":
                aligned_query += list(query[q:q + n])
            if op in "MI":
                q += n
        rebuilt, k = [], 0
        for num, dele, mis in re.findall(r"(\d+)|\^([ACGT]+)|([ACGT])", md):
            if num:
                rebuilt += aligned_query[k:k + int(num)]; k += int(num)
            elif dele:
                rebuilt += list(dele)
            else:
                rebuilt.append(mis); k += 1
        assert "".join(rebuilt) == target[b_start:b_end]


def test_cigar_edge_cases(D, pkg):
    target = "ACGTACGTAC"
    full = dict(ok=1, nm=0, a_start=0, a_end=10, b_start=0, b_end=10, a_len=10, b_len=10)
    assert D.aln_strings(full, [], target) == ("10M", "10", 10)
    ev = [(0 << 30) | 0, (1 << 30) | 1, (1 << 30) | 2, (0 << 30) | 3, (2 << 30) | 4]       # X at 0, DD at 1-2, X at 3, I before 4
    aln = dict(full, nm=5, a_end=9, a_len=9)
    assert D.aln_strings(aln, ev, target) == ("1M2D1M1I6M", "0A0^CG0T6", 6)
    with pytest.raises(pkg.StarphaseError):
        D.aln_strings(dict(full, nm=1), [(0 << 30) | 11], target)                            # an event outside the aligned span
    with pytest.raises(pkg.StarphaseError):
        D.aln_strings(dict(full, ok=0), [], target)


# ---------------------------------------------------------------- hla_debug.json
def mapping(nm, cigar, md):
    return dict(query_len=100, target_len=120, match_len=100 - nm, nm=nm, query_unmapped=0, target_unmapped=20, cigar=cigar, md=md)


def test_hla_debug_layout(D, pkg, tmp_path):
    dbg = D.HlaDebug()
    dbg.add_read("HLA-B", "read/2", "HLA:HLA00132", "07:02:01:01")
    dbg.add_read("HLA-A", "read/10")                                                  # ReadMappingStats::new(): no best match, no mappings
    dbg.add_read("HLA-A", "consensus1", "HLA:HLA00001", "01:01:01:01")
    dbg.add_mapping("HLA-A", "consensus1", "HLA:HLA00005", cdna=mapping(2, "100M", "10A20C68"), dna=None)
    dbg.add_mapping("HLA-A", "consensus1", "HLA:HLA00001", cdna=mapping(0, "100M", "100"), dna=mapping(1, "50M1D49M", "50^A49"))
    call = pkg.ffi.sp_hla_call()
    call.is_dual, call.dual_passed, call.counts1, call.counts2, call.maf, call.cdf = 1, 1, 25, 15, 0.375, 0.04424778761061947
    dbg.add_dual_stats("HLA-A", call)
    dbg.add_dual_stats("HLA-B", pkg.ffi.sp_hla_call())
    text = dbg.json()
    expected = {
        "read_mapping_stats": {
            "HLA-A": {
                "consensus1": {"best_match_id": "HLA:HLA00001", "best_match_star": "01:01:01:01", "mapping_stats": {
                    "HLA:HLA00001": {"cdna_mapping": mapping(0, "100M", "100"), "dna_mapping": mapping(1, "50M1D49M", "50^A49")},
                    "HLA:HLA00005": {"cdna_mapping": mapping(2, "100M", "10A20C68"), "dna_mapping": None}}},
                "read/10": {"best_match_id": None, "best_match_star": None, "mapping_stats": {}}},
            "HLA-B": {"read/2": {"best_match_id": "HLA:HLA00132", "best_match_star": "07:02:01:01", "mapping_stats": {}}}},
        "dual_passing_stats": {
            "HLA-A": {"is_passing": True, "is_dual": True, "counts1": 25, "counts2": 15, "maf": 0.375, "cdf": 0.04424778761061947},
            "HLA-B": {"is_passing": False, "is_dual": False, "counts1": None, "counts2": None, "maf": None, "cdf": None}}}
    assert text == json.dumps(expected, indent=2)                # key order, field order and indentation
    out = tmp_path / "hla_debug.json.gz"
    dbg.save(str(out))
    assert json.load(gzip.open(out)) == expected
    with pytest.raises(pkg.StarphaseError, match="Entry read/10 is already occupied"):
        dbg.add_read("HLA-A", "read/10")
    with pytest.raises(pkg.StarphaseError, match="Entry HLA:HLA00001 is already occupied!"):
        dbg.add_mapping("HLA-A", "consensus1", "HLA:HLA00001")
    with pytest.raises(pkg.StarphaseError, match="Entry HLA-A is already occupied"):
        dbg.add_dual_stats("HLA-A", call)


def test_hla_debug_without_dual_stats_is_null(D):
    assert json.loads(D.HlaDebug().json()) == {"read_mapping_stats": {}, "dual_passing_stats": None}


@pytest.mark.parametrize("value,text", [
    (0.25, "0.25"), (1.0, "1.0"), (1e-7, "1e-7"), (1.5e-7, "1.5e-7"), (1e-5, "0.00001"), (1e-6, "1e-6"), (1e15, "1000000000000000.0"),
    (1e16, "1e16"), (123456789.0, "123456789.0"), (0.1 + 0.2, "0.30000000000000004"), (5e-324, "5e-324"), (12345.678, "12345.678"),
    (1.2345678901234568e17, "1.2345678901234568e17"), (0.0, "0.0")])
def test_f64_text_is_ryu(D, pkg, value, text):
    """serde_json prints f64 through ryu: the shortest digits that round-trip, exponent form outside 1e-5 <= |v| < 1e16"""
    call = pkg.ffi.sp_hla_call()
    call.is_dual, call.maf, call.cdf = 1, value, value
    got = D.HlaDebug().add_dual_stats("G", call).json()
    assert f'"maf": {text},' in got and float(text) == value


# ---------------------------------------------------------------- cyp2d6_alleles.json
def test_cyp_alleles_layout(pkg):
    ffi = pkg.ffi
    labels = ["rs1", "rs2", "22:42126611 C>G", "rs4"]
    pr = ffi.sp_cyp_problem()
    arr = (C.c_char_p * 4)(*[x.encode() for x in labels])
    vi = np.array([1, 0, 1, 0], np.uint8)
    pr.n_variants, pr.var_label, pr.var_is_vi = 4, arr, vi.ctypes.data
    call = ffi.sp_cyp_call()
    call.n_consensus = 12
    types = {0: (1, b""), 2: (2, b"4.001"), 10: (2, b"1.001"), 11: (9, b"2.001"), 3: (6, b"")}          # REP6, CYP2D6*4.001, *1.001, FalseAllele, CYP2D7
    for h, (t, sub) in types.items():
        call.cons_type[h] = t
        call.cons_subtype[h].value = sub
    call.deep1, call.hap1, call.core1 = b"(2_CYP2D6*4.001 -rs2 +rs4)", b"*4.001", b"*4"
    call.deep2, call.hap2, call.core2 = b"(10_CYP2D6*1.001)", b"*1.001", b"*1"
    state = np.full((ffi.SP_CYP_MAXCONS, 4), 255, np.uint8)
    state[2] = [1, 3, 6, 2]                  # Match, Missing, UnknownUnexpected, Unexpected
    state[11] = [255, 5, 255, 4]             # a duplicate marked FalseAllele keeps its list
    rv = ffi.sp_cyp_region_variants()
    rv.state = state.ctypes.data
    for h in (2, 10, 11):
        rv.has_variants[h] = 1
    need = C.c_uint64(0)
    assert ffi.lib().sp_cyp_alleles_json(C.byref(pr), C.byref(call), C.byref(rv), None, 0, C.byref(need)) == 0
    buf = C.create_string_buffer(need.value)
    assert ffi.lib().sp_cyp_alleles_json(C.byref(pr), C.byref(call), C.byref(rv), buf, need.value, None) == 0
    rvj = lambda i, s: {"label": labels[i], "is_vi": bool(vi[i]), "variant_state": s}
    expected = {
        "hap1": {"deep_form": "(2_CYP2D6*4.001 -rs2 +rs4)", "suballele_form": "*4.001", "core_form": "*4"},
        "hap2": {"deep_form": "(10_CYP2D6*1.001)", "suballele_form": "*1.001", "core_form": "*1"},
        "alleles": {                         # BTreeMap<String, _>: "10_" < "11_" < "2_"
            "10_CYP2D6*1.001": [],
            "11_FalseAllele_2.001": [rvj(1, "AmbiguousMissing"), rvj(3, "AmbiguousUnexpected")],
            "2_CYP2D6*4.001": [rvj(0, "Match"), rvj(1, "Missing"), rvj(2, "UnknownUnexpected"), rvj(3, "Unexpected")]}}
    assert buf.value.decode() == json.dumps(expected, indent=2)
    small = C.create_string_buffer(16)
    assert ffi.lib().sp_cyp_alleles_json(C.byref(pr), C.byref(call), C.byref(rv), small, 16, None) == 6     # SP_ERR_CAPACITY
