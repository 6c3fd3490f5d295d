"""The minimap2 restatement of the oracle (oracle/mm2.c) -- the "second opinion" on the base-level aligner.

Pinned here: (1) its DP against a brute-force two-piece affine alignment; (2) the cases the reference's own tests hold for
`Aligner::map`: identical sequence => (len, 0, 0) over the full span (`test_reference_alleles`,
/root/reference/src/hla/caller.rs:1710-1773), a 4-base read => no mapping (`test_score_bad_read`, :1784-1809), one mismatch ranks
strictly worse and N mismatches everything (`test_weight_sequence`, /root/reference/src/cyp2d6/chaining.rs:1051-1080); (3) the documented
behaviours that distinguish it from the library's unit-cost contract: end clipping at a = 1 and its absence at a = 5
(/root/reference/src/hla/caller.rs:1370-1379), z-drop, strand."""
import numpy as np
import pytest

import mm2_ffi


@pytest.fixture(scope="module")
def mm():
    return mm2_ffi.Mm2()


def rnd(rng, n):
    return "".join("ACGT"[i] for i in rng.integers(0, 4, n))


def mutate(rng, s, n_sub=0, n_ins=0, n_del=0, lo=0, hi=None):
    s = list(s)
    hi = len(s) if hi is None else hi
    for _ in range(n_sub):
        p = int(rng.integers(lo, hi))
        s[p] = "ACGT"[("ACGT".index(s[p]) + 1 + int(rng.integers(0, 3))) % 4]
    for _ in range(n_ins):
        p = int(rng.integers(lo, hi))
        s.insert(p, "ACGT"[int(rng.integers(0, 4))])
    for _ in range(n_del):
        p = int(rng.integers(lo, min(hi, len(s))))
        del s[p]
    return "".join(s)


def rescore(cigar, t, q, o):
    i = j = sc = 0
    for ln, op in cigar:
        if op in "M=X":
            for x in range(ln):
                a, b = t[i + x], q[j + x]
                sc += -o.sc_ambi if "N" in (a, b) else (o.a if a == b else -o.b)
            i += ln
            j += ln
        else:
            sc -= min(o.q + ln * o.e, o.q2 + ln * o.e2)
            if op == "I":
                j += ln
            else:
                i += ln
    return sc, i, j


def revcomp(s):
    return s[::-1].translate(str.maketrans("ACGTN", "TGCAN"))


def test_global_dp_equals_bruteforce(mm):
    rng = np.random.default_rng(11)
    o = mm.opts()
    for case in range(300):
        n = int(rng.integers(1, 70))
        t = rnd(rng, n)
        q = mutate(rng, t, n_sub=int(rng.integers(0, 4)), n_ins=int(rng.integers(0, 3)), n_del=int(rng.integers(0, 3)))
        if case % 5 == 0 and n > 45:                    # a long gap: the second affine piece (26 + l) is the cheaper one beyond 20 bases
            p = int(rng.integers(5, n - 40))
            q = t[:p] + t[p + int(rng.integers(21, 35)):]
        if case % 7 == 0:
            q = q[:len(q) // 2] + "N" + q[len(q) // 2 + 1:]
        if not q:
            q = "A"
        r = mm.dp(t, q, o, band=200, mode=0)
        assert r["score"] == mm.brute(t, q, o), (t, q)
        sc, i, j = rescore(r["cigar"], t, q, o)
        assert (sc, i, j) == (r["score"], len(t), len(q))


def test_global_dp_with_match_score_5(mm):
    rng = np.random.default_rng(12)
    o = mm.opts(a=5)
    for _ in range(100):
        t = rnd(rng, int(rng.integers(10, 60)))
        q = mutate(rng, t, 2, 1, 1)
        r = mm.dp(t, q, o, band=200, mode=0)
        assert r["score"] == mm.brute(t, q, o)
        assert rescore(r["cigar"], t, q, o) == (r["score"], len(t), len(q))


def test_extension_ends_at_the_best_cell_and_clips_a_near_end_mismatch(mm):
    rng = np.random.default_rng(13)
    t = rnd(rng, 120)
    r = mm.dp(t, t, mm.opts(), mode=1)
    assert (r["max"], r["t_end"], r["q_end"]) == (120, 120, 120) and r["cigar"] == [(120, "M")]
    q = t[:117] + ("A" if t[117] != "A" else "C") + t[118:]            # a mismatch three bases before the end: +2 - 4 < 0
    r = mm.dp(t, q, mm.opts(), mode=1)
    assert (r["max"], r["t_end"], r["q_end"]) == (117, 117, 117)
    r = mm.dp(t, q, mm.opts(a=5), mode=1)                               # a = 5 (score_read): 2 * 5 - 4 > 0, no clipping
    assert (r["t_end"], r["q_end"]) == (120, 120) and r["max"] == 5 * 119 - 4
    q = t[:110] + ("A" if t[110] != "A" else "C") + t[111:]            # nine matches behind the mismatch pay for it
    r = mm.dp(t, q, mm.opts(), mode=1)
    assert (r["t_end"], r["q_end"]) == (120, 120) and r["max"] == 119 - 4


def test_extension_z_drop(mm):
    rng = np.random.default_rng(14)
    head = rnd(rng, 300)
    t = head + rnd(rng, 1500)
    q = head + rnd(rng, 1500)                                            # unrelated tails: the score falls by > 400 and the extension stops
    r = mm.dp(t, q, mm.opts(), mode=1, band=751)
    assert r["zdropped"] == 1 and 300 <= r["t_end"] <= 330 and r["max"] >= 300
    r = mm.dp(t, q, mm.opts(zdrop=-1), mode=1, band=751)
    assert r["zdropped"] == 0


def test_reference_pinned_cases(mm):
    rng = np.random.default_rng(15)
    allele = rnd(rng, 3200)
    h = mm.map_pair(allele, allele)
    assert len(h) == 1
    h = h[0]
    assert (h["nm"], h["q_start"], h["q_end"], h["t_start"], h["t_end"], h["rev"], h["primary"]) == (0, 0, 3200, 0, 3200, 0, 1)
    assert h["cigar"] == [(3200, "=")]
    assert mm.map_pair(allele, "ACGT") == []                             # test_score_bad_read: no seed, no mapping
    # test_weight_sequence: 218-base strings, one mismatch is strictly worse; N in the target: every candidate pays it alike
    cons = [rnd(rng, 218)]
    cons.append(cons[0][:100] + ("A" if cons[0][100] != "A" else "G") + cons[0][101:])
    seg = cons[0]
    hits = [mm.map_pair(seg, c) for c in cons]
    assert [x[0]["nm"] for x in hits] == [0, 1]
    seg_n = seg[:100] + "N" + seg[101:]
    hits = [mm.map_pair(seg_n, c) for c in cons]
    assert [x[0]["nm"] for x in hits] == [1, 1]
    assert hits[0][0]["n_ambi"] == 1


def test_mismatches_indels_and_nm(mm):
    rng = np.random.default_rng(16)
    for _ in range(20):
        t = rnd(rng, 3000)
        n_sub, n_ins, n_del = int(rng.integers(0, 10)), int(rng.integers(0, 5)), int(rng.integers(0, 5))
        q = mutate(rng, t, n_sub, n_ins, n_del, lo=50, hi=2900)           # away from the ends: nothing to clip
        h = mm.map_pair(t, q)
        assert len(h) == 1
        h = h[0]
        assert h["nm"] <= n_sub + n_ins + n_del
        assert (h["q_start"], h["q_end"], h["t_start"], h["t_end"]) == (0, len(q), 0, 3000)
        cons_t = sum(ln for ln, op in h["cigar"] if op in "=XD")
        cons_q = sum(ln for ln, op in h["cigar"] if op in "=XI")
        assert (cons_t, cons_q) == (3000, len(q))
        assert h["nm"] == sum(ln for ln, op in h["cigar"] if op in "XID")


def test_end_clipping_shows_up_as_unmapped(mm):
    rng = np.random.default_rng(17)
    t = rnd(rng, 2000)
    q = ("A" if t[0] != "A" else "C") + t[1:1997] + ("A" if t[1997] != "A" else "C") + t[1998:]
    h = mm.map_pair(t, q)[0]
    assert (h["q_start"], h["q_end"], h["nm"]) == (1, 1997, 0)            # both terminal mismatches are clipped away, not counted
    h = mm.map_pair(t, q, mm.opts(a=5))[0]
    assert (h["q_start"], h["q_end"], h["nm"]) == (1, 2000, 1)            # the first base still has nothing to gain (5 - 4 - ... the mismatch itself is the end)


def test_read_in_flank_and_strand(mm):
    rng = np.random.default_rng(18)
    allele = rnd(rng, 3300)
    read = rnd(rng, 2100) + mutate(rng, allele, 3, 1, 1, lo=100, hi=3000) + rnd(rng, 1800)
    h = mm.map_pair(allele, read)
    assert len(h) == 1 and h[0]["rev"] == 0
    assert (h[0]["t_start"], h[0]["t_end"]) == (0, 3300) and h[0]["q_start"] == 2100 and h[0]["nm"] <= 5
    h = mm.map_pair(allele, revcomp(read))
    assert len(h) == 1 and h[0]["rev"] == 1 and (h[0]["t_start"], h[0]["t_end"]) == (0, 3300) and h[0]["nm"] <= 5
    assert h[0]["q_end"] == len(read) - 2100                                # query coordinates are reported on the read as given
    assert mm.map_pair(allele, revcomp(read), mm.opts(forward_only=1)) == []


def test_divergent_stretch_splits_the_alignment(mm):
    rng = np.random.default_rng(19)
    a, b = rnd(rng, 1500), rnd(rng, 1500)
    t = a + rnd(rng, 400) + b
    q = a + rnd(rng, 400) + b                                            # 400 unrelated bases in the middle: 300 mismatches, z-drop
    h = mm.map_pair(t, q)
    assert len(h) == 2 and all(x["nm"] <= 12 for x in h)
    spans = sorted((x["t_start"], x["t_end"]) for x in h)
    assert spans[0][0] == 0 and 1500 <= spans[0][1] <= 1530 and 1870 <= spans[1][0] <= 1900 and spans[1][1] == 3400


def test_index_of_many_alleles_picks_the_source(mm):
    rng = np.random.default_rng(20)
    base = rnd(rng, 3000)
    alleles = [mutate(rng, base, n_sub=int(rng.integers(3, 25))) for _ in range(60)]
    idx = mm2_ffi.Index(mm, alleles)
    for src in (0, 17, 59):
        read = rnd(rng, 900) + mutate(rng, alleles[src], 2, 1, 0, lo=200, hi=2800) + rnd(rng, 700)
        hits = idx.map(read)
        assert 1 <= len(hits) <= 6                                       # primary + at most best_n secondaries
        best = min(hits, key=lambda h: (max(h["nm"], 0.1) / (h["t_end"] - h["t_start"]), h["rid"]))
        assert best["rid"] == src or alleles[best["rid"]] == alleles[src]
    idx.close()


# ------------------------------------------------------------------ the stages one by one, and the statement of K1's seeded mode

def test_stages_agree_with_the_map(mm):
    """omm_sketch / omm_anchors / omm_chain_stage are omm_map's own stages: the chains selected are the mappings that come back (same targets, in the chain
    order, when every selected chain survives the filters), minimizers come in position order and a (w,k)-window never goes without one"""
    rng = np.random.default_rng(21)
    base = rnd(rng, 3200)
    alleles = [mutate(rng, base, n_sub=int(rng.integers(2, 20))) for _ in range(80)]
    idx = mm2_ffi.Index(mm, alleles)
    read = rnd(rng, 700) + mutate(rng, alleles[11], 3, 1, 1, lo=100, hi=3000) + rnd(rng, 500)
    h, p, st = mm.sketch(read)
    assert len(h) > len(read) // 12 and np.all(np.diff(p) > 0) and p[0] >= 18 and np.all(np.diff(p) <= 19) and set(st.tolist()) <= {0, 1}
    x, y = idx.anchors(read)
    assert len(x) > 0 and np.all((x[1:] > x[:-1]) | ((x[1:] == x[:-1]) & (y[1:] >= y[:-1])))            # an_cmp's order
    regs, stats = idx.chain_stage(read)
    assert stats[0] == len(h) and stats[3] == len(x) and stats[5] == len(regs) and stats[6] == int((regs[:, 9] > 0).sum())
    assert np.all(np.diff(regs[:, 2]) <= 0)                                                             # by chain score, descending
    sel = regs[regs[:, 9] > 0]
    sel = sel[np.argsort(sel[:, 9])]
    hits = idx.map(read)
    assert sorted(int(r) for r in sel[:, 0]) == sorted(h2["rid"] for h2 in hits) and 1 <= len(hits) <= 6
    assert int(sel[0, 8]) >= 0
    idx.close()


def test_k1_seeded_statement_on_the_port_fixture(oracle):
    """omm_hla_k1_seeded -- minimap2's seeding / chaining / selection + the library's cell and re-score for the selected chains -- names the allele the port's
    seeded map names (base-level alignment by the restatement's own DP, tests/cpu_port_seeded.py), with its (NM, allele span): 40 reads of configs[1] spread
    over the committed fixture (tests/golden/concordance.json.gz; the GPU test holds all 10,000 to it)"""
    import gzip
    import json
    import os
    import __graft_entry__ as ge
    ge.load_package()
    from pb_starphase_amd import synth
    import hla_expected as hx
    doc = json.load(gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "concordance.json.gz"), "rt"))["hla"]
    fx = synth.HlaFixture()
    wl = synth.Config2Workload(fx, n_reads=10000, seed=1000)
    idx, dna_ids = hx.seed_index(oracle, fx)
    assert idx.mid_occ == 500 and idx.mm.L.omm_index_n_minimizers(idx.h) == 3582682
    for r in range(0, 10000, 250):
        pick, hits, n_chains = idx.k1_seeded(wl.reads[r])
        assert pick >= 0 and not hits[pick]["rev"] and len(hits) == 6 and n_chains > 500
        h = hits[pick]
        assert dna_ids[int(h["rid"])] == doc["winner"][r], r
        assert (int(h["nm"]), int(h["t_end"] - h["t_start"])) == (doc["nm"][r], doc["span"][r]), r
        assert [int(x["dp_max"]) for x in hits] == sorted((int(x["dp_max"]) for x in hits), reverse=True)       # output order: by peak score


def _islands():
    import json, os
    d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "chr6_hla_islands.json")))
    return {(i["start"], i["end"]): i["sequence"] for i in d["islands"]}


def _faux_alleles():
    """tests/golden/HLA-faux/hla_gen.fa: the two genomic alleles of the reference's test_data/HLA-faux (data: A*01:01:01:01, B*07:02:01:01)"""
    import os
    out, name = {}, None
    for line in open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "HLA-faux", "hla_gen.fa")):
        line = line.strip()
        if line.startswith(">"):
            name = line[1:].split()[1]
            out[name] = []
        elif line:
            out[name].append(line.upper())
    return {k: "".join(v) for k, v in out.items()}


# RefSeq gene records of the reference's test_data/refseq_faux/refseq_small.gff.gz (1-based inclusive, as GFF has them) and what HlaConfig::new must extend them to
# (HlaConfig::default_gene_collection, /root/reference/src/hla/alleles.rs:232-243)
HLACONFIG_CASES = (("A*01:01:01:01", 29942532, 29945870, (29942253, 29945870)), ("B*07:02:01:01", 31353875, 31357179, (31353361, 31357442)))


def hlaconfig_extend(map_best, islands, alleles):
    """HlaConfig::new for one allele per gene (/root/reference/src/hla/alleles.rs:108-196): the RefSeq record +- 2,000 bases is the target, the allele the query; of the
    mappings the one with the lowest (nm + unmapped) / len wins (strict <, first wins ties); the gene's coordinates are extended to cover it.
    map_best(target, query) -> list of mappings with q_start, q_end, t_start, t_end, nm"""
    out = {}
    for name, gff_start, gff_end, _want in HLACONFIG_CASES:
        start, end = gff_start - 1, gff_end                                   # 0-based half-open
        a_start, a_end = start - 2000, end + 2000
        (i_start, i_end), seq = next((k, v) for k, v in islands.items() if k[0] <= a_start and a_end <= k[1])
        target = seq[a_start - i_start:a_end - i_start]
        query = alleles[name]
        best, best_score = None, 1.0                                          # MappingStats::new(ref_len, ref_len, 0): score 1.0
        for m in map_best(target, query):
            unmapped = len(query) - (m["q_end"] - m["q_start"])
            score = max(m["nm"] + unmapped, 0.1) / len(query)                 # MappingStats::mapping_score, penalised (src/data_types/mapping.rs:60-84)
            if score < best_score:
                best, best_score = m, score
        assert best is not None, name
        out[name] = (min(start, a_start + best["t_start"]), max(end, a_start + best["t_end"]), best)
    return out


def test_hlaconfig_new(mm):
    """The reference's one real-data pin of minimap2 (`test_hlaconfig_new`, /root/reference/src/hla/alleles.rs:512-547): two real IMGT alleles mapped onto the RefSeq records
    +- 2,000 bases of the chr6 islands must extend the coordinates to HlaConfig::default()'s -- the allele that is 42 edits from the reference included, i.e. with
    minimap2's end clipping on real divergence."""
    got = hlaconfig_extend(lambda t, q: mm.map_pair(t, q), _islands(), _faux_alleles())
    for name, _s, _e, want in HLACONFIG_CASES:
        assert got[name][:2] == want, (name, got[name])
    a, b = got["A*01:01:01:01"][2], got["B*07:02:01:01"][2]
    assert (a["rev"], a["nm"]) == (0, 42) and (b["rev"], b["nm"]) == (1, 0)          # HLA-A lies on the forward strand of hg38, HLA-B on the reverse one
    assert 2000 - (29942532 - 1 - 29942253) == a["t_start"] and a["t_end"] == 29945755 - (29942531 - 2000)
