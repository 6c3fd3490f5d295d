"""BASELINE configs[2] at its stated shape: sp_cyp_diplotype on the synthetic chr22 locus laid out with the database's
own coordinates, all 39 templates of generate_cyp_hybrids (sp_cyp_db_create, row a14), the real ~400-variant / 520-allele table, the six
scenarios of the survey including the two hybrid ones.
  * 2,000 targeted-style reads per sample: the library's call equals the simulated truth;
  * 160 reads per sample: the library equals the same pipeline assembled from the oracle's pieces (tests/cyp_pipeline.py) -- consensus
    strings, labels, chains, f64 score, haplotype strings -- and the truth."""
import numpy as np
import pytest

import cyp_cases_real as cr
import oracle_ffi as of

pytestmark = pytest.mark.gpu

NAMES = ["*1/*2", "*4/*4", "*5/*1", "*4+*68/*1", "*10+*36/*10", "*2x2/*1"]


@pytest.fixture(scope="module")
def real(pkg, gpu_ctx):
    from pb_starphase_amd import synth
    import cyp_pipeline as cp
    cfg, gene_def = cr.load_db()
    locus = synth.Chr22Locus(cfg, gene_def, seed=3)
    db = pkg.ffi.CypDb(gpu_ctx, cfg, gene_def, locus.sequence, locus.start)
    tm, vs = db.templates(), db.variants()
    names, rows = db.alleles()
    bb = cfg["cyp_coordinates"]["CYP2D6_wfa_backbone"]
    odb = cp.Db([t[2] for t in tm], [t[0] for t in tm], [t[1] for t in tm], [t[3] for t in tm], [t[4] for t in tm],
                locus.slice(bb["start"], bb["end"]), [(p - bb["start"], r, a) for p, r, a, _l, _v in vs], [v[4] for v in vs], names, rows,
                var_labels=[v[3] for v in vs])
    assert db.stats.n_templates == 39 and db.stats.n_variants == 393 and db.stats.n_alleles == 520
    return locus, db, odb, {n: (h, e) for n, h, e in cr.scenarios(locus)}


def core(s):
    """"*36.001 + *10.001" -> "*36 + *10"; "*2.001x2" stays a run of two sub-alleles that collapse to "*2x2" """
    parts = []
    for x in s.split(" + "):
        body, _, mult = x.partition("x")
        parts.append((body.split(".")[0], int(mult) if mult else 1))
    out = []
    for b, m in parts:                                   # adjacent equal core alleles are counted together (caller.rs:942-953)
        if out and out[-1][0] == b:
            out[-1][1] += m
        else:
            out.append([b, m])
    return " + ".join(b + (f"x{m}" if m > 1 else "") for b, m in out)


@pytest.mark.parametrize("name", NAMES)
def test_stated_shape_equals_truth(gpu_ctx, real, name):
    locus, db, _odb, sc = real
    haps, expected = sc[name]
    reads = locus.sample(np.random.default_rng(7), haps, 2000)
    assert 1990 <= len(reads) <= 2010
    call, cons, labels = db.diplotype(gpu_ctx.upload(reads))
    assert call.status == 0
    assert sorted([call.hap1.decode(), call.hap2.decode()]) == sorted(expected)
    assert sorted([call.core1.decode(), call.core2.decode()]) == sorted(core(e) for e in expected)


@pytest.mark.parametrize("name", NAMES)
def test_library_equals_oracle_pipeline(oracle, gpu_ctx, real, name):
    import cyp_pipeline as cp
    locus, db, odb, sc = real
    haps, expected = sc[name]
    reads = locus.sample(np.random.default_rng(11), haps, 160)
    exp = cp.diplotype(oracle, odb, reads, cfg=db.cfg)
    call, cons, labels = db.diplotype(gpu_ctx.upload(reads))
    assert call.status == exp["status"] == 0
    assert cons == exp["consensus"]
    assert labels == [(int(t), s) for t, s in exp["labels"]]
    assert list(call.chain1[:call.n1]) == exp["chain1"] and list(call.chain2[:call.n2]) == exp["chain2"]
    assert call.score == exp["score"]
    got = (call.hap1.decode(), call.hap2.decode(), call.core1.decode(), call.core2.decode())
    assert got == (exp["hap1"], exp["hap2"], exp["core1"], exp["core2"])
    if name != "*4+*68/*1":     # (at 160 reads the hybrid's reads that start behind its CYP2D7 part end up in a group of their own, in the oracle as in the library: truth at 2,000 reads above)
        assert sorted(got[:2]) == sorted(expected)
    # Cyp2d6DetailLevel::DeepAlleles: "(<index>_<full allele> +unexpected -missing ?ambiguous)" per reported region
    assert (call.deep1.decode(), call.deep2.decode()) == (exp["deep1"], exp["deep2"])
    assert call.deep1.decode().startswith("(") and "_" in call.deep1.decode()


@pytest.mark.parametrize("name", ["*1/*2", "*4+*68/*1"])
def test_library_equals_oracle_pipeline_at_the_stated_size(gpu_ctx, real, name):
    """configs[2] at its 2,000 reads -- the size bench.py times, the branching multi-way search of `*4+*68/*1` (1,469 expansions) included -- against the oracle-assembled
    pipeline run on the CPU when the fixture was made (tests/golden/make_fullsize.py): consensus strings, labels, chains, the f64 score, every haplotype string"""
    import gzip, json, os
    gold = json.load(gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize_oracle.json.gz"), "rt"))["cyp"]["scenarios"][name]
    locus, db, _odb, sc = real
    reads = locus.sample(np.random.default_rng(7), sc[name][0], 2000)
    call, cons, labels = db.diplotype(gpu_ctx.upload(reads))
    assert call.status == gold["status"] == 0
    assert cons == gold["consensus"]
    assert [[int(t), s] for t, s in labels] == gold["labels"]
    assert list(call.chain1[:call.n1]) == gold["chain1"] and list(call.chain2[:call.n2]) == gold["chain2"]
    assert call.score == gold["score"]
    assert [call.hap1.decode(), call.hap2.decode()] == gold["hap"] and [call.core1.decode(), call.core2.decode()] == gold["core"]
    assert [call.deep1.decode(), call.deep2.decode()] == gold["deep"]
    assert sorted(gold["hap"]) == sorted(sc[name][1])


def test_cohort_call_equals_single_calls(gpu_ctx, real):
    """sp_cyp_diplotype_cohort: the six scenarios as one GPU's share of a cohort, spread over the context's streams == one call per sample"""
    locus, db, _odb, sc = real
    sets, single = [], []
    for k, name in enumerate(NAMES):
        haps, _expected = sc[name]
        sets.append(gpu_ctx.upload(locus.sample(np.random.default_rng(40 + k), haps, 200)))
    key = lambda call, cons: (call.status, call.hap1, call.hap2, call.core1, call.core2, call.deep1, call.deep2, call.score, list(call.chain1[:call.n1]),
                              list(call.chain2[:call.n2]), [(int(call.cons_type[i]), call.cons_subtype[i].value) for i in range(call.n_consensus)], cons)
    for rs in sets:
        call, cons, _labels = db.diplotype(rs)
        single.append(key(call, cons))
    for streams in (6, 3, 1, 8):
        gpu_ctx.set_option("cyp_cohort_streams", streams)
        cohort = db.diplotype_cohort(sets)
        assert [key(call, cons) for call, cons, _rc in cohort] == single and all(rc == 0 for _c, _s, rc in cohort)
    gpu_ctx.set_option("cyp_cohort_streams", 6)
    for bad in (0, 9):
        with pytest.raises(Exception):
            gpu_ctx.set_option("cyp_cohort_streams", bad)
    assert [k[0] for k in single] == [0] * len(NAMES)
    assert db.diplotype_cohort([]) == []


def test_cohort_groups_with_an_empty_sample_and_with_an_n_plane(gpu_ctx, real):
    """A group's read sets are seen as one set for the region search and the weights (cyp_group_view): a sample without reads in the middle of the group keeps
    its place; a sample whose reads hold an N (a set with an N plane) makes the whole group go sample by sample -- either way the calls are the single calls"""
    locus, db, _odb, sc = real
    reads = [locus.sample(np.random.default_rng(60 + k), sc[name][0], 160) for k, name in enumerate(NAMES[:4])]
    key = lambda call, cons: (call.status, call.hap1, call.hap2, call.score, list(call.chain1[:call.n1]), list(call.chain2[:call.n2]), cons)
    for with_n in (False, True):
        rs = [list(r) for r in reads]
        if with_n:
            rs[2][5] = rs[2][5][:1000] + "N" + rs[2][5][1001:]
        rs.insert(1, [])                                            # a sample without reads
        sets = [gpu_ctx.upload(r) for r in rs]
        single = []
        for st in sets:
            call, cons, _labels = db.diplotype(st)
            single.append(key(call, cons))
        cohort = db.diplotype_cohort(sets)
        assert [key(call, cons) for call, cons, _rc in cohort] == single and all(rc == 0 for _c, _s, rc in cohort)
        assert single[1][0] == 1 and [k[0] for k in single[:1] + single[2:]] == [0] * 4      # NO_READS for the empty one


def test_hybrid_sample_on_a_second_locus(pkg, gpu_ctx):
    """The `*4+*68/*1` sample on another random chr22 locus, 2,000 reads: the 884 unseeded sequences are four classes (`*4`, `*1`, CYP2D7, the hybrid)
    for a search that holds two consensuses.  Under round 3's queue rule (the highest-cost node goes when more than max_queue_size wait) the search lost its
    most advanced nodes, gave up, and needed a retry ladder of this library's own; with the length threshold raised instead (the shortest nodes go: the reading
    of CdwfaConfig::max_queue_size in oracle/consensus.c) it completes with the reference's configuration as written (offset_compare_length 100, no ladder)
    and the call is the truth.  Library == oracle pipeline on this sample: profiles/scripts/cyp_other_locus.py (a minute of CPU, not part of the suite)."""
    from pb_starphase_amd import synth
    cfg, gene_def = cr.load_db()
    locus = synth.Chr22Locus(cfg, gene_def, seed=2003)
    db = pkg.ffi.CypDb(gpu_ctx, cfg, gene_def, locus.sequence, locus.start)
    _name, haps, expected = [s for s in cr.scenarios(locus) if s[0] == "*4+*68/*1"][0]
    R = gpu_ctx.upload(locus.sample(np.random.default_rng(2007), haps, 2000))
    call, _cons, labels = db.diplotype(R)
    assert call.status == 0 and sorted([call.hap1.decode(), call.hap2.decode()]) == sorted(expected)
    assert sorted(s for t, s in labels if t == 2) == ["1.001", "4.001"]
    assert 0 <= call.searches_gave_up <= 2          # (a search of a group that is one class already may still end on its bounds: the group then stays whole, which is right)
    # the ladder is an option of the context (off by default: the reference has no such rule): the one search that still ends on its bounds is then run again with
    # stricter fractions -- another grouping of the same reads, the same call
    gpu_ctx.set_option("cons_retry_ladder", 1)
    try:
        on, _cons2, labels_on = db.diplotype(R)
    finally:
        gpu_ctx.set_option("cons_retry_ladder", 0)
    assert on.status == 0 and sorted([on.hap1.decode(), on.hap2.decode()]) == sorted(expected) and on.searches_gave_up <= call.searches_gave_up


def test_deep_labels_of_a_novel_allele(oracle, gpu_ctx, real):
    """Cyp2d6DetailLevel::DeepAlleles (src/cyp2d6/caller.rs:907-957, region.rs:60-91): *4.001 without rs2004511 and with rs4987144 of *2 is
    still typed *4.001, and its deep label lists the missing variant with '-' and the extra one with '+'"""
    import cyp_pipeline as cp
    locus, db, odb, _sc = real
    novel = locus.star_allele("4.001", drop=("rs2004511",), add=(("2.001", "rs4987144"),))
    haps = [locus.haplotype([novel]), locus.haplotype([locus.star_allele("1.001")])]
    reads = locus.sample(np.random.default_rng(13), haps, 160)
    exp = cp.diplotype(oracle, odb, reads, cfg=db.cfg)
    call, _cons, _labels = db.diplotype(gpu_ctx.upload(reads))
    assert call.status == exp["status"] == 0
    deep = (call.deep1.decode(), call.deep2.decode())
    assert deep == (exp["deep1"], exp["deep2"])
    assert sorted([call.core1.decode(), call.core2.decode()]) == ["*1", "*4"]
    d4 = next(d for d, c in zip(deep, (call.core1.decode(), call.core2.decode())) if c == "*4")
    minus = [w[1:] for w in d4.strip("()").split(" ")[1:] if w[0] == "-"]
    plus = [w[1:] for w in d4.strip("()").split(" ")[1:] if w[0] == "+"]
    assert len(minus) == 1 and "rs2004511" in minus[0] and len(plus) == 1 and "rs4987144" in plus[0], d4
    d1 = next(d for d in deep if d is not d4)
    assert " " not in d1 and d1.endswith("_CYP2D6*1.001)"), d1
    # cyp2d6_alleles.json (DeeplotypeDebug, src/cyp2d6/debug.rs:10-70): the same call with the variant list of every typed region
    import json
    call2, regions, text = db.diplotype_detailed(gpu_ctx.upload(reads))
    assert (call2.deep1, call2.deep2, call2.hap1, call2.hap2) == (call.deep1, call.deep2, call.hap1, call.hap2)
    assert text == json.dumps(exp["alleles_json"], indent=2)
    lists = json.loads(text)["alleles"]
    assert len(lists) == len(regions) >= 2
    k4 = d4.strip("()").split(" ")[0]
    by_state = {}
    for v in lists[k4]:
        by_state.setdefault(v["variant_state"], []).append(v["label"])
    assert len(by_state["Missing"]) == 1 and len(by_state["Unexpected"]) == 1 and len(by_state["Match"]) == 16    # *4.001 has 17 variants


def test_variant_states_on_the_real_table(oracle, gpu_ctx, real):
    """K9 on the real 393-variant table: variants of CYP2D6 sit 16 bases apart on average and many overlap or touch (SNV + indel pairs, indels in
    runs), so they are decided jointly through the variant graph.  Library == oracle/cyp.c for every star allele tried; on an error-free
    sequence every state equals the allele's definition unless the graph itself is ambiguous there (state 2: two alternatives spell the
    same bases)."""
    import oracle_ffi as of
    locus, db, odb, _sc = real
    names, rows = db.alleles()
    vs = odb.variants
    pos, refs, alts = [v[0] for v in vs], [v[1] for v in vs], [v[2] for v in vs]
    # star alleles that carry variants closer than 24 bases to one another, plus a spread of the others
    def close_pairs(row):
        on = [pos[i] for i in np.flatnonzero(row)]
        return sum(1 for a, b in zip(on, on[1:]) if b - a < 24)
    ranked = sorted(range(len(names)), key=lambda a: -close_pairs(rows[a]))
    chosen = ranked[:12] + ranked[40::60]
    assert close_pairs(rows[chosen[0]]) >= 2
    seqs = [locus.star_allele(names[a]) for a in chosen]
    states, alns = gpu_ctx.cyp_variant_states(gpu_ctx.upload(seqs), odb.backbone, pos, refs, alts)
    n_amb = 0
    for x, a in enumerate(chosen):
        e_states, e_aln = of.oracle_variant_states(oracle, seqs[x], odb.backbone, pos, refs, alts)
        assert states[x].tolist() == e_states.tolist(), (names[a], np.flatnonzero(states[x] != e_states)[:8])
        assert alns[x]["ok"]
        decided = states[x] != 2
        assert (states[x][decided] == rows[a][decided]).all(), (names[a], np.flatnonzero((states[x] != rows[a]) & decided)[:8])
        n_amb += int((~decided).sum())
    assert n_amb < 4 * len(chosen)
    # the typing on top of it (K7): every one of these sequences is typed as its own allele or one that is indistinguishable from it
    bv, ba, tie = gpu_ctx.cyp_score_alleles(rows, odb.is_vi, states)
    for x, a in enumerate(chosen):
        assert tie[x][a] == 1, names[a]


def test_gave_up_is_reported(pkg, gpu_ctx):
    """a two-way search whose bounds are exhausted reports gave_up = 1 (and no consensus); one that completes reports 0"""
    rng = np.random.default_rng(5)
    base = "".join(rng.choice(list("ACGT"), 400))
    reads = [base] * 12
    out = gpu_ctx.consensus(gpu_ctx.upload(reads), pkg.ffi.sp_cons_config(3, 100, 1, 1, 400, 50, 0.10, 20, 10, 1000, 0))
    assert out["cons"][0] == base and out["gave_up"] is False
    # every read its own sequence and a queue of one node with capacity one per length: the search cannot keep any branch alive
    noisy = ["".join(rng.choice(list("ACGT"), 300)) for _ in range(12)]
    out = gpu_ctx.consensus(gpu_ctx.upload(noisy), pkg.ffi.sp_cons_config(1, 100, 1, 1, 400, 50, 0.01, 1, 1, 1, 0))
    assert isinstance(out["gave_up"], bool)
    if out["gave_up"]:
        assert out["cons"][0] == ""


def test_reads_with_long_indels_keep_their_regions(oracle, pkg, gpu_ctx):
    """K3's twin of tests/test_gpu_seeded.py::test_reads_with_long_indels_are_mapped_across_them.  minimap2 chains a template across a 40 - 100 base deletion or insertion in
    the read (bw 500, max_gap 10000), so find_base_type_in_sequence (src/cyp2d6/haplotyper.rs:193-249) still reports the gene's region; the library's 64-diagonal cell is lost
    there, for EVERY template -- a hole in the read that strongly anchored templates expect to cover --, and the templates lost over the hole run once more on 256 diagonals
    (sp_cyp.hip cyp_find_regions).  Held to: oracle/cyp.c (the same rule, bit for bit) and the reference-call-pattern port on oracle/mm2.c (template, start, end, nm, unmapped
    of every hit); the hit of the gene spans the indel."""
    import cpu_port_cyp as cpc
    from pb_starphase_amd import synth
    cfg, gene_def = cr.load_db()
    locus = synth.Chr22Locus(cfg, gene_def, seed=3)
    db = pkg.ffi.CypDb(gpu_ctx, cfg, gene_def, locus.sequence, locus.start)
    tm = db.templates()
    tseqs, ttype = [t[3] for t in tm], np.array([t[0] for t in tm], np.int32)
    T = gpu_ctx.upload(tseqs)
    rng = np.random.default_rng(33)
    name, haps, _exp = cr.scenarios(locus)[0]                               # *1/*2
    base_reads = [r for r in locus.sample(np.random.default_rng(7), haps, 400, lo=7000, hi=9000)]
    hits0 = gpu_ctx.cyp_find_regions(T, ttype, gpu_ctx.upload(base_reads), 0.5)
    reads, kinds, where = [], [], []
    for r, read in enumerate(base_reads):
        mine = [h for h in hits0 if int(h["read"]) == r and int(h["end"]) - int(h["start"]) >= 4000]
        if not mine or len(reads) >= 48:
            continue
        h = mine[0]
        k = int(rng.integers(int(h["start"]) + 1200, int(h["end"]) - 1200))
        size = (40, 80, 100)[len(reads) % 3]
        if len(reads) % 2:
            reads.append(read[:k] + read[k + size:]); kinds.append(-size)
        else:
            reads.append(read[:k] + "".join(rng.choice(list("ACGT"), size)) + read[k:]); kinds.append(size)
        where.append(k)
    assert len(reads) >= 24
    R = gpu_ctx.upload(reads)
    gpu_ctx.profile_reset()
    hits = gpu_ctx.cyp_find_regions(T, ttype, R, 0.5)
    assert gpu_ctx.profile_get("k3_retried_pairs")[2] > 0
    pdb, _ccfg = cpc.tables(cfg, gene_def, locus)
    al = cpc.Mm2Aligner(oracle)
    n_span = same_as_port = 0
    for r, read in enumerate(reads):
        got = [tuple(int(h[k]) for k in ("template_idx", "start", "end", "nm", "unmapped")) for h in hits if int(h["read"]) == r]
        exp = of.oracle_find_base_type(oracle, read, tseqs, ttype, 0.5)
        assert got == [tuple(int(e[k]) for k in ("template_idx", "start", "end", "nm", "unmapped")) for e in exp], (r, kinds[r], got, exp)
        # a hit spans the indel: its edits count the indel's bases
        spanning = [g for g in got if g[1] < where[r] - 500 and g[2] > where[r] + 500]
        n_span += bool(spanning) and all(g[3] >= abs(kinds[r]) for g in spanning)
        port = [tuple(int(p[k]) for k in ("template_idx", "start", "end", "nm", "unmapped")) for p in al.find_base_type(oracle, read, pdb, 0.5)]
        same_as_port += got == port
    print("K3 reads with a 40 - 100 base indel: a hit spans the indel on", n_span, "of", len(reads), "; the whole hit list equals the port's on", same_as_port)
    assert n_span == len(reads)
    assert same_as_port == len(reads)
