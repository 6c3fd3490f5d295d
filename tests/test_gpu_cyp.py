"""GPU parity for K5 (sp_cyp_best_chain_pair) against oracle/cyp.c: identical chains, indices and bit-identical f64 scores."""
import numpy as np
import pytest

import cyp_cases
import oracle_ffi as of

pytestmark = pytest.mark.gpu


def gpu_chain_pair(ctx, inp):
    n_reads = max(len(inp.chain_names), len(inp.score_names))
    rco = inp.read_chain_off if len(inp.read_chain_off) == n_reads + 1 else np.zeros(n_reads + 1, np.int32)
    rwo = inp.read_w_off if len(inp.read_w_off) == n_reads + 1 else np.zeros(n_reads + 1, np.int32)
    return ctx.cyp_best_chain_pair(inp.types, inp.subtypes, inp.cfg["translate"], inp.cfg["connections"], inp.cfg["singletons"],
                                   rco, inp.chain_off, inp.chain_items, rwo, inp.w_ed, inp.w_ov,
                                   inp.infer, inp.normalize_all, inp.ignore, inp.penalties)


def same(oracle, gpu_ctx, inp):
    exp = of.oracle_chain_pair(oracle, inp)
    rc, got = gpu_chain_pair(gpu_ctx, inp)
    assert rc == exp.status, (rc, exp.status)
    if rc == 0:
        assert got.n_possible == exp.n_possible
        assert (got.index1, got.index2) == (exp.index1, exp.index2)
        assert list(got.chain1[:got.n1]) == list(exp.chain1[:exp.n1]) and list(got.chain2[:got.n2]) == list(exp.chain2[:exp.n2])
        for f in ("score", "ln_ed_penalty", "mn_llh_penalty", "allele_expected_penalty", "unexpected_chain_penalty", "inferred_chain_penalty"):
            assert getattr(got, f) == getattr(exp, f), (f, getattr(got, f), getattr(exp, f))       # bit-identical f64 (<= 1e-5 required)
        assert got.edit_distance == exp.edit_distance
    return rc, exp


@pytest.mark.parametrize("case", cyp_cases.reference_cases(), ids=lambda c: c[0])
def test_reference_cases(oracle, gpu_ctx, case):
    """the reference's own scenarios (src/cyp2d6/chaining.rs:950-1195) through the C ABI"""
    name, inp, status, chains, dang = case
    rc, exp = same(oracle, gpu_ctx, inp)
    assert rc == status
    if status == 0:
        assert [list(exp.chain1[:exp.n1]), list(exp.chain2[:exp.n2])] == chains


def test_synthetic_loci(oracle, gpu_ctx):
    rng = np.random.default_rng(17)
    n_ok = 0
    for k in range(10):
        labels, obs, sc, infer = cyp_cases.synthetic_problem(rng, n_d6=3 + k % 3, n_reads=40 + 20 * k, infer=bool(k % 2))
        for normalize_all in (True, False):
            inp = of.ChainInputs(labels, obs, sc, infer, normalize_all, of.DEFAULT_PENALTIES, False)
            rc, exp = same(oracle, gpu_ctx, inp)
            n_ok += rc == 0
    assert n_ok >= 10


def test_chain_pairs_by_workgroup_and_by_thread(oracle, gpu_ctx):
    """K5's two kernels (one workgroup per pair for few pairs, one thread per pair for many) on the same problems: both equal the oracle bit for bit;
    reads with many equally good windows take the workgroup kernel's from-memory path (more than eight addends)"""
    rng = np.random.default_rng(23)
    seen = {0: 0, 1 << 20: 0}
    try:
        for k in range(8):
            labels, obs, sc, infer = cyp_cases.synthetic_problem(rng, n_d6=3 + k % 4, n_reads=300 + 150 * k, noise=0.0 if k % 2 else 0.15, infer=bool(k % 2))
            inp = of.ChainInputs(labels, obs, sc, infer, True, of.DEFAULT_PENALTIES, False)
            for limit in seen:
                gpu_ctx.set_option("k5_block_pairs", limit)
                rc, exp = same(oracle, gpu_ctx, inp)
                seen[limit] += rc == 0
    finally:
        gpu_ctx.set_option("k5_block_pairs", 4096)
    assert min(seen.values()) >= 4
    with pytest.raises(Exception):
        gpu_ctx.set_option("k5_block_pairs", -1)


def test_topk_anchors(oracle, pkg, gpu_ctx):
    """a template that occurs twice in a read must give two placements (D6 and its D7 paralog / duplications)"""
    import ctypes as C
    from pb_starphase_amd import synth
    locus = synth.CypLocus(seed=4)
    rng = np.random.default_rng(1)
    reads = [locus.haplotype("dup"), locus.haplotype("normal")[2000:20000], "ACGT" * 10]
    T, R = gpu_ctx.upload(locus.templates), gpu_ctx.upload(reads)
    a = np.repeat(np.arange(len(locus.templates)), len(reads)).astype(np.uint32)
    b = np.tile(np.arange(len(reads)), len(locus.templates)).astype(np.uint32)
    diag, votes = gpu_ctx.anchor_batch_topk(T, R, a, b, 4)
    for x, (ti, ri) in enumerate(zip(a, b)):
        A, B = oracle.encode(locus.templates[ti]), oracle.encode(reads[ri])
        d = (C.c_int32 * 4)()
        v = (C.c_int32 * 4)()
        oracle.L.osp_anchor_topk(A.ctypes.data_as(C.c_void_p), len(A), B.ctypes.data_as(C.c_void_p), len(B), 4, d, v)
        assert votes[x].tolist() == list(v), (ti, ri, votes[x], list(v))
        assert [int(dd) for dd, vv in zip(diag[x], votes[x]) if vv > 0] == [int(dd) for dd, vv in zip(d, v) if vv > 0]
    d6 = locus.template_names.index("CYP2D6")
    assert (votes[d6 * len(reads) + 0] > 400).sum() >= 3       # two D6 copies + the D7 paralog in the duplicated haplotype


def test_find_regions_and_weights(oracle, pkg, gpu_ctx):
    """K3 / K4 against the oracle restatements of find_base_type_in_sequence / weight_sequence"""
    from pb_starphase_amd import synth
    locus = synth.CypLocus(seed=5)
    rng = np.random.default_rng(2)
    reads = locus.reads(rng, 6, "normal") + locus.reads(rng, 3, "deletion") + locus.reads(rng, 3, "dup", mean_len=16000)
    reads.append("".join(rng.choice(list("ACGT"), 4000)))
    T, R = gpu_ctx.upload(locus.templates), gpu_ctx.upload(reads)
    for max_missing in (1.0, 0.5):
        hits = gpu_ctx.cyp_find_regions(T, locus.template_types, R, max_missing)
        n_total = 0
        for r, read in enumerate(reads):
            exp = of.oracle_find_base_type(oracle, read, locus.templates, locus.template_types, max_missing)
            got = hits[hits["read"] == r]
            assert len(got) == len(exp), (r, got, exp)
            for g, e in zip(got, exp):
                assert tuple(int(g[k]) for k in exp.dtype.names) == tuple(int(x) for x in e.tolist()), (r, g, e)
            n_total += len(exp)
        assert n_total >= 20
    # K4: segments cut out of the reads at the K3 hits, weighted against "consensuses" (here: the templates + a noisy copy)
    hits = gpu_ctx.cyp_find_regions(T, locus.template_types, R, 1.0)
    segs = [reads[h["read"]][h["start"]:h["end"]] for h in hits][:40] + ["".join(rng.choice(list("ACGT"), 1500))]
    cons = locus.templates + [synth.mutate(rng, locus.d6, 5, 2, 2), "N" * 50]
    allowed = np.array([1] * len(locus.templates) + [1, 0], np.uint8)
    Cs, Ss = gpu_ctx.upload(cons), gpu_ctx.upload(segs)
    ed, ov, kept = gpu_ctx.cyp_weight_segments(Cs, allowed, Ss)
    for s, seg in enumerate(segs):
        e_ed, e_ov, e_kept = of.oracle_weight_sequence(oracle, seg, cons, allowed)
        assert ed[s].tolist() == e_ed.tolist() and ov[s].tolist() == e_ov.tolist() and kept[s] == e_kept, (s, ed[s], e_ed)
    assert kept[:-1].sum() >= len(segs) - 3 and kept[-1] == 0


def test_score_alleles(oracle, pkg, gpu_ctx):
    """K7 vs the oracle restatement of the assign_haplotype scoring loop, on the bundled CYP2D6 definitions
    (520 star alleles; the variant table is rebuilt from cyp2d6_gene_def of the fixture database)"""
    import ctypes as C, gzip, json, os
    db = json.load(gzip.open(os.path.join(os.path.dirname(__file__), "golden", "cyp2d6_db_v0.14.1.json.gz")))["cyp2d6_gene_def"]
    names = sorted(db)
    variants = sorted({(v["position"], v["reference"], v["alternate"]) for a in db.values() for v in a["variants"]})
    vidx = {v: i for i, v in enumerate(variants)}
    is_vi = np.zeros(len(variants), np.uint8)
    hap = np.zeros((len(names), len(variants)), np.uint8)
    for ai, n in enumerate(names):
        for v in db[n]["variants"]:
            i = vidx[(v["position"], v["reference"], v["alternate"])]
            hap[ai, i] = 1
            if v.get("extras", {}).get("VI") is not None:
                is_vi[i] = 1
    assert hap.shape[0] >= 500 and hap.shape[1] >= 350 and is_vi.sum() >= 100
    rng = np.random.default_rng(8)
    states = []
    for k in range(12):
        st = hap[int(rng.integers(len(names)))].copy()
        flip = rng.random(len(variants))
        st[flip < 0.01] ^= 1
        st[(flip >= 0.01) & (flip < 0.03)] = 2
        st[(flip >= 0.03) & (flip < 0.08)] = 3
        states.append(st)
    states.append(np.full(len(variants), 3, np.uint8))            # nothing set: only Unknown stays at (0, 0)
    states = np.array(states, np.uint8)
    bv, ba, tie = gpu_ctx.cyp_score_alleles(hap, is_vi, states)
    for s in range(len(states)):
        ev, ea = C.c_uint32(), C.c_uint32()
        et = np.zeros(len(names), np.uint8)
        oracle.L.osp_cyp_score_alleles(len(variants), len(names), hap.ctypes.data_as(C.c_void_p), is_vi.ctypes.data_as(C.c_void_p),
                                       states[s].ctypes.data_as(C.c_void_p), C.byref(ev), C.byref(ea), et.ctypes.data_as(C.c_void_p))
        assert (bv[s], ba[s]) == (ev.value, ea.value)
        assert tie[s].tolist() == et.tolist()
    assert bv[-1] == 0 and ba[-1] == 0


def make_variant_panel(rng, backbone, n=48):
    """non-overlapping variants on the backbone: SNVs, small insertions / deletions (anchored), one longer deletion, one two-alt site"""
    pos = sorted(rng.choice(np.arange(150, len(backbone) - 150, 40), n, replace=False).tolist())
    out = []
    for k, p in enumerate(pos):
        ref = backbone[p]
        kind = k % 6
        if kind in (0, 1, 2):
            out.append((p, ref, rng.choice([c for c in "ACGT" if c != ref])))
        elif kind == 3:
            out.append((p, ref, ref + "".join(rng.choice(list("ACGT"), int(rng.integers(1, 4))))))
        elif kind == 4:
            out.append((p, backbone[p:p + int(rng.integers(2, 5))], ref))
        else:
            out.append((p, backbone[p:p + 10], ref))
    p0 = out[0][0]
    out.append((p0, out[0][1], next(c for c in "ACGT" if c not in (out[0][1], out[0][2]))))      # second alternate at the first site
    return out


def apply_variants(backbone, variants, chosen):
    s, shift = backbone, 0
    for i in sorted(chosen, key=lambda i: variants[i][0]):
        p, ref, alt = variants[i]
        assert s[p + shift:p + shift + len(ref)] == ref
        s = s[:p + shift] + alt + s[p + shift + len(ref):]
        shift += len(alt) - len(ref)
    return s


def test_variant_states(oracle, pkg, gpu_ctx):
    """K9 against the oracle, and against the alleles the sequences were built from"""
    from pb_starphase_amd import synth
    rng = np.random.default_rng(44)
    backbone = "".join(rng.choice(list("ACGT"), 6200))
    variants = make_variant_panel(rng, backbone)
    nv = len(variants)
    pos, refs, alts = [v[0] for v in variants], [v[1] for v in variants], [v[2] for v in variants]
    seqs, truth = [], []
    for k in range(8):
        chosen = sorted(rng.choice(nv - 1, int(rng.integers(0, 14)), replace=False).tolist())
        full = apply_variants(backbone, variants, chosen)
        lo, hi = (0, len(full)) if k % 3 == 0 else (int(rng.integers(100, 900)), len(full) - int(rng.integers(100, 900)))
        s = full[lo:hi]
        if k % 2:
            s = synth.hifi_errors(rng, s)
        seqs.append(s)
        truth.append(set(chosen))
    seqs.append("".join(rng.choice(list("ACGT"), 2000)))                                           # does not align: all states 3
    states, alns = gpu_ctx.cyp_variant_states(gpu_ctx.upload(seqs), backbone, pos, refs, alts)
    n_called = 0
    for i, seq in enumerate(seqs):
        e_states, e_aln = of.oracle_variant_states(oracle, seq, backbone, pos, refs, alts)
        assert states[i].tolist() == e_states.tolist(), (i, np.flatnonzero(states[i] != e_states))
        if e_aln is None:
            assert not alns[i]["ok"] and (states[i] == 3).all()
            continue
        assert (int(alns[i]["a_start"]), int(alns[i]["a_end"]), int(alns[i]["b_start"]), int(alns[i]["b_end"]), int(alns[i]["nm"])) == e_aln
        if i < len(truth) and i % 2 == 0:                                                         # noise-free sequences: the states are the truth
            for v in range(nv - 1):
                if states[i][v] == 3:
                    continue
                n_called += 1
                assert states[i][v] == (1 if v in truth[i] else 0), (i, v, variants[v])
    assert n_called > 100
    assert (states[-1] == 3).all()


def test_weight_sequence_reference_vectors(oracle, pkg, gpu_ctx):
    """test_weight_sequence (src/cyp2d6/chaining.rs:1051-1080) through K4"""
    cons = gpu_ctx.upload(cyp_cases.WEIGHT_SEQUENCE_CONSENSUS)
    segs = gpu_ctx.upload(cyp_cases.WEIGHT_SEQUENCE_SEGMENTS)
    ed, ov, kept = gpu_ctx.cyp_weight_segments(cons, np.ones(3, np.uint8), segs)
    s0 = list(zip(ed[0].tolist(), ov[0].tolist()))
    s1 = list(zip(ed[1].tolist(), ov[1].tolist()))
    assert kept.tolist() == [1, 1]
    assert min(s0) == s0[0] and s0[0] < s0[1] and s0[0] < s0[2]
    assert s1[0] == s1[1] == s1[2]
    for k, seg in enumerate(cyp_cases.WEIGHT_SEQUENCE_SEGMENTS):
        e_ed, e_ov, e_kept = of.oracle_weight_sequence(oracle, seg, cyp_cases.WEIGHT_SEQUENCE_CONSENSUS, np.ones(3, np.uint8))
        assert ed[k].tolist() == e_ed.tolist() and ov[k].tolist() == e_ov.tolist() and kept[k] == e_kept


def test_find_regions_at_four_to_five_percent_divergence(oracle, pkg, gpu_ctx):
    """max_ed_frac = 0.05 (src/cyp2d6/haplotyper.rs:160,228-232): a 6.2 kb template hit with 4.5 % edits (280 > the old 255-edit cap) is
    kept, one with 5.6 % is dropped; library == oracle either way."""
    from pb_starphase_amd import synth
    rng = np.random.default_rng(12)
    template = "".join(rng.choice(list("ACGT"), 6200))
    flank = lambda n: "".join(rng.choice(list("ACGT"), n))
    reads, want = [], []
    for frac, kept in ((0.030, True), (0.045, True), (0.048, True), (0.056, False), (0.070, False)):
        n_ed = int(frac * len(template))
        body = synth.mutate(rng, template, n_sub=n_ed - 20, n_ins=10, n_del=10)
        reads.append(flank(700) + body + flank(900)); want.append(kept)
    T, R = gpu_ctx.upload([template]), gpu_ctx.upload(reads)
    types = np.array([2], np.int32)                                              # CYP2D6: unmapped bases are not penalised in the filter
    hits = gpu_ctx.cyp_find_regions(T, types, R, 1.0)
    for r, read in enumerate(reads):
        exp = of.oracle_find_base_type(oracle, read, [template], types, 1.0)
        got = hits[hits["read"] == r]
        assert len(got) == len(exp) == (1 if want[r] else 0), (r, len(got), len(exp))
        for g, e in zip(got, exp):
            assert tuple(int(g[k]) for k in exp.dtype.names) == tuple(int(x) for x in e.tolist())
    nm = [int(h["nm"]) for h in hits]
    assert max(nm) > 255 and all(n <= 0.05 * 6200 + 1 for n in nm), nm


def test_variant_states_on_clustered_variants(oracle, pkg, gpu_ctx):
    """K9 on the graphs of tests/test_oracle_cyp.py::test_variant_states_against_joint_enumeration (overlapping deletion + SNV, adjacent
    insertion, a third base at a SNV): GPU == oracle, whose states that test proves equal to brute-force joint enumeration."""
    rng = np.random.default_rng(9)
    n_two = n = 0
    for rep in range(8):
        backbone = "".join(rng.choice(list("ACGT"), 900))
        p = int(rng.integers(200, 500))
        other = lambda c: "ACGT"[("ACGT".index(c) + 1 + int(rng.integers(0, 3))) % 4]
        variants = sorted([(p, backbone[p], other(backbone[p])), (p, backbone[p:p + 3], backbone[p]),
                           (p + 3, backbone[p + 3], backbone[p + 3] + "".join(rng.choice(list("ACGT"), 2))), (p + 5, backbone[p + 5], other(backbone[p + 5])),
                           (p + 60, backbone[p + 60], other(backbone[p + 60])), (p + 80, backbone[p + 80:p + 84], backbone[p + 80]),
                           (p + 200, backbone[p + 200:p + 202], backbone[p + 200]), (p + 201, backbone[p + 201], other(backbone[p + 201]))], key=lambda v: v[0])
        pos, refs, alts = [v[0] for v in variants], [v[1] for v in variants], [v[2] for v in variants]
        seqs = []
        for k in range(6):
            chosen, end = [], -1
            for v in range(len(variants)):                                        # a random compatible subset, left to right
                if variants[v][0] >= end and rng.random() < 0.5:
                    chosen.append(v); end = variants[v][0] + len(variants[v][1])
            s = apply_variants(backbone, variants, chosen)
            if k % 3 == 1:
                q = p + int(rng.integers(0, 8)); s = s[:q] + other(s[q]) + s[q + 1:]
            if k % 3 == 2:
                v = next(x for x in range(len(variants)) if variants[x][0] == p + 60)
                third = next(c for c in "ACGT" if c not in (variants[v][1], variants[v][2]))
                s = backbone[:p + 60] + third + backbone[p + 61:]
            seqs.append(s)
        states, _alns = gpu_ctx.cyp_variant_states(gpu_ctx.upload(seqs), backbone, pos, refs, alts)
        for i, s in enumerate(seqs):
            e_states, _ = of.oracle_variant_states(oracle, s, backbone, pos, refs, alts)
            assert states[i].tolist() == e_states.tolist(), (rep, i, states[i].tolist(), e_states.tolist())
            n += len(pos); n_two += int((e_states == 2).sum())
    assert n >= 300 and n_two >= 4
