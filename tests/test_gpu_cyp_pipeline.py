"""BASELINE config 3 (CYP2D6, targeted-style reads) end to end through the C ABI:
K3 sp_cyp_find_regions -> K4 sp_cyp_weight_segments -> sp_cyp_build_chains -> K5 sp_cyp_best_chain_pair -> sp_cyp_chain_to_hap,
against the same pipeline assembled from the oracle's restatements (src/cyp2d6/caller.rs:126-139,429-583,634-640,907-957).
The consensus step between K3 and K4 (waffle_con) is not part of this round: the consensus set is the
sample's true region sequences, labelled as the reference's typing step would label them."""
import ctypes as C

import numpy as np
import pytest

import oracle_ffi as of

pytestmark = pytest.mark.gpu

MAX_MISSING_CHAIN_FRAC = 0.5          # src/cyp2d6/caller.rs:114-119


def make_sample(locus, synth, rng, scenario, n_reads):
    d6_1 = locus.d6
    d6_2 = synth.mutate(rng, locus.d6, 10, 1, 1)
    normal = locus.haplotype("normal")
    dup = locus.haplotype("dup")
    haps = {"*1/*2": [normal, normal.replace(d6_1, d6_2)],
            "*5/*1": [locus.haplotype("deletion"), normal],
            "*2x2/*1": [dup.replace(d6_1, d6_2), normal]}[scenario]
    reads = []
    total = sum(len(h) for h in haps)
    for hap in haps:
        for _ in range(int(round(n_reads * len(hap) / total))):      # equal coverage of the two haplotypes
            ln = int(min(len(hap), max(4000, rng.normal(12000, 3000))))
            s = int(rng.integers(0, len(hap) - ln + 1))
            reads.append(synth.hifi_errors(rng, hap[s:s + ln]))
    labels = [("REP6", None), ("CYP2D6", "1"), ("CYP2D6", "2"), ("link_region", None), ("REP7", None), ("spacer", None), ("CYP2D7", None)]
    cons = [locus.rep6, d6_1, d6_2, locus.link, locus.rep7, locus.spacer, locus.d7]
    if "*5" in scenario:                               # a *5 consensus only exists in a sample that carries the deletion
        labels.append(("CYP2D6*5", None))
        cons.append(locus.templates[locus.template_names.index("CYP2D6*5")])
    return reads, labels, cons


def k5_inputs(labels, false_allele, built, ed, ov):
    labels = [("FalseAllele", s) if false_allele[h] else (t, s) for h, (t, s) in enumerate(labels)]
    obs = {f"r{r:06d}": built["chains"][k] for k, r in enumerate(built["read_index"])}
    scores = {f"r{r:06d}": [[(int(ed[sg][c]), float(ov[sg][c])) for c in range(len(labels))] for sg in built["w_rows"][k]]
              for k, r in enumerate(built["read_index"])}
    return labels, of.ChainInputs(labels, obs, scores, False, True, of.DEFAULT_PENALTIES, False)


def product_pipeline(pkg, ctx, locus, reads, labels, cons):
    T, R = ctx.upload(locus.templates), ctx.upload(reads)
    hits = ctx.cyp_find_regions(T, locus.template_types, R, MAX_MISSING_CHAIN_FRAC)
    segs = [reads[h["read"]][h["start"]:h["end"]] for h in hits]
    seg_off = np.zeros(len(reads) + 1, np.uint32)
    np.add.at(seg_off, hits["read"] + 1, 1)
    seg_off = np.cumsum(seg_off).astype(np.uint32)
    allowed = np.ones(len(cons), np.uint8)
    ed, ov, kept = ctx.cyp_weight_segments(ctx.upload(cons), allowed, ctx.upload(segs))
    types = np.array([of.REGION_TYPES[t] for t, _ in labels], np.int32)
    built = pkg.ffi.build_chains(types, seg_off, ed.reshape(-1), kept)
    labels2, inp = k5_inputs(labels, built["false_allele"], built, ed, ov)
    from test_gpu_cyp import gpu_chain_pair
    rc, res = gpu_chain_pair(ctx, inp)
    return hits, (ed, ov, kept), built, labels2, inp, rc, res


def oracle_pipeline(oracle, locus, reads, labels, cons):
    hits, seg_off, segs = [], [0], []
    for r, read in enumerate(reads):
        h = of.oracle_find_base_type(oracle, read, locus.templates, locus.template_types, MAX_MISSING_CHAIN_FRAC)
        hits.append(h)
        segs += [read[int(x["start"]):int(x["end"])] for x in h]
        seg_off.append(len(segs))
    allowed = np.ones(len(cons), np.uint8)
    ed, ov, kept = [], [], []
    for s in segs:
        e, o, k = of.oracle_weight_sequence(oracle, s, cons, allowed)
        ed.append(e); ov.append(o); kept.append(k)
    ed, ov, kept = np.array(ed, np.uint64), np.array(ov, np.float64), np.array(kept, np.uint8)
    types = np.array([of.REGION_TYPES[t] for t, _ in labels], np.int32)
    built = of.oracle_build_chains(oracle, types, np.array(seg_off, np.uint32), ed.reshape(-1), kept)
    labels2, inp = k5_inputs(labels, built["false_allele"], built, ed, ov)
    return hits, (ed, ov, kept), built, labels2, inp, of.oracle_chain_pair(oracle, inp)


def product_hap(pkg, chain, labels, cfg):
    types = np.array([of.REGION_TYPES[t] for t, _ in labels], np.int32)
    st = of._strs([s for _, s in labels])
    k, v = of._strs([a for a, _ in cfg["translate"]]), of._strs([b for _, b in cfg["translate"]])
    ch = np.array(chain, np.int32)
    out = C.create_string_buffer(512)
    pkg.ffi.lib().sp_cyp_chain_to_hap(ch.ctypes.data, len(ch), types.ctypes.data, st, len(cfg["translate"]), k, v, 0, out, 512)
    return out.value.decode()


@pytest.mark.parametrize("scenario,expected", [("*1/*2", {"*1", "*2"}), ("*5/*1", {"*5", "*1"}), ("*2x2/*1", {"*2x2", "*1"})])
def test_config3_pipeline(oracle, pkg, gpu_ctx, scenario, expected):
    from pb_starphase_amd import synth
    locus = synth.CypLocus(seed=11)
    rng = np.random.default_rng(hash(scenario) % 1000 if False else {"*1/*2": 1, "*5/*1": 2, "*2x2/*1": 3}[scenario])
    reads, labels, cons = make_sample(locus, synth, rng, scenario, 160)
    g_hits, (g_ed, g_ov, g_kept), g_built, g_labels, g_inp, g_rc, g_res = product_pipeline(pkg, gpu_ctx, locus, reads, labels, cons)
    o_hits, (o_ed, o_ov, o_kept), o_built, o_labels, o_inp, o_res = oracle_pipeline(oracle, locus, reads, labels, cons)
    # K3: identical regions, read by read
    for r in range(len(reads)):
        got = g_hits[g_hits["read"] == r]
        assert len(got) == len(o_hits[r])
        for g, e in zip(got, o_hits[r]):
            assert tuple(int(g[k]) for k in e.dtype.names) == tuple(int(x) for x in e.tolist())
    # K4: identical weights
    assert g_ed.tolist() == o_ed.tolist() and g_ov.tolist() == o_ov.tolist() and g_kept.tolist() == o_kept.tolist()
    # chain building: identical chain sets, counts and FalseAllele flags
    assert g_built["read_index"] == o_built["read_index"] and g_built["chains"] == o_built["chains"] and g_built["w_rows"] == o_built["w_rows"]
    assert g_built["unique_counts"].tolist() == o_built["unique_counts"].tolist() and g_labels == o_labels
    # K5: identical pair, bit-identical score
    assert g_rc == o_res.status == 0
    assert list(g_res.chain1[:g_res.n1]) == list(o_res.chain1[:o_res.n1]) and list(g_res.chain2[:g_res.n2]) == list(o_res.chain2[:o_res.n2])
    assert g_res.score == o_res.score and g_res.edit_distance == o_res.edit_distance
    # the call: same strings from the library and the oracle, and the simulated truth
    cfg = of.default_cyp_config()
    got = {product_hap(pkg, list(g_res.chain1[:g_res.n1]), g_labels, cfg), product_hap(pkg, list(g_res.chain2[:g_res.n2]), g_labels, cfg)}
    exp = {of.chain_hap_string(oracle, list(o_res.chain1[:o_res.n1]), o_labels, 0, cfg), of.chain_hap_string(oracle, list(o_res.chain2[:o_res.n2]), o_labels, 0, cfg)}
    assert got == exp == expected, (got, exp)


@pytest.mark.parametrize("scenario,expected", [("*1/*4", {"*1", "*4"}), ("*5/*2", {"*5", "*2"}), ("*4x2/*1", {"*4x2", "*1"}), ("*2/*10", {"*2", "*10"})])
def test_reads_to_diplotype(oracle, pkg, gpu_ctx, scenario, expected):
    """sp_cyp_diplotype (diplotype_cyp2d6, src/cyp2d6/caller.rs:39-741) from raw reads: K3 -> K8 multi-way consensus -> merge -> K9 + K7 typing
    -> K4 -> chains -> K5 -> strings, against the same pipeline assembled from the oracle (tests/cyp_pipeline.py) and against the truth"""
    import cyp_fixture as cf
    import cyp_pipeline as cp
    from pb_starphase_amd import synth
    locus = synth.CypLocus(seed=11)
    db, d6 = cf.make_db(locus, synth, np.random.default_rng(5))
    reads = cf.sample(locus, synth, np.random.default_rng(7), d6, scenario, 120)
    exp = cp.diplotype(oracle, db, reads)
    cfg = of.default_cyp_config()
    call, cons, labels = gpu_ctx.cyp_diplotype(gpu_ctx.upload(db.seqs), db.types, db.subtypes, db.deep, db.backbone, db.variants, db.is_vi,
                                                db.allele_subtypes, db.hap_matrix, cfg, gpu_ctx.upload(reads))
    assert call.status == exp["status"] == 0
    assert cons == exp["consensus"]
    assert labels == [(int(t), s) for t, s in exp["labels"]]
    assert list(call.chain1[:call.n1]) == exp["chain1"] and list(call.chain2[:call.n2]) == exp["chain2"]
    assert call.score == exp["score"]
    got = (call.hap1.decode(), call.hap2.decode(), call.core1.decode(), call.core2.decode())
    assert got == (exp["hap1"], exp["hap2"], exp["core1"], exp["core2"])
    assert {got[0], got[1]} == expected
    # no reads at all
    call, cons, labels = gpu_ctx.cyp_diplotype(gpu_ctx.upload(db.seqs), db.types, db.subtypes, db.deep, db.backbone, db.variants, db.is_vi,
                                                db.allele_subtypes, db.hap_matrix, cfg, gpu_ctx.upload(["ACGT" * 500]))
    assert call.status == 1 and call.n_consensus == 0
