"""Pins oracle/cyp.c to the reference's CYP2D6 chaining tests (src/cyp2d6/chaining.rs:950-1195, caller.rs:972-1006)."""
import numpy as np
import pytest

import cyp_cases
import oracle_ffi as of


@pytest.mark.parametrize("case", cyp_cases.reference_cases(), ids=lambda c: c[0])
def test_find_best_chain_pair_reference_cases(oracle, case):
    name, inp, status, chains, dang = case
    res = of.oracle_chain_pair(oracle, inp)
    assert res.status == status, name
    if status == 0:
        got = [list(res.chain1[:res.n1]), list(res.chain2[:res.n2])]
        assert got == chains, (name, got)
        assert cyp_cases.danglers(inp, res) == dang


def test_convert_chain_to_hap(oracle):
    """src/cyp2d6/caller.rs:972-1006 test_convert_chain_to_hap"""
    labels = [("CYP2D7", None), ("CYP2D6", "1.001"), ("CYP2D6", "10"), ("CYP2D6", "1.002"), ("CYP2D6", "1.002")]
    assert of.chain_hap_string(oracle, [2, 2, 1, 0], labels, 1) == "*1.001 + *10x2"
    assert of.chain_hap_string(oracle, [3, 1, 0], labels, 1) == "*1.001 + *1.002"
    assert of.chain_hap_string(oracle, [3, 1, 0], labels, 0) == "*1x2"
    assert of.chain_hap_string(oracle, [3, 4], labels, 1) == "*1.002x2"
    # *5 is dropped when anything else is on the chain; hybrids translate through cyp_translate
    labels2 = [("CYP2D6*5", None), ("CYP2D6", "4.013"), ("Hybrid", "CYP2D6::CYP2D7::exon2"), ("REP6", None)]
    assert of.chain_hap_string(oracle, [3, 0], labels2, 1) == "*5"
    assert of.chain_hap_string(oracle, [3, 2, 1, 0], labels2, 0) == "*4 + *68"


def test_label_grammar(oracle):
    """is_allowed_label_pair (src/cyp2d6/region_label.rs:178-222): the canonical chain is legal, shortcuts are not"""
    T = of.REGION_TYPES
    ok = oracle.L.osp_cyp_is_allowed_label_pair
    canon = ["REP6", "CYP2D6", "link_region", "REP7", "spacer", "CYP2D7"]
    for a, b in zip(canon, canon[1:]):
        assert ok(T[a], T[b])
    assert ok(T["REP7"], T["Hybrid"]) and ok(T["CYP2D6*5"], T["spacer"]) and ok(T["REP6"], T["CYP2D6*5"])
    for a, b in (("CYP2D6", "CYP2D6"), ("CYP2D6", "REP7"), ("link_region", "CYP2D6"), ("CYP2D7", "link_region"),
                 ("CYP2D6*5", "CYP2D6*5"), ("spacer", "REP7"), ("CYP2D6", "REP6"), ("REP6", "CYP2D7")):
        assert not ok(T[a], T[b]), (a, b)


def test_synthetic_problems_are_solvable(oracle):
    rng = np.random.default_rng(5)
    n_ok = 0
    for _ in range(6):
        labels, obs, sc, infer = cyp_cases.synthetic_problem(rng)
        inp = of.ChainInputs(labels, obs, sc, infer, True, of.DEFAULT_PENALTIES, False)
        res = of.oracle_chain_pair(oracle, inp)
        assert res.status in (0, 17, 18)
        n_ok += res.status == 0
        if res.status == 0:
            assert res.n_possible >= 2 and res.score >= 0.0
    assert n_ok >= 4


def test_weight_sequence_reference_vectors(oracle):
    """src/cyp2d6/chaining.rs:1051-1080: one mismatch ranks strictly worse, an unknown base makes the three consensuses equal"""
    import numpy as np
    import oracle_ffi as of
    assert len(cyp_cases.WEIGHT_SEQUENCE_CONSENSUS[0]) == 218
    allowed = np.ones(3, np.uint8)
    ed, ov, kept = of.oracle_weight_sequence(oracle, cyp_cases.WEIGHT_SEQUENCE_SEGMENTS[0], cyp_cases.WEIGHT_SEQUENCE_CONSENSUS, allowed)
    score = list(zip(ed.tolist(), ov.tolist()))
    assert kept and min(score) == score[0] and score[0] < score[1] and score[0] < score[2]
    ed, ov, kept = of.oracle_weight_sequence(oracle, cyp_cases.WEIGHT_SEQUENCE_SEGMENTS[1], cyp_cases.WEIGHT_SEQUENCE_CONSENSUS, allowed)
    score = list(zip(ed.tolist(), ov.tolist()))
    assert kept and score[0] == score[1] == score[2]


def test_overlap_score(oracle):
    """src/cyp2d6/haplotyper.rs:936-941"""
    import ctypes as C
    f = oracle.L.osp_cyp_overlap_score
    f.restype = C.c_double
    assert f(0, 1, 1, 2) == 0.0 and f(0, 10, 1, 5) == 1.0 and f(0, 10, 5, 100) == 0.5 and f(15, 100, 0, 20) == 0.25


# ------------------------------------------------------------------ K9: the graph statement against brute-force joint enumeration
def _edit_distance(a, b):
    """plain global unit-cost edit distance (row by row; the in-row dependency is a running minimum)"""
    a = np.frombuffer(a.encode(), np.uint8); b = np.frombuffer(b.encode(), np.uint8)
    idx = np.arange(len(b) + 1)
    row = idx.copy()
    for i in range(1, len(a) + 1):
        diag = row[:-1] + (b != a[i - 1])
        new = np.empty_like(row)
        new[0] = i
        new[1:] = np.minimum(diag, row[1:] + 1)
        row = np.minimum.accumulate(new - idx) + idx          # insertions along the row
    return int(row[-1])


def test_variant_states_against_joint_enumeration(oracle):
    """osp_cyp_variant_states (alignment to the variant graph) against the definition it has to meet: over all
    compatible subsets of the variants inside the aligned stretch, the subsets of minimum edit distance to the sequence; a variant is
    1 when every such subset holds it, 0 when none does, 2 when they disagree.  Variants sit next to and on top of each other."""
    rng = np.random.default_rng(9)
    n_checked = n_two = 0
    for rep in range(6):
        backbone = "".join(rng.choice(list("ACGT"), 260))
        # a cluster: SNV, an overlapping 2-base deletion, an adjacent insertion, a far SNV and a far deletion
        p = int(rng.integers(60, 120))
        other = lambda c: "ACGT"[("ACGT".index(c) + 1 + int(rng.integers(0, 3))) % 4]
        variants = [(p, backbone[p], other(backbone[p])),
                    (p, backbone[p:p + 3], backbone[p]),
                    (p + 3, backbone[p + 3], backbone[p + 3] + "".join(rng.choice(list("ACGT"), 2))),
                    (p + 5, backbone[p + 5], other(backbone[p + 5])),
                    (p + 60, backbone[p + 60], other(backbone[p + 60])),
                    (p + 80, backbone[p + 80:p + 84], backbone[p + 80])]
        variants.sort(key=lambda v: v[0])
        pos, refs, alts = [v[0] for v in variants], [v[1] for v in variants], [v[2] for v in variants]
        nv = len(variants)
        compatible = []
        for mask in range(1 << nv):
            chosen = [v for v in range(nv) if mask >> v & 1]
            if all(variants[x][0] + len(variants[x][1]) <= variants[y][0] for x, y in zip(chosen, chosen[1:])):
                compatible.append(chosen)

        def applied(chosen, lo, hi):
            out, at = [], lo
            for v in chosen:
                out.append(backbone[at:variants[v][0]]); out.append(variants[v][2]); at = variants[v][0] + len(variants[v][1])
            out.append(backbone[at:hi])
            return "".join(out)
        for k in range(5):
            truth = compatible[int(rng.integers(0, len(compatible)))]
            seq = applied(truth, 0, len(backbone))
            if k % 2:                                                            # a substitution inside the cluster makes ties likely
                q = min(len(seq) - 1, p + int(rng.integers(0, 8)))
                seq = seq[:q] + other(seq[q]) + seq[q + 1:]
            if k == 4:                                                           # a third base at the far SNV: reference and alternate cost the same
                v = next(x for x in range(nv) if variants[x][0] == p + 60)
                third = next(c for c in "ACGT" if c not in (variants[v][1], variants[v][2]))
                seq, truth = backbone[:p + 60] + third + backbone[p + 61:], []
            states, aln = of.oracle_variant_states(oracle, seq, backbone, pos, refs, alts)
            assert aln is not None
            a0, a1, b0, b1, _nm = aln
            inside = [v for v in range(nv) if variants[v][0] >= b0 and variants[v][0] + len(variants[v][1]) <= b1]
            subsets = [c for c in compatible if all(v in inside for v in c)]
            dist = [_edit_distance(seq[a0:a1], applied(c, b0, b1)) for c in subsets]
            best = [set(c) for c, d in zip(subsets, dist) if d == min(dist)]
            for v in range(nv):
                want = 3 if v not in inside else 1 if all(v in c for c in best) else 0 if not any(v in c for c in best) else 2
                assert states[v] == want, (rep, k, v, variants[v], states.tolist(), best)
                n_checked += 1; n_two += want == 2
            if k % 2 == 0 and k < 4:                                             # the isolated variants of a noise-free sequence are the ones it was built from
                far = [v for v in range(nv) if variants[v][0] >= p + 60]             # (inside the cluster another subset may spell the same bases)
                assert [v for v in far if states[v] == 1] == [v for v in truth if v in far]
    assert n_checked >= 100 and n_two >= 1
