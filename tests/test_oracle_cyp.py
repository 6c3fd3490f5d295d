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
