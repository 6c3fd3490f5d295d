"""Pins the oracle's utility restatements to the reference's unit tests (no GPU needed)."""
import ctypes as C
import math

import numpy as np
import pytest


def test_hpc(oracle):
    """src/util/homopolymers.rs:72-78 test_hpc"""
    assert oracle.hpc("AACAAAAAAGGGTAACAA") == "ACAGTACA"


def test_hpc_pos(oracle):
    """src/util/homopolymers.rs:80-92 test_hpc_pos"""
    seq = "AACCCGTTTT"
    for i, c in enumerate(seq):
        assert oracle.hpc_pos(seq, i) == "ACGT".index(c)
    assert oracle.hpc_pos(seq, 100) == 4            # past the end: number of runs


def test_hpc_guide(oracle):
    """src/util/homopolymers.rs:94-100 test_hpc_guide"""
    assert oracle.hpc("GAACCCGTTTT") == "GACGT"
    assert oracle.hpc_pos("ATTGGGGGAACCCGTTTT", 6) == 2


def test_reverse_complement(oracle):
    """src/util/sequence.rs:30-42"""
    assert oracle.revcomp("ACCGGGTN") == "NACCCGGT"
    with pytest.raises(ValueError):
        oracle.revcomp("b")


def _mn(oracle, probs, obs):
    p = np.array(probs, np.float64)
    o = np.array(obs, np.uint64)
    return oracle.L.osp_multinomial_ln_pmf(p.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p), len(probs))


def test_multinomial(oracle):
    """src/util/stats.rs:46-71 test_multinomial"""
    assert abs(_mn(oracle, [1.0], [10]) - 0.0) < 1e-6
    assert abs(_mn(oracle, [0.25, 0.75], [1, 3]) - math.log(4.0 * 0.25 * 0.75 ** 3)) < 1e-6
    assert abs(_mn(oracle, [0.25, 0.75], [3, 1]) - math.log(4.0 * 0.25 ** 3 * 0.75)) < 1e-6
    assert abs(_mn(oracle, [0.25, 0.25, 0.5], [1, 1, 2]) - math.log(12.0 * 0.25 * 0.25 * 0.5 ** 2)) < 1e-6
    assert abs(_mn(oracle, [0.25, 0.25, 0.5], [2, 2, 0]) - math.log(6.0 * 0.25 ** 4)) < 1e-6


def test_statrs_functions(oracle):
    """statrs 0.16 formulas against scipy (the values the reference's tests depend on)"""
    from scipy import stats, special
    L = oracle.L
    for n in (0, 1, 5, 20, 170, 171, 500, 10000):
        assert abs(L.osp_ln_factorial(n) - special.gammaln(n + 1)) < 1e-9 * max(1.0, special.gammaln(n + 1))
    for p, n, x in ((0.5, 23, 3), (0.5, 30, 10), (0.45, 60, 21), (0.5, 35, 18)):
        assert abs(L.osp_binomial_cdf(p, n, x) - stats.binom.cdf(x, n, p)) < 1e-12
        assert abs(L.osp_binomial_ln_pmf(p, n, x) - stats.binom.logpmf(x, n, p)) < 1e-10
    assert abs(L.osp_normal_ln_pdf(20.0, 2.0, 21.0) - stats.norm.logpdf(21.0, 20.0, 2.0)) < 1e-12


def test_splice_read(oracle):
    """splice_read (src/hla/caller.rs:1518-1576) on hand-made records"""
    # read of 30 bases at pos 100: 10M 2I 8M 3D 10M ; exons [95,105) [112,120) [121,140)
    cigar = [(10, 0), (2, 1), (8, 0), (3, 2), (10, 0)]
    segs, off = oracle.splice_read(100, cigar, [(95, 105), (112, 120), (121, 140)])
    # exon 1: ref 100..104 -> read 0..4 ; exon 2: ref 112..117 mapped (118-120 deleted) -> read 14..19 ; exon 3: ref 121..130 -> read 20..29
    assert segs == [(0, 5), (14, 20), (20, 30)]
    assert off == 5
    segs, off = oracle.splice_read(100, [(4, 4), (10, 0)], [(50, 60), (102, 104)])
    assert segs == [(6, 8)] and off == 10 + 0
