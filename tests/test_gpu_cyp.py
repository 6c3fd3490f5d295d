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
