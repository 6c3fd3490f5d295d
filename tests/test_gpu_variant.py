"""GPU parity for K6 (sp_variant_solve) against oracle/variant.c on the reference's scenarios and on random problems."""
import ctypes as C

import numpy as np
import pytest

import variant_glue as vg
from test_oracle_variant import CASES, SV_CASES

pytestmark = pytest.mark.gpu


def gpu_struct(pkg, prob):
    p = pkg.ffi.sp_variant_problem()
    s = prob.struct()
    for f, _ in p._fields_:
        setattr(p, f, getattr(s, f))
    return p


@pytest.mark.parametrize("case", CASES, ids=lambda c: c[1])
def test_reference_scenarios(oracle, pkg, gpu_ctx, case):
    db, vcf, with_ref, dips, inexact = case
    _g, prob = vg.load_case(oracle, db, vcf, with_ref)
    exp = vg.oracle_solve(oracle, prob)
    got = gpu_ctx.variant_solve(gpu_struct(pkg, prob))
    assert got == exp
    called = vg.call_gene(oracle, prob, solver=lambda pr: gpu_ctx.variant_solve(gpu_struct(pkg, pr)))
    assert [frozenset(d) if d[0] != d[1] else d for d in called["diplotypes"]] == [frozenset(d) if d[0] != d[1] else d for d in dips]
    # the whole product path: strings normalised by the library (sp_variant_normalize), search on the GPU -- same problem, same call
    _g2, prob2 = vg.load_case(vg.ProductNormalizer(pkg), db, vcf, with_ref)
    assert prob2.obs == prob.obs and [h["slots"] for h in prob2.haps] == [h["slots"] for h in prob.haps]
    assert gpu_ctx.variant_solve(gpu_struct(pkg, prob2)) == exp


@pytest.mark.parametrize("case", SV_CASES, ids=lambda c: c[0])
def test_sv_scenarios(oracle, pkg, gpu_ctx, case):
    """test_multiple_sv_haplotypes (src/diplotyper.rs:2276-2304): SV records labelled by sp_variant_is_deletion, solved by K6,
    strings built by sp_inexact_haplotype / sp_diplotype_string"""
    sv_vcf, dips, inexact = case
    _g, prob = vg.load_case(vg.ProductNormalizer(pkg), "DPYD-sv-test", "DPYD-sv-test/empty_small.vcf.gz", True, sv_vcf_key=sv_vcf,
                            is_deletion=lambda defs, s, e: defs.is_deletion(s, e))
    _g, oprob = vg.load_case(oracle, "DPYD-sv-test", "DPYD-sv-test/empty_small.vcf.gz", True, sv_vcf_key=sv_vcf)
    assert prob.obs == oprob.obs and prob.obs_sv.tolist() == oprob.obs_sv.tolist()
    assert gpu_ctx.variant_solve(gpu_struct(pkg, prob)) == vg.oracle_solve(oracle, oprob)
    got = vg.call_gene(oracle, prob, solver=lambda pr: gpu_ctx.variant_solve(gpu_struct(pkg, pr)))
    assert got["diplotypes"] == dips
    L = pkg.ffi.lib()

    def full(hap):
        rel = dict(Match=1, Unexpected=2, Missing=3)
        vs = list(hap[1])
        labels = (C.c_char_p * max(1, len(vs)))(*[v[0].encode() for v in vs])
        vi, st = np.array([v[1] for v in vs] or [0], np.uint8), np.array([rel[v[2]] for v in vs] or [0], np.int32)
        mt, out = C.c_int32(0), C.create_string_buffer(512)
        L.sp_inexact_haplotype(hap[0].encode(), len(vs), labels, vi.ctypes.data, st.ctypes.data, C.byref(mt), out, 512)
        return out.value
    if inexact is None:
        assert got["inexact"] is None
    else:
        strings = []
        for a, b in got["inexact"]:
            out = C.create_string_buffer(512)
            L.sp_diplotype_string(full(a), full(b), 0, out, 512)
            strings.append(out.value.decode())
        assert strings == inexact


class RandomProblem(vg.Problem):
    def __init__(self, rng, n_vars, n_haps, n_obs):
        self.var_list = list(range(n_vars))
        self.haps = []
        slot_off, alt_off, alt_var = [0], [0], []
        for h in range(n_haps):
            ns = int(rng.integers(0, 6))
            used = rng.choice(n_vars, ns, replace=False).tolist() if ns else []
            for v in used:
                alts = [v]
                if rng.random() < 0.2:
                    alts.append(-1)                                # optional variant (None alternative)
                if rng.random() < 0.15:
                    alts.append(int(rng.integers(n_vars)))         # IUPAC-style alternative
                alt_var += alts
                alt_off.append(len(alt_var))
            slot_off.append(len(alt_off) - 1)
            self.haps.append({"name": f"h{h}", "core_allele": None if rng.random() < 0.5 else "c"})
        self.hap_is_sv = (rng.random(n_haps) < 0.05).astype(np.uint8)
        self.hap_is_core = np.array([1 if h["core_allele"] is None else 0 for h in self.haps], np.uint8)
        self.slot_off = np.array(slot_off, np.int32)
        self.alt_off = np.array(alt_off, np.int32)
        self.alt_var = np.array(alt_var or [0], np.int32)
        self.var_is_core = (rng.random(n_vars) < 0.6).astype(np.uint8)
        self.obs = sorted(rng.choice(n_vars, n_obs, replace=False).tolist())
        self.obs_var = np.array(self.obs, np.int32)
        self.obs_gt = rng.choice([1, 2, 3, 4], n_obs).astype(np.int32)
        ps = rng.integers(100, 103, n_obs).astype(np.int64)
        self.obs_ps = np.where((self.obs_gt == 2) | (self.obs_gt == 3), ps, -1).astype(np.int64)
        self.obs_sv = np.where(rng.random(n_obs) < 0.03, rng.integers(0, 3, n_obs), -1).astype(np.int32)


def test_random_problems(oracle, pkg, gpu_ctx):
    rng = np.random.default_rng(33)
    n_nontrivial = 0
    for k in range(40):
        prob = RandomProblem(rng, n_vars=24, n_haps=int(rng.integers(3, 60)), n_obs=int(rng.integers(0, 11)))
        exp = vg.oracle_solve(oracle, prob)
        got = gpu_ctx.variant_solve(gpu_struct(pkg, prob))
        assert got == exp, (k, got, exp)
        n_nontrivial += len(exp[1]) > 0
    assert n_nontrivial >= 20


def test_or_group_with_a_hom_and_a_het_alternative(oracle, pkg, gpu_ctx):
    """A side's variant list is the homozygous calls first, then its heterozygous ones (solve_diplotype, src/diplotyper.rs:1219-1222,1269-1317):
    when both alternatives of an OR-group are observed -- one homozygous, one heterozygous -- the homozygous one takes the slot whatever
    the order of the calls, and the heterozygous one counts as extra (quant_match, src/data_types/normalized_variant.rs:443-462)"""
    rng = np.random.default_rng(34)
    # the problem that showed it (the 45th of seed 34, written out): hets 1, 9; homs 16, 20; haplotype h0 = {9|16}, {21}, {6}
    p = RandomProblem.__new__(RandomProblem)
    p.var_list = list(range(24))
    p.haps = [{"name": f"h{h}", "core_allele": None if h == 4 else "c"} for h in range(7)]
    p.hap_is_sv = np.zeros(7, np.uint8); p.hap_is_core = np.array([0, 0, 0, 0, 1, 0, 0], np.uint8)
    p.slot_off = np.array([0, 3, 4, 7, 9, 10, 14, 15], np.int32)
    p.alt_off = np.array([0, 2, 3, 4, 5, 6, 7, 10, 11, 12, 13, 14, 15, 16, 18, 19], np.int32)
    p.alt_var = np.array([9, 16, 21, 6, 18, 17, 22, 7, -1, 8, 17, 2, 8, 6, 22, 2, 3, 15, 0], np.int32)
    p.var_is_core = np.array([0, 1, 1, 0, 0, 1, 0, 1, 1, 1, 0, 1, 1, 1, 1, 1, 0, 1, 0, 1, 1, 0, 0, 0], np.uint8)
    p.obs = [1, 9, 16, 20]; p.obs_var = np.array([1, 9, 16, 20], np.int32); p.obs_gt = np.array([1, 1, 4, 4], np.int32)
    p.obs_ps = np.full(4, -1, np.int64); p.obs_sv = np.full(4, -1, np.int32)
    exp = vg.oracle_solve(oracle, p)
    assert exp[0] == (0, 4, 2, 2) and sorted(set(d[:2] for d in exp[1])) == [(1, 1), (1, 6), (6, 1), (6, 6)]
    assert gpu_ctx.variant_solve(gpu_struct(pkg, p)) == exp
    # many problems built to have such slots
    hits = 0
    for k in range(200):
        q = RandomProblem(rng, n_vars=8, n_haps=int(rng.integers(3, 12)), n_obs=int(rng.integers(3, 8)))
        hits += any(len(set(q.alt_var[q.alt_off[s]:q.alt_off[s + 1]].tolist()) & set(q.obs_var.tolist())) > 1 for s in range(len(q.alt_off) - 1))
        assert gpu_ctx.variant_solve(gpu_struct(pkg, q)) == vg.oracle_solve(oracle, q), k
    assert hits >= 20


def test_batch_equals_single_solves(oracle, pkg, gpu_ctx):
    """sp_variant_solve_batch: the solves of a panel in one call, handed out to the context's streams == one sp_variant_solve each"""
    rng = np.random.default_rng(34)
    probs = [RandomProblem(rng, n_vars=24, n_haps=int(rng.integers(3, 60)), n_obs=int(rng.integers(0, 11))) for _ in range(60)]
    structs = [gpu_struct(pkg, p) for p in probs]
    single = [gpu_ctx.variant_solve(s) for s in structs]
    for streams in (3, 1):
        gpu_ctx.set_option("hla_split_streams", streams)
        assert gpu_ctx.variant_solve_batch(structs) == single
    gpu_ctx.set_option("hla_split_streams", 3)
    assert single == [vg.oracle_solve(oracle, p) for p in probs]
    assert gpu_ctx.variant_solve_batch([]) == []
