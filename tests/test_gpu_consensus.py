"""GPU parity for K8 (sp_consensus / sp_consensus_dual) against oracle/consensus.c: identical consensus strings, read
assignment, per-read edit counts and split column."""
import numpy as np
import pytest

import consensus_cases
import oracle_ffi as of
from test_oracle_consensus import run_case

pytestmark = pytest.mark.gpu


def same(a, b):
    assert a["cons"] == b["cons"]
    assert a["is_dual"] == b["is_dual"] and a["split_at"] == b["split_at"]
    assert a["is_cons1"].tolist() == b["is_cons1"].tolist()
    assert a["score1"].tolist() == b["score1"].tolist() and a["score2"].tolist() == b["score2"].tolist()
    assert a["nodes_expanded"] == b["nodes_expanded"]                                   # the search took the same path


def gpu_cfg(pkg, ladder=False, **kw):
    """ladder: sp_consensus_priority's retry of searches that give up (a rule of this library, off unless asked for: sp_cons_config.no_retry_ladder)"""
    c = of.cons_config(**kw)
    return pkg.ffi.sp_cons_config(c.min_count, c.dual_max_ed_delta, c.allow_early_termination, c.allow_dual, c.offset_window, c.offset_compare_length, c.min_af,
                                  c.max_queue_size, c.max_capacity_per_size, c.max_nodes_wo_constraint, 0 if ladder else 1)


def test_cases_match_oracle(oracle, pkg, gpu_ctx):
    from pb_starphase_amd import synth
    fx = synth.HlaFixture()
    cs, _ = consensus_cases.cases(fx, synth, oracle)
    for name, reads, offs, kw, two_pass in cs:
        exp = run_case(oracle, reads, offs, kw, two_pass)
        got = gpu_ctx.consensus(gpu_ctx.upload(reads), gpu_cfg(pkg, **kw), offsets=offs, two_pass=two_pass)
        try:
            same(got, exp)
        except AssertionError as e:
            raise AssertionError(f"case {name}: {e}")


def test_read_subsets_and_groups(oracle, pkg, gpu_ctx):
    """the reference re-runs one consensus per read group (src/hla/caller.rs:706-747): read_idx selects the group"""
    from pb_starphase_amd import synth
    fx = synth.HlaFixture()
    cs, (s1, s2) = consensus_cases.cases(fx, synth, oracle)
    name, reads, offs, kw, _ = cs[2]
    R = gpu_ctx.upload(reads)
    dual = gpu_ctx.consensus(R, gpu_cfg(pkg, **kw), two_pass=True)
    kw1 = dict(kw, dual=False)
    seen = set()
    for grp in (np.flatnonzero(dual["is_cons1"]), np.flatnonzero(~dual["is_cons1"])):
        got = gpu_ctx.consensus(R, gpu_cfg(pkg, **kw1), read_idx=grp.astype(np.uint32))
        exp = of.oracle_consensus(oracle, [reads[i] for i in grp], None, of.cons_config(**kw1))
        same(got, exp)
        seen.add(got["cons"][0])
    assert seen == {s1, s2}


def test_random_small_inputs(oracle, pkg, gpu_ctx):
    """short random haplotypes, heavy noise, ragged offsets: every rule of the contract gets exercised"""
    import os
    from pb_starphase_amd import synth
    seeds = [int(x) for x in os.environ.get("SP_FUZZ_SEEDS", "9").split(",")]          # (a list of seeds for a longer hunt)
    n_dual = 0
    for it in range(25 * len(seeds)):
        if it % 25 == 0:
            rng = np.random.default_rng(seeds[it // 25])
        L = int(rng.integers(150, 700))
        h1 = "".join(rng.choice(list("ACGT"), L))
        h2 = synth.mutate(rng, h1, int(rng.integers(1, 4)), int(rng.integers(0, 2)), int(rng.integers(0, 2))) if L > 200 else h1
        reads, offs = [], []
        for _ in range(int(rng.integers(1, 14))):
            hap = h1 if rng.random() < 0.5 else h2
            a = int(rng.integers(0, L // 3)) if rng.random() < 0.5 else 0
            b = int(rng.integers(2 * L // 3, len(hap) + 1))
            reads.append(synth.hifi_errors(rng, hap[a:b], p_sub=0.004, p_ins=0.004, p_del=0.004))
            offs.append(None if a == 0 else a + int(rng.integers(0, 40)))
        if all(o is not None for o in offs):
            offs[0] = None
        kw = dict(early_termination=bool(rng.integers(0, 2)), dual=True, min_count=int(rng.integers(1, 4)), min_af=float(rng.choice([0.1, 0.25])),
                  dual_max_ed_delta=int(rng.choice([2, 20, 100])), offset_window=int(rng.choice([60, 120, 400])), offset_compare_length=int(rng.choice([20, 50, 64])))
        two_pass = bool(rng.integers(0, 2))
        exp = run_case(oracle, reads, offs, kw, two_pass)
        got = gpu_ctx.consensus(gpu_ctx.upload(reads), gpu_cfg(pkg, **kw), offsets=offs, two_pass=two_pass)
        try:
            same(got, exp)
        except AssertionError as e:
            raise AssertionError(f"iteration {it}: {e}")
        n_dual += exp["is_dual"]
    assert n_dual >= 5 * len(seeds) - 4


def test_low_complexity_and_divergent_inputs(oracle, pkg, gpu_ctx):
    """the inputs that make a read's states expensive (tests/consensus_fuzz.py): repeats and noisy reads (many tips per column), haplotypes 1-5 % apart with recombinant
    reads (the worse state goes through a window ahead of the other, is dropped at dual_max_ed_delta, or is rebuilt where the other catches up), placement windows
    of 700 bases (the wide-window search)"""
    import consensus_fuzz
    from pb_starphase_amd import synth
    for seed in (1003, 2001, 2005):
        rng = np.random.default_rng(seed)
        for it in range(25):
            _L, reads, offs, kw, two_pass = consensus_fuzz.problem(rng, seed, synth)
            exp = run_case(oracle, reads, offs, kw, two_pass)
            got = gpu_ctx.consensus(gpu_ctx.upload(reads), gpu_cfg(pkg, **kw), offsets=offs, two_pass=two_pass)
            try:
                same(got, exp)
            except AssertionError as e:
                raise AssertionError(f"seed {seed} iteration {it}: {e}")


@pytest.mark.parametrize("side_orders,compound", [(0, 0), (1, 1), (2, 2), (3, 3), (3, 50)])
def test_side_orders_and_branching_windows_leave_the_search_unchanged(oracle, pkg, gpu_ctx, side_orders, compound):
    """sp_ctx_set_option "k8_side_orders" / "k8_compound" (round 6): a step launch also makes the window or expansion another waiting node will need (adopted later without a
    launch), and a window may be ordered with the children of the branch its lookahead votes foresee -- speculation only: consensus, assignment, scores and the number of nodes
    expanded stay the oracle's under every setting (0 / 0 = off; 2 = branching windows ordered but never taken; 3 = ordered wherever one is foreseen; 50 = at a 50 % vote share),
    as launch pairs (the resident kernels carry neither), and the counters show that the paths ran"""
    import consensus_fuzz
    from pb_starphase_amd import synth
    gpu_ctx.set_option("k8_persistent", 0); gpu_ctx.set_option("k8_side_orders", side_orders); gpu_ctx.set_option("k8_compound", compound)
    try:
        gpu_ctx.profile_reset()
        for seed in (5, 1003, 2002, 2005):
            rng = np.random.default_rng(seed)
            for it in range(12):
                _L, reads, offs, kw, two_pass = consensus_fuzz.problem(rng, seed, synth)
                exp = run_case(oracle, reads, offs, kw, two_pass)
                got = gpu_ctx.consensus(gpu_ctx.upload(reads), gpu_cfg(pkg, **kw), offsets=offs, two_pass=two_pass)
                try:
                    same(got, exp)
                except AssertionError as e:
                    raise AssertionError(f"side orders {side_orders}, branching windows {compound}, seed {seed} iteration {it}: {e}")
        made = gpu_ctx.profile_get("cons_side_windows")[2] + gpu_ctx.profile_get("cons_side_expansions")[2]
        adopted, ordered, taken = (gpu_ctx.profile_get(k)[2] for k in ("cons_adopted", "cons_compound", "cons_compound_ok"))
        assert gpu_ctx.profile_get("cons_persistent_batches")[2] == 0
        if side_orders == 0:
            assert made == 0 and adopted == 0
        else:
            assert made > 0 and adopted > 0
        if compound == 0:
            assert ordered == 0 and taken == 0
        elif compound == 2:
            assert ordered > 0 and taken == 0
        else:
            assert ordered > 0 and taken > 0
    finally:
        gpu_ctx.set_option("k8_persistent", 2); gpu_ctx.set_option("k8_side_orders", 1); gpu_ctx.set_option("k8_compound", 1)


def test_persistent_kernels_run_the_same_search(oracle, pkg, gpu_ctx):
    """sp_ctx_set_option "k8_persistent": batches of small problems as two persistent kernels (step workgroups + a control workgroup per problem, handing over through
    one word each in memory: write-through stores, sc1 loads, memory-side atomics) instead of a launch pair per step -- consensus, assignment, scores and the number of nodes expanded are the oracle's, as in the default mode;
    a batch of many problems (each at its own pace, no lockstep) and a multi-way problem included"""
    import consensus_fuzz
    from pb_starphase_amd import synth
    gpu_ctx.set_option("k8_persistent", 1)
    try:
        gpu_ctx.profile_reset()
        for seed in (3, 1002, 2004):
            rng = np.random.default_rng(seed)
            for it in range(12):
                _L, reads, offs, kw, two_pass = consensus_fuzz.problem(rng, seed, synth)
                exp = run_case(oracle, reads, offs, kw, two_pass)
                got = gpu_ctx.consensus(gpu_ctx.upload(reads), gpu_cfg(pkg, **kw), offsets=offs, two_pass=two_pass)
                try:
                    same(got, exp)
                except AssertionError as e:
                    raise AssertionError(f"seed {seed} iteration {it}: {e}")
        assert gpu_ctx.profile_get("cons_persistent_batches")[2] >= 30
        # forty problems in one batch
        rng = np.random.default_rng(9)
        bases = ["".join(rng.choice(list("ACGT"), int(rng.integers(300, 900)))) for _ in range(40)]
        reads = [synth.hifi_errors(rng, b) for b in bases for _ in range(9)]
        S = gpu_ctx.upload(reads)
        kw = dict(early_termination=True, dual=True)
        outs = gpu_ctx.consensus_batch([dict(reads=S, read_idx=np.arange(9 * j, 9 * j + 9, dtype=np.uint32), cfg=gpu_cfg(pkg, **kw)) for j in range(40)])
        for j, got in enumerate(outs):
            same(got, run_case(oracle, reads[9 * j:9 * j + 9], None, kw, False))
        # a multi-way problem (the CYP2D6 caller's shape)
        base = "".join(rng.choice(list("ACGT"), 900))
        alleles = [base] + [synth.mutate(rng, base, 5, 1, 0) for _ in range(2)]
        raw = [synth.hifi_errors(rng, a) for a in alleles for _ in range(8)]
        hpc = [oracle.hpc(r) for r in raw]
        kw = dict(early_termination=True, dual=True, offset_window=100, offset_compare_length=100)
        e_group, e_cons = of.oracle_priority_consensus(oracle, [hpc, raw], of.cons_config(**kw), None, None)
        g_group, g_cons = gpu_ctx.consensus_priority([gpu_ctx.upload(hpc), gpu_ctx.upload(raw)], gpu_cfg(pkg, **kw), None, None)
        assert g_group.tolist() == e_group.tolist() and g_cons == e_cons
    finally:
        gpu_ctx.set_option("k8_persistent", 2)            # (the default: the library decides)


def test_a_batch_whose_control_workgroups_do_not_come_up_runs_as_launch_pairs(oracle, pkg, monkeypatch):
    """the persistent mode's way out: the control workgroups are launched first and the host waits half a second for all of them to have started; when they have not
    (SP_K8_FORCE_READY_TIMEOUT takes that path whatever the device does) the abort word ends the ones that did, both streams are drained and THE SAME batch runs as a launch
    pair per step -- same results --, the context keeps away from the mode for the next 64 batches and says so in sp_ctx_get_info().warning"""
    import consensus_fuzz
    from pb_starphase_amd import synth
    ctx = pkg.Context(0)
    ctx.set_option("k8_persistent", 1)
    monkeypatch.setenv("SP_K8_FORCE_READY_TIMEOUT", "1")
    rng = np.random.default_rng(77)
    cases = [consensus_fuzz.problem(rng, 77, synth) for _ in range(4)]
    for _L, reads, offs, kw, two_pass in cases:
        same(ctx.consensus(ctx.upload(reads), gpu_cfg(pkg, **kw), offsets=offs, two_pass=two_pass), run_case(oracle, reads, offs, kw, two_pass))
    assert ctx.profile_get("cons_persistent_batches")[2] == 0
    assert "launch by launch" in ctx.info()["warning"]
    # without the switch the context is still backing off (64 batches): launch pairs, same results; then the mode comes back
    monkeypatch.delenv("SP_K8_FORCE_READY_TIMEOUT")
    _L, reads, offs, kw, two_pass = cases[0]
    S, cfg = ctx.upload(reads), gpu_cfg(pkg, **kw)
    exp = run_case(oracle, reads, offs, kw, two_pass)
    for _ in range(70):
        got = ctx.consensus(S, cfg, offsets=offs, two_pass=two_pass)
    same(got, exp)
    assert ctx.profile_get("cons_persistent_batches")[2] > 0
    ctx.close()


def test_batch_equals_one_by_one(oracle, pkg, gpu_ctx):
    """independent problems advanced in lockstep (sp_consensus_batch / sp_consensus_dual_batch) give what they give alone"""
    from pb_starphase_amd import synth
    fx = synth.HlaFixture()
    cs, _ = consensus_cases.cases(fx, synth, oracle)
    for two_pass in (False, True):
        chosen = [c for c in cs if c[4] == two_pass]
        sets = [gpu_ctx.upload(reads) for _, reads, _, _, _ in chosen]
        probs = [dict(reads=S, cfg=gpu_cfg(pkg, **kw), offsets=offs) for S, (_, _, offs, kw, _) in zip(sets, chosen)]
        got = gpu_ctx.consensus_batch(probs, two_pass=two_pass)
        for g, (name, reads, offs, kw, tp) in zip(got, chosen):
            try:
                same(g, run_case(oracle, reads, offs, kw, tp))
            except AssertionError as e:
                raise AssertionError(f"case {name} (two_pass={two_pass}): {e}")


def test_priority_consensus(oracle, pkg, gpu_ctx):
    """multi-way consensus (the PriorityConsensusDWFA role, src/cyp2d6/caller.rs:162-270): reads of four similar alleles plus two
    seeded region types, two levels (homopolymer-compressed, raw), clipped reads with offsets"""
    from pb_starphase_amd import synth
    rng = np.random.default_rng(31)
    base = "".join(rng.choice(list("ACGT"), 1500))
    alleles = [base, synth.mutate(rng, base, 4, 1, 0), synth.mutate(rng, base, 3, 0, 1), synth.mutate(rng, synth.mutate(rng, base, 4, 1, 0), 3, 0, 0)]
    rep = "".join(rng.choice(list("ACGT"), 700))
    spacer = "".join(rng.choice(list("ACGT"), 500))
    raw, offs, seeds, truth = [], [], [], []
    for a, seq in enumerate(alleles):
        for _ in range(9):
            clip = int(rng.integers(40, 300)) if rng.random() < 0.3 else 0
            raw.append(synth.hifi_errors(rng, seq[clip:])); offs.append(None if clip == 0 else clip + 50); seeds.append(None); truth.append(a)
    for s_id, seq in ((1, rep), (3, spacer)):
        for _ in range(6):
            raw.append(synth.hifi_errors(rng, seq)); offs.append(None); seeds.append(s_id); truth.append(10 + s_id)
    order = rng.permutation(len(raw))
    raw, offs, seeds, truth = [raw[i] for i in order], [offs[i] for i in order], [seeds[i] for i in order], [truth[i] for i in order]
    hpc = [oracle.hpc(r) for r in raw]
    hoffs = [None if o is None else oracle.hpc_pos(alleles[t], o - 50) + 50 for o, t in zip(offs, truth)]
    kw = dict(early_termination=True, dual=True, offset_window=100, offset_compare_length=100)        # the CYP2D6 caller's configuration (src/cyp2d6/caller.rs:144-159)
    e_group, e_cons = of.oracle_priority_consensus(oracle, [hpc, raw], of.cons_config(**kw), [hoffs, offs], seeds)
    g_group, g_cons = gpu_ctx.consensus_priority([gpu_ctx.upload(hpc), gpu_ctx.upload(raw)], gpu_cfg(pkg, **kw), [hoffs, offs], seeds)
    assert g_group.tolist() == e_group.tolist() and g_cons == e_cons
    # and the grouping is the truth: one group per allele / region type, its raw consensus is that sequence
    by_group = {}
    for g, t in zip(g_group.tolist(), truth):
        by_group.setdefault(g, set()).add(t)
    assert all(len(v) == 1 for v in by_group.values()) and len(by_group) == 6
    want = {**{a: s for a, s in enumerate(alleles)}, 11: rep, 13: spacer}
    for g, ts in by_group.items():
        assert g_cons[g][1] == want[next(iter(ts))]


def test_priority_consensus_retry_ladder_is_opt_in(oracle, pkg, gpu_ctx):
    """the retry of two-way searches that give up (min_af 0.15 .. 0.40) is a rule of this library, not of waffle_con: off unless sp_cons_config.no_retry_ladder = 0.
    A four-allele mixture under search bounds of one node per length (a search that cannot keep its branches): with and without the ladder the library does what the
    oracle's statement of the same rule does"""
    from pb_starphase_amd import synth
    rng = np.random.default_rng(77)
    base = "".join(rng.choice(list("ACGT"), 900))
    alleles = [base] + [synth.mutate(rng, base, 6, 1, 1) for _ in range(3)]
    raw = [synth.hifi_errors(rng, a) for a in alleles for _ in range(7)]
    order = rng.permutation(len(raw))
    raw = [raw[i] for i in order]
    hpc = [oracle.hpc(r) for r in raw]
    for bounds in (dict(max_queue_size=1, max_capacity_per_size=1), dict(max_queue_size=2, max_capacity_per_size=2, max_nodes_wo_constraint=5), {}):
        kw = dict(early_termination=True, dual=True, offset_window=100, offset_compare_length=100, **bounds)
        for ladder in (False, True):
            e_group, e_cons = of.oracle_priority_consensus(oracle, [hpc, raw], of.cons_config(**kw), None, None, retry_ladder=ladder)
            g_group, g_cons = gpu_ctx.consensus_priority([gpu_ctx.upload(hpc), gpu_ctx.upload(raw)], gpu_cfg(pkg, ladder=ladder, **kw), None, None)
            assert g_group.tolist() == e_group.tolist() and g_cons == e_cons, (bounds, ladder)


def test_priority_consensus_of_several_problems_in_lockstep(pkg, gpu_ctx):
    """sp_consensus_priority_many (the samples of a cohort): every problem comes out as it does alone, a problem with too many groups fails alone"""
    from pb_starphase_amd import synth
    kw = dict(early_termination=True, dual=True, offset_window=100, offset_compare_length=100)        # the CYP2D6 caller's configuration (src/cyp2d6/caller.rs:144-159)
    problems, alone = [], []
    for seed, n_alleles, depth in ((41, 3, 8), (42, 1, 5), (43, 4, 7), (44, 2, 12)):
        rng = np.random.default_rng(seed)
        base = "".join(rng.choice(list("ACGT"), int(rng.integers(600, 1400))))
        alleles = [base] + [synth.mutate(rng, base, int(rng.integers(3, 6)), int(rng.integers(0, 2)), int(rng.integers(0, 2))) for _ in range(n_alleles - 1)]
        raw, offs, seeds = [], [], []
        for seq in alleles:
            for _ in range(depth):
                clip = int(rng.integers(40, 200)) if rng.random() < 0.3 else 0
                raw.append(synth.hifi_errors(rng, seq[clip:])); offs.append(None if clip == 0 else clip + 50); seeds.append(None)
        hpc = ["".join(c for i, c in enumerate(r) if i == 0 or r[i - 1] != c) for r in raw]
        hoffs = [None if o is None else max(1, int(o * 0.75)) for o in offs]
        levels = [gpu_ctx.upload(hpc), gpu_ctx.upload(raw)]
        problems.append((levels, gpu_cfg(pkg, **kw), [hoffs, offs], seeds))
        alone.append(gpu_ctx.consensus_priority(levels, gpu_cfg(pkg, **kw), [hoffs, offs], seeds))
    many = gpu_ctx.consensus_priority_many(problems)
    assert [m[0] for m in many] == [0] * 4
    for (st, group_of, cons), (g1, c1) in zip(many, alone):
        assert group_of.tolist() == g1.tolist() and cons == c1
    n_groups = [len(c) for _g, c in alone]
    assert min(n_groups) == 1 and max(n_groups) >= 3
    # max_groups 2: the problems with more groups fail alone (status 6), the others are solved as before
    few = gpu_ctx.consensus_priority_many(problems, max_groups=2)
    assert [f[0] for f in few] == [6 if k > 2 else 0 for k in n_groups] and 0 in [f[0] for f in few] and 6 in [f[0] for f in few]
    for f, (g1, c1), k in zip(few, alone, n_groups):
        if k <= 2:
            assert f[1].tolist() == g1.tolist() and f[2] == c1
    assert gpu_ctx.consensus_priority_many([]) == []


def test_edge_cases(oracle, pkg, gpu_ctx):
    """empty inputs, no read at the start, room too small, bad configuration"""
    import ctypes as C
    reads = ["ACGTACGTAGCTAGCTAGGATCGATCGATCGGCTAGCTAGCATCGACTAGCTACGATCG" * 4] * 5
    S = gpu_ctx.upload(reads)
    # every read is late: nothing starts the consensus (as in the oracle)
    kw = dict(early_termination=True, dual=False)
    got = gpu_ctx.consensus(S, gpu_cfg(pkg, **kw), offsets=[30] * 5)
    same(got, of.oracle_consensus(oracle, reads, [30] * 5, of.cons_config(**kw)))
    assert got["cons"][0] == "" and got["score1"].tolist() == [-1] * 5
    # an empty selection
    got = gpu_ctx.consensus(S, gpu_cfg(pkg, **kw), read_idx=np.zeros(0, np.uint32), cap=64)
    assert got["cons"] == ["", None]
    # the consensus does not fit
    with pytest.raises(pkg.StarphaseError) as e:
        gpu_ctx.consensus(S, gpu_cfg(pkg, **kw), cap=50)
    assert e.value.code == 6
    # offset_compare_length beyond the 128 bases the placement search supports (64 with a window above 512)
    for bad in (dict(offset_compare_length=129), dict(offset_compare_length=100, offset_window=600), dict(offset_compare_length=128, offset_window=400)):
        with pytest.raises(pkg.StarphaseError) as e:
            gpu_ctx.consensus(S, gpu_cfg(pkg, **dict(kw, **bad)))
        assert e.value.code == 1
    # one read, two reads that disagree everywhere
    for rs in (reads[:1], ["ACGT" * 40, "TTGCA" * 30]):
        for two_pass in (False, True):
            kw2 = dict(early_termination=False, dual=True, min_count=1)
            same(gpu_ctx.consensus(gpu_ctx.upload(rs), gpu_cfg(pkg, **kw2), two_pass=two_pass), run_case(oracle, rs, None, kw2, two_pass))


def test_many_problems_in_one_batch(oracle, pkg, gpu_ctx):
    """a cohort-style batch: 40 small independent problems (more than one launch sequence holds) give what they give alone"""
    from pb_starphase_amd import synth
    rng = np.random.default_rng(77)
    probs, exps = [], []
    for k in range(40):
        hap = "".join(rng.choice(list("ACGT"), int(rng.integers(200, 500))))
        other = synth.mutate(rng, hap, 3, 0, 0) if k % 3 else hap
        reads = [synth.hifi_errors(rng, hap if rng.random() < 0.5 else other, p_sub=0.002, p_ins=0.002, p_del=0.002) for _ in range(int(rng.integers(3, 12)))]
        kw = dict(early_termination=bool(k % 2), dual=True, min_count=2)
        probs.append(dict(reads=gpu_ctx.upload(reads), cfg=gpu_cfg(pkg, **kw)))
        exps.append(run_case(oracle, reads, None, kw, True))
    got = gpu_ctx.consensus_batch(probs, two_pass=True)
    for k, (g, e) in enumerate(zip(got, exps)):
        try:
            same(g, e)
        except AssertionError as err:
            raise AssertionError(f"problem {k}: {err}")
