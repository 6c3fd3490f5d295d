"""oracle/consensus.c: the consensus contract does what the reference needs from waffle_con (parity unpinned: no reference test
runs a consensus) -- it reconstructs the haplotypes the reads were drawn from and separates the reads."""
import numpy as np

import consensus_cases
import oracle_ffi as of


def run_case(oracle, reads, offs, kw, two_pass):
    cfg = of.cons_config(**kw)
    if two_pass:
        return of.dual_consensus_two_pass(lambda rd, o, c: of.oracle_consensus(oracle, rd, o, c), reads, offs, cfg)
    return of.oracle_consensus(oracle, reads, offs, cfg)


def test_consensus_quality(pkg, oracle):
    from pb_starphase_amd import synth
    fx = synth.HlaFixture()
    cs, (s1, s2) = consensus_cases.cases(fx, synth, oracle)
    got = {name: run_case(oracle, reads, offs, kw, tp) for name, reads, offs, kw, tp in cs}
    assert got["single_full_span"]["cons"] == [s1, None] and got["single_early_termination"]["cons"][0] == s1
    assert int(got["single_full_span"]["score1"].max()) <= 15
    d = got["dual_full_span"]
    assert d["is_dual"] and set(d["cons"]) == {s1, s2}
    assert 14 <= int(d["is_cons1"].sum()) <= 22 and int(d["is_cons1"].sum()) in (16, 20)         # the two read groups, exactly
    # one pass splits at the first column with >= 10 % support for a second base; with these reads that is a real difference too
    assert got["dual_one_pass"]["is_dual"]
    o = got["dual_offsets"]
    assert o["is_dual"]
    for c in o["cons"]:
        assert c in s1 or c in s2 or s1 in c or s2 in c or min(_ed_to(c, s1), _ed_to(c, s2)) <= 3
    assert got["single_offsets"]["cons"][0] in s1 or _ed_to(got["single_offsets"]["cons"][0], s1) <= 3
    h = got["dual_hpc"]
    assert h["is_dual"] and set(h["cons"]) == {oracle.hpc(s1), oracle.hpc(s2)}
    n = got["single_with_n_and_junk"]
    assert n["cons"][0] == s1 and n["score1"][-1] == -1 or n["score1"][-1] > 100                   # the unrelated read is lost or far away
    assert got["one_read"]["cons"][0] == cs[-1][1][0]
    # a recurrent artefact at 15 % ahead of the real 50 % difference: a greedy first-column split would take the artefact (column 100);
    # the best-first search pays for that choice later, comes back to the node that kept one consensus and splits at the real
    # difference (column 300) -- it expands more nodes than the consensus has columns
    for a in (got["dual_artefact_column_first"], got["dual_artefact_one_pass"]):
        assert a["is_dual"] and a["split_at"] == 300 and int(a["is_cons1"].sum()) == 20
        assert a["nodes_expanded"] > 600
    low = got["dual_low_coverage"]
    assert low["is_dual"] and set(low["cons"]) == {s1, s2} and int(low["is_cons1"].sum()) == 3
    hp = got["single_homopolymer_biased"]
    assert hp["cons"][0] == s1                                            # 40 % of the reads carry an extra base in a run: the cheaper form wins


def _ed_to(a, b):
    """edit distance of a to its best-matching substring of b (plain DP; small inputs only)"""
    prev = np.zeros(len(b) + 1, np.int32)
    for i in range(1, len(a) + 1):
        cur = np.empty(len(b) + 1, np.int32)
        cur[0] = i
        sub = prev[:-1] + (np.frombuffer(b.encode(), np.uint8) != ord(a[i - 1]))
        best = np.minimum(sub, prev[1:] + 1)
        for j in range(1, len(b) + 1):
            cur[j] = min(best[j - 1], cur[j - 1] + 1)
        prev = cur
    return int(prev.min())
