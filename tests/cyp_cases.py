"""The reference's find_best_chain_pair scenarios (src/cyp2d6/chaining.rs:912-1195) as data, shared by the oracle tests
and the GPU parity tests, plus a synthetic generator for bigger problems."""
import math

import numpy as np

import oracle_ffi as of


def create_pairwise_chains(num_labels, chains):
    """src/cyp2d6/chaining.rs:918-947 create_pairwise_chains"""
    obs, scores = {}, {}
    idx = 0
    for chain in chains:
        assert len(chain) >= 2
        for w in range(len(chain) - 1):
            name = f"read_{idx}"
            obs[name] = [list(chain[w:w + 2])]
            weights = []
            for h in chain:
                row = [(100, 1.0)] * num_labels
                row[h] = (0, 1.0)
                weights.append(row)
            scores[name] = weights
            idx += 1
    return obs, scores


def reference_cases():
    cases = []
    # test_find_best_chain_pair (:950-981)
    labels = [("CYP2D6", "A"), ("CYP2D6", "B"), ("CYP2D6", "C"), ("CYP2D6", "D")]
    obs = {"seq_1": [[0, 2]], "seq_2": [[1, 1]]}
    sc = {"seq_1": [[(0, 1.0), (1, 1.0), (1, 1.0), (1, 1.0)], [(1, 1.0), (1, 1.0), (0, 1.0), (1, 1.0)]],
          "seq_2": [[(1, 1.0), (0, 1.0), (1, 1.0), (1, 1.0)], [(1, 1.0), (0, 1.0), (1, 1.0), (1, 1.0)]]}
    cases.append(("basic", of.ChainInputs(labels, obs, sc, False, True, of.DEFAULT_PENALTIES, True), 0, [[0, 2], [1, 1]], ["3_CYP2D6*D"]))
    # test_ambiguous_find_best_chain_pair (:984-1048)
    labels = [("CYP2D6", "A"), ("CYP2D6", "B")]
    obs = {"seq_0": [[1]], "seq_1": [[1, 0]], "seq_2": [[0, 0]], "seq_3": [[0]], "seq_4": [[1]], "seq_5": [[1, 0]], "seq_6": [[0]]}
    b, a = [(10, 1.0), (0, 1.0)], [(0, 1.0), (10, 1.0)]
    sc = {"seq_0": [b], "seq_1": [b, a], "seq_2": [a, a], "seq_3": [a], "seq_4": [b], "seq_5": [b, a], "seq_6": [a]}
    ln01 = -math.log(0.01)
    cases.append(("ambiguous_no_lasso", of.ChainInputs(labels, obs, sc, False, True, (0.0, ln01, 0.0, 2.0), True), 0, [[1], [1, 0, 0, 0]], []))
    cases.append(("ambiguous_lasso", of.ChainInputs(labels, obs, sc, False, True, (3.0, ln01, 0.0, 2.0), True), 0, [[1], [1, 0, 0]], []))
    # test_inferred_alleles (:1083-1136)
    labels = [("CYP2D6", "3"), ("link_region", None), ("REP7", None), ("spacer", None), ("CYP2D7", None), ("CYP2D6", "4"),
              ("Hybrid", "CYP2D6::CYP2D7::exon2")]
    obs, sc = create_pairwise_chains(len(labels), [[0, 1], [2, 3, 4], [5, 1], [2, 3, 6]])
    cases.append(("inferred_off", of.ChainInputs(labels, obs, sc, False, True, of.DEFAULT_PENALTIES, False), 0, [[0, 1], [5, 1]],
                  ["2_REP7", "3_spacer", "4_CYP2D7", "6_CYP2D6::CYP2D7::exon2"]))
    cases.append(("inferred_on", of.ChainInputs(labels, obs, sc, True, True, of.DEFAULT_PENALTIES, False), 0,
                  [[0, 1, 2, 3, 4], [5, 1, 2, 3, 6]], []))
    # test_chaining_errors (:1139-1162)
    labels = [("CYP2D7", None), ("link_region", None), ("spacer", None), ("UNKNOWN", None)]
    cases.append(("no_head", of.ChainInputs(labels, {}, {}, False, True, of.DEFAULT_PENALTIES, False), 16, None, None))
    # test_double5_targeted (:1165-1195)
    labels = [("CYP2D6*5", None)]
    obs = {f"read{x}": [[0]] for x in range(2)}
    sc = {f"read{x}": [[(0, 1.0)]] for x in range(2)}
    cases.append(("double5", of.ChainInputs(labels, obs, sc, True, False, of.DEFAULT_PENALTIES, False), 0, [[0], [0]], []))
    return cases


def danglers(inp, res, oracle=None):
    """DanglingAllele warnings (chaining.rs:575-589): "<index>_<full_allele>" for consensuses not in the result"""
    used = set(list(res.chain1[:res.n1]) + list(res.chain2[:res.n2]))
    names = []
    inv = {v: k for k, v in of.REGION_TYPES.items()}
    for i in range(inp.H):
        if i in used:
            continue
        t, s = inv[int(inp.types[i])], inp.subtypes[i]
        if t == "CYP2D6" and s:
            full = f"CYP2D6*{s}"
        elif t == "Hybrid" and s:
            full = s
        elif t == "FalseAllele" and s:
            full = f"FalseAllele_{s}"
        else:
            full = t
        names.append(f"{i}_{full}")
    return names


def synthetic_problem(rng, n_d6=4, n_reads=60, noise=0.15, infer=True):
    """A CYP2D6-like locus: REP6 -> D6 alleles -> link -> REP7 -> spacer -> D7, two haplotypes with duplications/hybrids;
    reads see windows of 1..4 consecutive regions with noisy edit distances."""
    labels = [("REP6", None)]
    d6 = []
    stars = ["1", "2", "4", "10", "41", "17", "35", "9"]
    for k in range(n_d6):
        d6.append(len(labels))
        labels.append(("CYP2D6", stars[k % len(stars)] + (".00%d" % (1 + k // len(stars)))))
    hyb = len(labels); labels.append(("Hybrid", "CYP2D6::CYP2D7::exon2"))
    link = len(labels); labels.append(("link_region", None))
    rep7 = len(labels); labels.append(("REP7", None))
    spacer = len(labels); labels.append(("spacer", None))
    d7 = len(labels); labels.append(("CYP2D7", None))
    H = len(labels)

    def hap(alleles):
        c = [0]
        for i, a in enumerate(alleles):
            c += [a, link, rep7]
            if i + 1 < len(alleles):
                pass
        c += [spacer, d7]
        return c
    h1 = hap([d6[int(rng.integers(n_d6))]] + ([hyb] if rng.random() < 0.4 else []))
    h2 = hap([d6[int(rng.integers(n_d6))]] * (2 if rng.random() < 0.4 else 1))
    obs, sc = {}, {}
    for r in range(n_reads):
        src = h1 if rng.random() < 0.5 else h2
        ln = int(rng.integers(1, 5))
        s = int(rng.integers(0, max(1, len(src) - ln + 1)))
        win = src[s:s + ln]
        rows, best = [], []
        for h in win:
            row = []
            for x in range(H):
                same_type = labels[x][0] == labels[h][0]
                base = 0 if x == h else (int(rng.integers(3, 40)) if same_type else int(rng.integers(200, 900)))
                if rng.random() < noise:
                    base += int(rng.integers(0, 4))
                ov = 1.0 if same_type else float(np.round(rng.random() * 0.5, 3))
                if rng.random() < noise:
                    ov = float(np.round(0.6 + 0.4 * rng.random(), 3))
                row.append((base, ov))
            rows.append(row)
            mn = min(v[0] for v in row)
            best.append([i for i, v in enumerate(row) if v[0] == mn])
        chains = [[]]
        for opts in best:                                  # cartesian extension by arg-min consensuses (caller.rs:462-487)
            chains = [c + [o] for c in chains for o in opts][:8]
        name = "read_%04d" % r
        obs[name] = chains
        sc[name] = rows
    return labels, obs, sc, infer


# test_weight_sequence (src/cyp2d6/chaining.rs:1051-1080): three consensuses that differ in one base, a segment equal to the first one,
# and the same segment with that base unknown
_WS_HEAD = "AGCCCATTCTGGCCCCTTCCCCACATGCCAGGACAATGTAGTCCTTGTCACCAATCTGGGCAGTCAGAGTTGGGTCAGTGGGG"
_WS_TAIL = ("ACATGGGATTATGGGCAAGGGTAACAGCCCATTCTGGCCCCTTCCCCACATGCCAGGACAATGTAGTCCTTGTCACCAATCTGGGCAGTCAGAGTTGGGTCAGTGGGGGACATGGGATTATGGGCAAGGGTAAC")
WEIGHT_SEQUENCE_CONSENSUS = [_WS_HEAD + b + _WS_TAIL for b in "ACG"]
WEIGHT_SEQUENCE_SEGMENTS = [_WS_HEAD + "A" + _WS_TAIL, _WS_HEAD + "N" + _WS_TAIL]
