"""Shared read sets for the consensus tests (oracle quality on CPU, HIP == oracle on GPU)."""
import numpy as np


def hla_pair(fx, g=0, i=10, j=500):
    fl = fx.full_length_alleles(g)
    return fx.dna[fl[i]], fx.dna[fl[j]]


def cases(fx, synth, oracle):
    """-> list of (name, reads, offsets, cfg kwargs, two_pass)"""
    rng = np.random.default_rng(42)
    s1, s2 = hla_pair(fx)
    out = []
    full1 = [synth.hifi_errors(rng, s1) for _ in range(24)]
    out.append(("single_full_span", full1, None, dict(early_termination=False, dual=False), False))
    out.append(("single_early_termination", full1, None, dict(early_termination=True, dual=False), False))
    mix = [synth.hifi_errors(rng, s1) for _ in range(20)] + [synth.hifi_errors(rng, s2) for _ in range(16)]
    order = rng.permutation(len(mix))
    mix = [mix[i] for i in order]
    out.append(("dual_full_span", mix, None, dict(early_termination=False, dual=True), True))
    out.append(("dual_one_pass", mix, None, dict(early_termination=False, dual=True), False))
    # partial reads with offsets, as run_dual_consensus_with_offsets passes them: None for the leftmost, else start - min + window/2
    parts, starts = [], []
    for hap in (s1, s2):
        for _ in range(18):
            a = int(rng.integers(0, 1500)) if rng.random() < 0.7 else 0
            b = int(rng.integers(len(hap) - 1200, len(hap) + 1))
            parts.append(synth.hifi_errors(rng, hap[a:b]))
            starts.append(max(0, a + int(rng.integers(-30, 31))) if a else 0)       # the caller's offsets are estimates
    mn = min(starts)
    offs = [None if s == mn else s - mn + 200 for s in starts]
    out.append(("dual_offsets", parts, offs, dict(early_termination=True, dual=True), True))
    out.append(("single_offsets", parts[:18], offs[:18], dict(early_termination=True, dual=False), False))
    # homopolymer-compressed sequences (the reference's first attempt)
    hpc = [oracle.hpc(r) for r in mix]
    out.append(("dual_hpc", hpc, None, dict(early_termination=True, dual=True), True))
    # reads with non-ACGT bases and one unrelated read
    noisy = [r for r in full1[:10]]
    noisy[3] = noisy[3][:700] + "N" + noisy[3][701:]
    noisy[5] = noisy[5][:1500] + "NNN" + noisy[5][1503:]
    noisy.append("".join(rng.choice(list("ACGT"), 900)))
    out.append(("single_with_n_and_junk", noisy, None, dict(early_termination=True, dual=False), False))
    # a recurrent artefact (15 % of the reads of both haplotypes) ahead of the real difference: the one-pass rule would split there,
    # the two-pass policy must not (the speculative second pass of the library is rejected and re-run)
    h1 = "".join(rng.choice(list("ACGT"), 600))
    h2 = h1[:300] + ("A" if h1[300] != "A" else "C") + h1[301:]
    art = []
    for i in range(40):
        hap = h1 if i % 2 == 0 else h2
        if i % 20 < 3:
            hap = hap[:100] + ("G" if hap[100] != "G" else "T") + hap[101:]
        art.append(hap)
    out.append(("dual_artefact_column_first", art, None, dict(early_termination=False, dual=True), True))
    out.append(("dual_artefact_one_pass", art, None, dict(early_termination=False, dual=True), False))
    # low coverage: three reads per haplotype (min_count 3 is exactly met)
    few = [synth.hifi_errors(rng, s1) for _ in range(3)] + [synth.hifi_errors(rng, s2) for _ in range(3)]
    out.append(("dual_low_coverage", [few[i] for i in (0, 3, 1, 4, 2, 5)], None, dict(early_termination=False, dual=True), False))
    # homopolymer-biased error: 8 of 20 reads carry one extra base in the longest run of the sequence
    run_at = max(range(len(s1) - 6), key=lambda p: len(s1[p:p + 6].rstrip(s1[p])) == 0 and 6 or 0)
    biased = [(s1[:run_at] + s1[run_at] + s1[run_at:]) if i % 5 < 2 else s1 for i in range(20)]
    out.append(("single_homopolymer_biased", biased, None, dict(early_termination=False, dual=False), False))
    # tiny inputs
    out.append(("two_reads", full1[:2], None, dict(early_termination=True, dual=True), True))
    out.append(("one_read", full1[:1], None, dict(early_termination=False, dual=False), False))
    return out, (s1, s2)
