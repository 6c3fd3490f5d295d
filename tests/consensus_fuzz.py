"""Random consensus problems by class (shared by profiles/scripts/k8fuzz.py and tests/test_gpu_consensus.py).
seed < 1000  : random haplotypes 150-700 bases, 1-3 edits apart, HiFi-like errors, ragged offsets
seed >= 1000 : low-complexity haplotypes (homopolymers, tandem repeats) and reads with up to 9 % errors: wavefronts keep several tips for many columns
seed >= 2000 : haplotypes 1-5 % apart (as two gene copies are), 400-1,500 bases, 30 % of the reads switch haplotype: the worse state of a read falls dozens of
               edits behind, is dropped at dual_max_ed_delta, or draws level again behind a switch; placement windows up to 700 bases"""
import numpy as np


def problem(rng, seed, synth):
    L = int(rng.integers(150, 700))
    h1 = "".join(rng.choice(list("ACGT"), L))
    perr = 0.004
    if seed >= 1000:
        parts, tot = [], 0
        while tot < L:
            if rng.random() < 0.5:
                motif = "".join(rng.choice(list("ACGT"), int(rng.integers(1, 5)))); seg = (motif * 40)[:int(rng.integers(8, 60))]
            else:
                seg = "".join(rng.choice(list("ACGT"), int(rng.integers(5, 40))))
            parts.append(seg); tot += len(seg)
        h1 = "".join(parts)[:L]
        perr = float(rng.choice([0.004, 0.01, 0.03]))
    h2 = synth.mutate(rng, h1, int(rng.integers(1, 4)), int(rng.integers(0, 2)), int(rng.integers(0, 2))) if L > 200 else h1
    if seed >= 2000:
        L = int(rng.integers(400, 1500)); h1 = "".join(rng.choice(list("ACGT"), L))
        k = max(3, int(L * float(rng.choice([0.01, 0.03, 0.05])) / 1.0)); k = min(k, (L - 40) // 12 - 1)
        h2 = synth.mutate(rng, h1, k - k // 8 - k // 8, k // 8, k // 8)
    reads, offs = [], []
    for _ in range(int(rng.integers(1, 14))):
        hap = h1 if rng.random() < 0.5 else h2
        if seed >= 2000 and rng.random() < 0.3:
            x = int(rng.integers(L // 4, 3 * L // 4)); other = h2 if hap is h1 else h1
            hap = hap[:x] + other[min(x, len(other)):]
        a = int(rng.integers(0, L // 3)) if rng.random() < 0.5 else 0
        b = int(rng.integers(2 * L // 3, len(hap) + 1))
        reads.append(synth.hifi_errors(rng, hap[a:b], p_sub=perr, p_ins=perr, p_del=perr))
        offs.append(None if a == 0 else a + int(rng.integers(0, 40)))
    if all(o is not None for o in offs):
        offs[0] = None
    kw = dict(early_termination=bool(rng.integers(0, 2)), dual=True, min_count=int(rng.integers(1, 4)), min_af=float(rng.choice([0.1, 0.25])),
              dual_max_ed_delta=int(rng.choice([2, 20, 100])), offset_window=int(rng.choice([60, 120, 400] if seed < 2000 else [60, 120, 400, 700])),
              offset_compare_length=int(rng.choice([20, 50, 64] if seed < 2000 else [20, 50, 64, 100, 128])))
    if kw["offset_window"] + kw["offset_compare_length"] > 512 and kw["offset_compare_length"] > 64:        # (the wide-window search compares at most 64 bases)
        kw["offset_compare_length"] = 64
    two_pass = bool(rng.integers(0, 2))
    return L, reads, offs, kw, two_pass
