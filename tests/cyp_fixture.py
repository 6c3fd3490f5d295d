"""Synthetic CYP2D6 typing database + samples for the reads-to-diplotype tests (config 3)."""
import numpy as np

import cyp_pipeline as cp
import oracle_ffi as of


def make_db(locus, synth, rng):
    """backbone = the D6 region as it lies in a haplotype (200 bases of flank each side); a small variant panel; four star alleles"""
    T = of.REGION_TYPES
    backbone = locus.rep6[-200:] + locus.d6 + locus.link[:200]
    pos = sorted(rng.choice(np.arange(400, len(backbone) - 400, 60), 24, replace=False).tolist())
    variants = []
    for k, p in enumerate(pos):
        ref = backbone[p]
        if k % 5 == 3:
            variants.append((p, ref, ref + "".join(rng.choice(list("ACGT"), 2))))
        elif k % 5 == 4:
            variants.append((p, backbone[p:p + 3], ref))
        else:
            variants.append((p, ref, str(rng.choice([c for c in "ACGT" if c != ref]))))
    defs = {"1": [], "10": [0, 5, 11], "2": [2, 7, 8, 13, 19], "4": [1, 3, 4, 9, 16, 21, 23]}
    subtypes = sorted(defs)                                                   # BTreeMap<Cyp2d6RegionLabel, _> order of the star alleles
    hap_matrix = np.zeros((len(subtypes), len(variants)), np.uint8)
    for a, s in enumerate(subtypes):
        hap_matrix[a, defs[s]] = 1
    is_vi = np.zeros(len(variants), np.uint8)
    is_vi[[0, 2, 1, 3]] = 1
    deep = [n == "CYP2D6" for n in locus.template_names]
    sub = [None if t != T["Hybrid"] else n for n, t in zip(locus.template_names, locus.template_types)]
    db = cp.Db(locus.template_names, locus.template_types, sub, locus.templates, deep, backbone, variants, is_vi, subtypes, hap_matrix)
    d6 = {}
    for s in subtypes:
        seq, shift = backbone, 0
        for i in sorted(defs[s], key=lambda i: variants[i][0]):
            p, ref, alt = variants[i]
            seq = seq[:p + shift] + alt + seq[p + shift + len(ref):]
            shift += len(alt) - len(ref)
        d6[s] = seq[200:len(seq) - 200]
    return db, d6


def sample(locus, synth, rng, d6, scenario, n_reads=200):
    normal = locus.haplotype("normal")
    dup = locus.haplotype("dup")
    with_d6 = lambda hap, s: hap.replace(locus.d6, d6[s])
    haps = {"*1/*4": [with_d6(normal, "1"), with_d6(normal, "4")],
            "*2/*10": [with_d6(normal, "2"), with_d6(normal, "10")],
            "*5/*2": [locus.haplotype("deletion"), with_d6(normal, "2")],
            "*4x2/*1": [with_d6(dup, "4"), with_d6(normal, "1")]}[scenario]
    total = sum(len(h) for h in haps)
    reads = []
    for hap in haps:
        for _ in range(int(round(n_reads * len(hap) / total))):
            ln = int(min(len(hap), max(5000, rng.normal(14000, 3000))))
            s = int(rng.integers(0, len(hap) - ln + 1))
            reads.append(synth.hifi_errors(rng, hap[s:s + ln]))
    order = rng.permutation(len(reads))
    return [reads[i] for i in order]
