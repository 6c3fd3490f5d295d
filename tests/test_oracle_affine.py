"""oracle/affine.c -- the two-piece affine re-score the library reports beside its own counts -- against the minimap2 restatement (oracle/mm2.c) on the pair
classes of the hot path: how often (nm, target span, query span) are identical.  Measured when written (larger samples: 900 / 150 / 3,593 pairs):
K1 pairs (read x truth allele and x random alleles of the gene) 100 %, K2 pairs (allele x consensus, a = 5) 100 %, K3 hits inside the 5 % filter 90.8 % on 64
diagonals and 98.7 % on 256 (templates that map to the other paralog bridge gaps wider than 32 bases).  Also the routine's own edge cases."""
import numpy as np
import pytest

import mm2_ffi
import oracle_ffi as of


@pytest.fixture(scope="module")
def mm(oracle):
    return mm2_ffi.Mm2(oracle)


def same(out, h):
    return out[1:] == (h["nm"], h["t_start"], h["t_end"], h["q_start"], h["q_end"])


def test_k1_and_k2_pairs_equal_the_restatement(oracle, mm, pkg):
    from pb_starphase_amd import synth
    fx = synth.HlaFixture()
    wl = synth.Config2Workload(fx, n_reads=400, seed=1000)
    rng = np.random.default_rng(5)
    tot = ok = 0
    for r in rng.choice(len(wl.reads), 40, replace=False):
        g, a = wl.read_truth[r]
        others = [x for x in range(len(fx.ids)) if fx.gene_of[x] == g and fx.dna[x]]
        re = oracle.encode(wl.reads[r])
        for al in [a] + [int(x) for x in rng.choice(others, 2)]:
            te = oracle.encode(fx.dna_fwd(al))
            hits = [h for h in mm.map_pair(te, re) if not h["rev"]]
            if not hits:
                continue
            d, _v = oracle.anchor(te, re)                                      # the library's own diagonal for the pair (read_pos - allele_pos)
            tot += 1
            ok += same(of.oracle_affine(oracle, te, re, d), hits[0])
    assert tot >= 100 and ok == tot
    # score_read's pairs: alleles against a consensus with a = 5
    g, cons, _cdna, _a = wl.consensus[0]
    ce = oracle.encode(cons)
    o5 = mm.opts(a=5)
    tot = ok = 0
    for al in rng.choice([x for x in range(len(fx.ids)) if fx.gene_of[x] == g and fx.dna[x]], 40, replace=False):
        qe = oracle.encode(fx.dna[al])
        hits = [h for h in mm.map_pair(ce, qe, o5) if not h["rev"]]
        if not hits:
            continue
        tot += 1
        ok += same(of.oracle_affine(oracle, ce, qe, hits[0]["q_start"] - hits[0]["t_start"], a=5), hits[0])
    assert tot >= 30 and ok == tot


def test_k3_hits_inside_the_filter(oracle, mm, pkg):
    import cyp_cases_real as cr
    import cpu_port_cyp as cpc
    from pb_starphase_amd import synth
    cfg, gene_def = cr.load_db()
    locus = synth.Chr22Locus(cfg, gene_def, seed=3)
    db, _ccfg = cpc.tables(cfg, gene_def, locus)
    sc = {n: (h, e) for n, h, e in cr.scenarios(locus)}
    reads = locus.sample(np.random.default_rng(7), sc["*4+*68/*1"][0], 300)
    rng = np.random.default_rng(6)
    tot = ok64 = ok256 = 0
    for r in rng.choice(len(reads), 12, replace=False):
        te = oracle.encode(reads[r])
        for t in range(len(db.seqs)):
            qe = oracle.encode(db.seqs[t])
            for h in mm.map_pair(te, qe):
                um = len(qe) - (h["q_end"] - h["q_start"])
                if h["rev"] or max(h["nm"], 0.1) / (len(qe) - um) > 0.05:
                    continue
                tot += 1
                k0 = h["q_start"] - h["t_start"]
                ok64 += same(of.oracle_affine(oracle, te, qe, k0, 64), h)
                ok256 += same(of.oracle_affine(oracle, te, qe, k0, 256), h)
    assert tot >= 150
    assert ok64 >= 0.85 * tot and ok256 >= 0.96 * tot and ok256 >= ok64


def test_edge_cases(oracle):
    seq = "ACGTTGCAAGCTAGCTAGGATCGATTAGCTAGCATCGACTACGATCGTAGCTAGCATGCATGCAT" * 4
    n = len(seq)
    assert of.oracle_affine(oracle, seq, seq, 0) == (n, 0, 0, n, 0, n)                       # identical: every base, nm 0
    assert of.oracle_affine(oracle, seq, seq, 0, a=5) == (5 * n, 0, 0, n, 0, n)
    assert of.oracle_affine(oracle, "", seq, 0) == (0, 0, 0, 0, 0, 0)
    assert of.oracle_affine(oracle, seq, seq, 500)[0] == 0                                   # a band that misses the rectangle
    # a mismatch two bases from the end is clipped with a = 1 (-4 + 2 < 0) and kept with a = 5
    mut = seq[:n - 3] + ("A" if seq[n - 3] != "A" else "C") + seq[n - 2:]
    assert of.oracle_affine(oracle, seq, mut, 0) == (n - 3, 0, 0, n - 3, 0, n - 3)
    assert of.oracle_affine(oracle, seq, mut, 0, a=5) == (5 * (n - 1) - 4, 1, 0, n, 0, n)
    # an N scores -1 and counts in nm; a 3-base deletion is one gap of 6 + 2 * 3
    withn = seq[:100] + "N" + seq[101:]
    assert of.oracle_affine(oracle, seq, withn, 0) == (n - 2, 1, 0, n, 0, n)
    dele = seq[:120] + seq[123:]
    assert of.oracle_affine(oracle, seq, dele, 0) == (n - 3 - 12, 3, 0, n, 0, n - 3)
    # the query contained in the target: spans on both
    assert of.oracle_affine(oracle, seq, seq[40:200], -40) == (160, 0, 40, 200, 0, 160)
