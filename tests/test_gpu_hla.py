"""GPU parity for the fused HLA entry points (K1 sp_hla_realign_reads, K2 sp_hla_score_consensus)."""
import numpy as np
import pytest

import hla_expected as hx

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small(pkg, gpu_ctx):
    from pb_starphase_amd import synth
    fx = synth.HlaFixture(max_alleles_per_gene=90, seed=2)
    db = fx.make_db(pkg, gpu_ctx)
    yield fx, db
    db.close()


def test_k2_score_consensus(oracle, pkg, gpu_ctx, small):
    from pb_starphase_amd import synth
    fx, db = small
    rng = np.random.default_rng(4)
    for g in range(len(fx.genes)):
        cands = [a for a in range(len(fx.ids)) if fx.gene_of[a] == g and fx.dna[a]]
        for trial, a in enumerate(rng.choice(cands, 3, replace=False).tolist()):
            # consensus = the allele inside a little reference flank (gene strand), plus a few edits on later trials
            ref = fx.gene_ref[g] if fx.gene_fwd[g] else synth.revcomp(fx.gene_ref[g])
            dna = ref[:60] + fx.dna[a] + ref[-60:]
            cdna = fx.cdna[a]
            if trial == 1:
                dna = synth.mutate(rng, dna, 2, 1, 1)
                cdna = synth.mutate(rng, cdna, 1, 0, 0)
            if trial == 2:
                cdna = cdna[5:-7]
            for require_dna in (False, True):
                best, n_scored, stats = db.score_consensus(g, dna, cdna, require_dna=require_dna)
                ebest, estats = hx.k2_expected(oracle, fx, g, dna, cdna, require_dna=require_dna)
                assert n_scored == len(estats)
                assert best == ebest, (fx.ids[best] if best >= 0 else None, fx.ids[ebest] if ebest >= 0 else None)
                for al, st in estats.items():
                    assert stats[al].tolist() == st, (al, stats[al], st)
                if trial == 0:
                    assert best == a or fx.cdna[best] == fx.cdna[a]
    # nothing maps: 4-bp consensus (reference: test_score_bad_read, src/hla/caller.rs:1784-1809)
    best, n_scored, stats = db.score_consensus(0, "ACGT", "N", require_dna=True, disable_cdna=True)
    assert best == -1
    assert all(s == [-1] * 6 for s in stats[[a for a in range(len(fx.ids)) if fx.gene_of[a] == 0 and fx.dna[a]]].tolist())


def test_k1_realign_reads(oracle, pkg, gpu_ctx, small):
    from pb_starphase_amd import synth
    fx, db = small
    rng = np.random.default_rng(9)
    reads = []
    for g in range(len(fx.genes)):
        full = fx.full_length_alleles(g)
        for a in rng.choice(full, 2, replace=False).tolist():
            hap, gs = fx.haplotype(g, a)
            reads += synth.simulate_reads(rng, hap, gs, len(fx.dna[a]), 5, mean_len=6500, sd_len=1200)
    # noisy reads: the accepted allele has > 8 and > 32 edits, so the deeper passes of the pruned mode are exercised
    reads.append(synth.mutate(rng, reads[1], 20, 10, 10))
    reads.append(synth.mutate(rng, reads[12], 60, 35, 35))
    reads.append(synth.mutate(rng, reads[7], 150, 60, 60))          # beyond the 3 % cut-off
    reads.append("".join(rng.choice(list("ACGT"), 5000)))          # junk read: no allele
    reads.append(reads[0][:3100])                                    # truncated read
    rs = gpu_ctx.upload(reads)
    out, cells = db.realign_reads(rs, cells=True)
    exp, ecells = hx.k1_expected(oracle, fx, reads)
    assert (cells == ecells).all(), np.argwhere(cells != ecells)[:10]
    n_real = 0
    for r, e in enumerate(exp):
        o = out[r]
        assert o["status"] == e["status"], (r, o, e)
        assert o["best_allele"] == e["best_allele"], (r, o, e)
        if e["best_allele"] >= 0:
            assert o["gene"] == e["gene"]
            assert (o["nm"], o["target_len"], o["unmapped"]) == (e["nm"], e["target_len"], e["unmapped"])
            assert tuple(int(x) for x in o["aln"].tolist()) == e["aln"]
        if e["status"] == 0:
            n_real += 1
            assert (o["seg_start"], o["seg_end"], o["dna_offset"], o["hpc_offset"]) == \
                   (e["seg_start"], e["seg_end"], e["dna_offset"], e["hpc_offset"]), (r, o, e)
    assert n_real >= len(reads) - 5
    assert out[len(reads) - 2]["best_allele"] == -1
    # production mode (no cell matrix requested): exact branch-and-bound must not change a single output field
    for _ in range(3):
        pruned = db.realign_reads(rs)
        assert pruned.tobytes() == out.tobytes()


def test_type_consensus(oracle, pkg, gpu_ctx, small):
    """sp_hla_type_consensus = score_consensus + splice_read (src/hla/caller.rs:1258-1319,1518-1576): the hg38-forward
    consensus goes in, the library places it on the gene reference, splices the cDNA and types it."""
    from pb_starphase_amd import synth
    fx, db = small
    rng = np.random.default_rng(21)
    for g in range(len(fx.genes)):
        full = fx.full_length_alleles(g)
        for trial, a in enumerate(rng.choice(full, 2, replace=False).tolist()):
            hap, s = fx.haplotype(g, a)
            cons = hap[max(0, s - 150):s + len(fx.dna[a]) + 150]
            if trial == 1:
                cons = synth.mutate(rng, cons, 2, 1, 1)
            best, n_scored, stats, cdna = db.type_consensus(g, cons)
            # expected, step by step with the oracle
            ref = fx.gene_ref[g][fx.buffer:len(fx.gene_ref[g]) - fx.buffer]
            d, v = oracle.anchor(fx.gene_ref[g], cons)                     # cons_pos - ref_pos (buffered reference)
            assert v >= 2
            al, ev = oracle.wfa(cons, ref, -d - fx.buffer)
            assert al.ok
            cigar = oracle.cigar(al, ev)
            bam = [(l, {7: 0, 8: 0, 1: 1, 2: 2}[op]) for l, op in cigar]
            if al.a_start:
                bam.insert(0, (al.a_start, 4))
            exons = [(e0 - fx.buffer, e1 - fx.buffer) for e0, e1 in fx.exons[g]]
            segs, _ = oracle.splice_read(al.b_start, bam, exons)
            spliced = "".join(cons[x:y] for x, y in segs)
            fwd = bool(fx.gene_fwd[g])
            e_dna = cons if fwd else synth.revcomp(cons)
            e_cdna = spliced if fwd else synth.revcomp(spliced)
            assert cdna == e_cdna
            ebest, estats = hx.k2_expected(oracle, fx, g, e_dna, e_cdna)
            assert best == ebest and n_scored == len(estats)
            for al_i, st in estats.items():
                assert stats[al_i].tolist() == st
            if trial == 0:
                assert best == a or (fx.cdna[best] == fx.cdna[a] and fx.dna[best] == fx.dna[a])
    # empty / unalignable consensus => unknown (caller.rs:1263-1267,1282-1287)
    assert db.type_consensus(0, "")[0:2] == (-1, 0)
    junk = "".join(rng.choice(list("ACGT"), 3000))
    assert db.type_consensus(0, junk)[0] == -1
