"""BASELINE configs[3] in miniature: one 30x-WGS-like sample through the library on one GPU context -- a variant gene (K6), HLA-A/-B
with ~45 reads per gene against the full bundled IMGT/HLA database (K1 -> K8 -> K2), CYP2D6 with ~100 reads (K3 -> K8 -> K9/K7 -> K4 -> K5).
Every call must equal the truth the sample was simulated from."""
import numpy as np
import pytest

import oracle_ffi as of
import variant_glue as vg

pytestmark = pytest.mark.gpu


def test_wgs_like_sample(oracle, pkg, gpu_ctx):
    from pb_starphase_amd import synth
    import cyp_fixture as cf
    from test_gpu_variant import gpu_struct
    rng = np.random.default_rng(2024)
    # ---- variant gene: CACNA1S heterozygous for c.3257G>A (src/diplotyper.rs:1652-1790 scenario, committed fixture)
    _g, prob = vg.load_case(vg.ProductNormalizer(pkg), "CACNA1S", "CACNA1S/het.vcf.gz", False)
    called = vg.call_gene(oracle, prob, solver=lambda pr: gpu_ctx.variant_solve(gpu_struct(pkg, pr)))
    assert [frozenset(d) for d in called["diplotypes"]] == [frozenset(("Reference", "c.3257G>A"))]
    # ---- HLA: 45 reads per gene, reads only partly overlap the gene (offsets are exercised)
    fx = synth.HlaFixture()
    db = fx.make_db(pkg, gpu_ctx)
    truth, reads = {}, []
    for g in range(len(fx.genes)):
        pick = rng.choice(fx.full_length_alleles(g), 2, replace=False).tolist()
        truth[g] = sorted(pick)
        for a in pick:
            hap, s = fx.haplotype(g, a)
            reads += synth.simulate_reads(rng, hap, s, len(fx.dna[a]), 23, mean_len=6000, sd_len=1500, min_overlap=2500)
    reads = [reads[i] for i in rng.permutation(len(reads))]
    R = gpu_ctx.upload(reads)
    k1 = db.realign_reads(R)
    calls, _ = db.diplotype_genes(list(range(len(fx.genes))), R, k1)
    for g, (call, c1, c2) in enumerate(calls):
        got = sorted([call.allele1, call.allele2])
        same = lambda a, b: a == b or (a >= 0 and b >= 0 and fx.cdna[a] == fx.cdna[b] and fx.dna[a] == fx.dna[b])
        assert call.status == 0 and call.is_dual and call.dual_passed, (g, call.n_reads, call.counts1, call.counts2)
        assert all(same(a, b) for a, b in zip(got, truth[g])) or all(same(a, b) for a, b in zip(got, truth[g][::-1])), (g, got, truth[g])
    # ---- CYP2D6: ~100 reads, *1/*4
    locus = synth.CypLocus(seed=11)
    cdb, d6 = cf.make_db(locus, synth, np.random.default_rng(5))
    creads = cf.sample(locus, synth, rng, d6, "*1/*4", 100)
    call, cons, labels = gpu_ctx.cyp_diplotype(gpu_ctx.upload(cdb.seqs), cdb.types, cdb.subtypes, cdb.deep, cdb.backbone, cdb.variants, cdb.is_vi,
                                                cdb.allele_subtypes, cdb.hap_matrix, of.default_cyp_config(), gpu_ctx.upload(creads))
    assert call.status == 0 and {call.hap1.decode(), call.hap2.decode()} == {"*1", "*4"}
