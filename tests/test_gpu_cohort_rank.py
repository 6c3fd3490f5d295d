"""BASELINE configs[4] as one rank sees it: 256 samples sharded over 8 ranks -> this rank's 32 samples.  All of them go through one K1
call + sp_hla_diplotype_cohort (HLA-A / -B) and one sp_cyp_diplotype_cohort call (CYP2D6); four of them also get their 18 variant-gene calls (one sp_variant_solve_batch call); the per-(sample,
gene) records of the rank go through the gather of pb-starphase_amd/shard.py (world size 1 here; tests/test_shard_gloo.py and
tests/test_gpu_bench.py run it with two ranks).  Every HLA and CYP2D6 call equals the simulated truth, every variant-gene solve the oracle."""
import gzip
import json
import os

import numpy as np
import pytest

import cyp_cases_real as cr
import variant_glue as vg
from test_gpu_panel import synthetic_observations

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_one_rank_of_the_cohort(oracle, pkg, gpu_ctx):
    from pb_starphase_amd import synth, shard
    from test_gpu_variant import gpu_struct
    world, rank, n_total = 8, 3, 256
    mine = shard.partition(n_total, world, rank)
    assert len(mine) == 32 and mine[0] == 96
    fx = synth.HlaFixture()
    db = fx.make_db(pkg, gpu_ctx)
    genes = list(range(len(fx.genes)))
    reads, sample_of, truth = [], [], {}
    for k, s in enumerate(mine):
        rng = np.random.default_rng(10_000 + s)                                 # a sample's data depends on its global id, not on the rank
        for g in genes:
            pick = sorted(rng.choice(fx.full_length_alleles(g), 2, replace=False).tolist())
            truth[(s, g)] = pick
            for a in pick:
                hap, st = fx.haplotype(g, a)
                rs = synth.simulate_reads(rng, hap, st, len(fx.dna[a]), 22, mean_len=7000, sd_len=1500, min_overlap=2500)
                reads += rs; sample_of += [k] * len(rs)
    R = gpu_ctx.upload(reads)
    k1 = db.realign_reads(R)
    cohort, _ = db.diplotype_cohort(len(mine), sample_of, genes, R, k1)
    same = lambda a, b: a == b or (a >= 0 and b >= 0 and fx.cdna[a] == fx.cdna[b] and fx.dna[a] == fx.dna[b])
    records = []
    for k, s in enumerate(mine):
        for g in genes:
            c = cohort[k][g][0]
            got = sorted([c.allele1, c.allele2])
            assert c.status == 0 and all(same(x, y) for x, y in zip(got, truth[(s, g)])), (s, g, got, truth[(s, g)])
            records.append((s, g, c.allele1, c.allele2))
    # CYP2D6 and the variant genes of four of the rank's samples
    cfg, gene_def = cr.load_db()
    locus = synth.Chr22Locus(cfg, gene_def, seed=3)
    cdb = pkg.ffi.CypDb(gpu_ctx, cfg, gene_def, locus.sequence, locus.start)
    scen = cr.scenarios(locus)
    entries = json.load(gzip.open(os.path.join(GOLDEN, "gene_entries_v0.14.1.json.gz")))["gene_entries"]
    prepared = {name: vg.load_database_haplotypes(oracle, entries[name], None) for name in sorted(entries)}
    # CYP2D6 of every sample of the rank: one sp_cyp_diplotype_cohort call, the samples spread over the context's streams
    cyp_sets = [gpu_ctx.upload(locus.sample(np.random.default_rng(20_000 + s), scen[s % 3][1], 100, lo=8000, hi=16000)) for s in mine]   # *1/*2, *4/*4, *5/*1
    for s, (call, _cons, rc) in zip(mine, cdb.diplotype_cohort(cyp_sets)):
        expected = scen[s % 3][2]
        assert rc == 0 and call.status == 0 and sorted([call.hap1.decode(), call.hap2.decode()]) == sorted(expected), (s, call.hap1, call.hap2, expected)
        records.append((s, len(genes), 0, 0))                                   # (the string call travels beside the integer table in a real run)
    # the variant genes of four samples: their solves in one sp_variant_solve_batch call
    problems, tags = [], []
    for s in mine[:4]:
        rng = np.random.default_rng(30_000 + s)
        for gi, name in enumerate(sorted(entries)):
            vh, hl = prepared[name]
            _h1, _h2, obs, _ph = synthetic_observations(rng, hl)
            problems.append(vg.Problem(vh, hl, obs, entries[name].get("structural_variants"))); tags.append((s, gi, name))
    structs = [gpu_struct(pkg, p) for p in problems]
    for (s, gi, name), prob, got in zip(tags, problems, gpu_ctx.variant_solve_batch(structs)):
        assert got == vg.oracle_solve(oracle, prob), (s, name)
        records.append((s, len(genes) + 1 + gi, 0, 0))
    table = shard.gather_calls(np.array(records, shard.CALL_DTYPE))
    assert len(table) == 32 * 3 + 4 * len(entries) and table["sample"].min() == 96 and table["sample"].max() == 127
    assert (np.diff(table["sample"]) >= 0).all()


def test_all_256_samples_in_one_call_equal_the_single_calls(pkg, gpu_ctx):
    """BASELINE configs[4] on one GPU: the 256 samples of the cohort in ONE sp_hla_diplotype_cohort and ONE sp_cyp_diplotype_cohort call (the samples in lockstep groups
    on the context's streams) against every sample called alone (sp_hla_diplotype_genes on its own reads, sp_cyp_diplotype): the same alleles, statuses and consensus
    sequences, sample by sample."""
    import bench                                                                # (the repository root is on sys.path: tests/conftest.py)
    from pb_starphase_amd import synth
    fx = synth.HlaFixture()
    db = fx.make_db(pkg, gpu_ctx)
    cfg, gene_def = cr.load_db()
    locus = synth.Chr22Locus(cfg, gene_def, seed=3)
    cdb = pkg.ffi.CypDb(gpu_ctx, cfg, gene_def, locus.sequence, locus.start)
    scen = cr.scenarios(locus)
    panel = bench.VariantPanel(pkg)
    n = 256
    share = bench.CohortShare(pkg, fx, locus, scen, panel, list(range(n)))
    genes = list(range(len(fx.genes)))
    R = gpu_ctx.upload(share.hla_reads)
    k1 = db.realign_reads(R)
    cohort, _ = db.diplotype_cohort(n, share.sample_of, genes, R, k1)
    sample_of = np.array(share.sample_of)
    cyp_sets = [gpu_ctx.upload_format(pkg.ffi.SP_SEQ_BAM4, *p) for p in share.cyp_payloads]
    cyp_all = cdb.diplotype_cohort(cyp_sets)
    hla_good = cyp_good = 0
    for s in range(n):
        idx = np.nonzero(sample_of == s)[0]
        Rs = gpu_ctx.upload([share.hla_reads[i] for i in idx])
        k1s = db.realign_reads(Rs)
        for f in ("status", "best_allele", "gene", "nm", "unmapped", "mm2_score", "mm2_nm"):
            assert (k1s[f] == k1[f][idx]).all(), (s, f)                         # a read's K1 record does not depend on the reads beside it
        alone, _ = db.diplotype_genes(genes, Rs, k1s)
        for g in genes:
            a, b = cohort[s][g], alone[g]
            assert (a[0].status, a[0].allele1, a[0].allele2, a[1], a[2]) == (b[0].status, b[0].allele1, b[0].allele2, b[1], b[2]), (s, g)
            hla_good += all(bench.same_allele(fx, x, y) for x, y in zip(sorted([a[0].allele1, a[0].allele2]), share.hla_truth[(s, g)]))
        call, cons, _labels = cdb.diplotype(cyp_sets[s])
        c = cyp_all[s]
        assert c[2] == 0 and (c[0].status, c[0].hap1, c[0].hap2, c[0].n_consensus, c[0].searches_gave_up) == (call.status, call.hap1, call.hap2, call.n_consensus, call.searches_gave_up), s
        assert sorted(c[1]) == sorted(cons), s
        cyp_good += sorted([call.hap1.decode(), call.hap2.decode()]) == sorted(share.cyp_expected[s])
        Rs.close()
    print("256 samples: HLA calls equal to the truth", hla_good, "of", n * len(genes), "; CYP2D6", cyp_good, "of", n)
    assert hla_good >= n * len(genes) - 2 and cyp_good == n                     # (the two HLA misses: tests/test_gpu_concordance.py, the CPU port makes the same calls)
