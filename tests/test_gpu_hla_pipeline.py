"""Reads to diplotype through the C ABI (sp_hla_realign_reads + sp_hla_diplotype_gene) against the same pipeline assembled from
the oracle: identical consensus strings, read grouping, typed alleles and calls; and the calls are the simulated truth."""
import numpy as np
import pytest

import hla_expected as hx
import hla_pipeline as hp

pytestmark = pytest.mark.gpu


def simulate(fx, synth, rng, g, alleles, n_per_hap):
    reads = []
    for a in alleles:
        hap, s = fx.haplotype(g, a)
        reads += synth.simulate_reads(rng, hap, s, len(fx.dna[a]), n_per_hap)
    order = rng.permutation(len(reads))
    return [reads[i] for i in order]


def same_alleles(fx, a, b):
    return a == b or (a >= 0 and b >= 0 and fx.cdna[a] == fx.cdna[b] and fx.dna[a] == fx.dna[b])


@pytest.mark.parametrize("scenario", ["het_both_genes", "hom_a_het_b"])
def test_reads_to_diplotype(oracle, pkg, gpu_ctx, scenario):
    from pb_starphase_amd import synth
    fx = synth.HlaFixture(max_alleles_per_gene=150, seed=4)
    db = fx.make_db(pkg, gpu_ctx)
    rng = np.random.default_rng(11 if scenario == "het_both_genes" else 12)
    truth, reads = {}, []
    for g in range(len(fx.genes)):
        fl = fx.full_length_alleles(g)
        pick = rng.choice(fl, 2, replace=False).tolist()
        if scenario == "hom_a_het_b" and g == 0:
            pick = [pick[0], pick[0]]
        truth[g] = pick
        reads += simulate(fx, synth, rng, g, pick, 14)
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    R = gpu_ctx.upload(reads)
    k1_gpu = db.realign_reads(R)
    k1_exp, _ = hx.k1_expected_seeded(oracle, fx, reads)
    for g in range(len(fx.genes)):
        call, c1, c2, is1 = db.diplotype_gene(g, R, k1_gpu)
        exp = hp.diplotype_gene(oracle, fx, g, reads, k1_exp, synth)
        assert call.status == exp["status"] == 0 and call.n_reads == exp["n_reads"]
        assert (c1, c2) == (exp["cons1"], exp["cons2"])
        assert is1.tolist() == [bool(x) for x in exp["is_cons1"]]
        assert (call.is_dual, call.dual_passed, call.used_dna_dual, call.counts1, call.counts2) == \
               (exp["is_dual"], exp["dual_passed"], exp["used_dna_dual"], exp["counts1"], exp["counts2"])
        assert (call.typed1, call.typed2, call.allele1, call.allele2) == (exp["typed1"], exp["typed2"], exp["allele1"], exp["allele2"])
        assert abs(call.maf - exp["maf"]) <= 1e-12 and abs(call.cdf - exp["cdf"]) <= 1e-9
        # and the call is the truth the reads were simulated from
        got = sorted([call.allele1, call.allele2])
        want = sorted(truth[g])
        assert all(same_alleles(fx, a, b) for a, b in zip(got, want)) or all(same_alleles(fx, a, b) for a, b in zip(got, want[::-1])), (g, got, want)
    # all genes in one call (their consensus problems advance in lockstep): same calls, same consensuses
    both, is1_all = db.diplotype_genes(list(range(len(fx.genes))), R, k1_gpu)
    for g, (call, c1, c2) in enumerate(both):
        one, o1, o2, o_is1 = db.diplotype_gene(g, R, k1_gpu)
        assert (c1, c2) == (o1, o2)
        assert bytes(call) == bytes(one)
        assert is1_all[(k1_gpu["status"] == 0) & (k1_gpu["gene"] == g)].tolist() == o_is1.tolist()
    # a gene without reads
    empty = gpu_ctx.upload(["ACGT" * 200])
    call, c1, c2, _ = db.diplotype_gene(0, empty, db.realign_reads(empty))
    assert call.status == 1 and call.n_reads == 0 and (c1, c2) == ("", "")


def test_absent_capable_gene(oracle, pkg, gpu_ctx):
    """the hemizygous branch of the gene loop (src/hla/caller.rs:676-701,919-923; is_hemizygous_better :1583-1653) on the GPU path: a gene
    flagged absent-capable (HLA-DRB3/4/5 in the reference, src/hla/alleles.rs:62-69) with the reads of ONE haplotype at haploid coverage
    is called (absent, allele); with two haplotypes at diploid coverage it stays a heterozygous call.  Library == oracle pipeline."""
    import hla_expected as hx
    import hla_pipeline as hp
    from pb_starphase_amd import synth
    fx = synth.HlaFixture(max_alleles_per_gene=150, seed=4)
    db = fx.make_db(pkg, gpu_ctx)
    rng = np.random.default_rng(404)
    g = 0
    pick = rng.choice(fx.full_length_alleles(g), 2, replace=False).tolist()
    per_hap = 22
    sets = {}
    for name, alleles in (("hemizygous", pick[:1]), ("diploid", pick)):
        reads = []
        for a in alleles:
            hap, st = fx.haplotype(g, a)
            reads += synth.simulate_reads(rng, hap, st, len(fx.dna[a]), per_hap, mean_len=7000, sd_len=1500, min_overlap=2500)
        sets[name] = [reads[i] for i in rng.permutation(len(reads))]
    same = lambda a, b: a == b or (a >= 0 and b >= 0 and fx.cdna[a] == fx.cdna[b] and fx.dna[a] == fx.dna[b])
    for name, reads in sets.items():
        R = gpu_ctx.upload(reads)
        k1 = db.realign_reads(R)
        cfg = pkg.ffi.hla_call_config(absent_capable=True, normalized_coverage=float(per_hap))
        call, c1, c2, is1 = db.diplotype_gene(g, R, k1, cfg=cfg)
        k1_exp, _cells = hx.k1_expected_seeded(oracle, fx, reads)
        exp = hp.diplotype_gene(oracle, fx, g, reads, k1_exp, synth, absent_capable=True, normalized_coverage=float(per_hap))
        assert call.is_hemizygous == exp["is_hemizygous"] == (1 if name == "hemizygous" else 0)
        assert (call.allele1, call.allele2) == (exp["allele1"], exp["allele2"]) and (c1, c2) == (exp["cons1"], exp["cons2"])
        if name == "hemizygous":
            assert call.allele1 == -2 and same(call.allele2, pick[0]) and not call.is_dual
        else:
            got = sorted([call.allele1, call.allele2])
            assert call.is_dual and call.dual_passed and (all(same(x, y) for x, y in zip(got, sorted(pick))) or all(same(x, y) for x, y in zip(got, sorted(pick)[::-1])))


def test_cohort_equals_sample_by_sample(oracle, pkg, gpu_ctx):
    """BASELINE configs[4] in miniature: several WGS-sized samples in one read set, one realignment call, every (sample, gene)
    consensus problem in lockstep (sp_hla_diplotype_cohort) -- the same calls and consensuses as one sample at a time, and the truth"""
    from pb_starphase_amd import synth
    fx = synth.HlaFixture()
    db = fx.make_db(pkg, gpu_ctx)
    rng = np.random.default_rng(99)
    n_samples = 5
    reads, sample_of, truth = [], [], []
    for s in range(n_samples):
        t = {}
        for g in range(len(fx.genes)):
            pick = rng.choice(fx.full_length_alleles(g), 2, replace=False).tolist()
            if s == 3 and g == 1:
                pick = [pick[0], pick[0]]                          # one homozygous gene
            t[g] = sorted(pick)
            for a in pick:
                hap, st = fx.haplotype(g, a)
                rs = synth.simulate_reads(rng, hap, st, len(fx.dna[a]), 20, mean_len=7000, sd_len=1500, min_overlap=2500)
                reads += rs; sample_of += [s] * len(rs)
        truth.append(t)
    order = rng.permutation(len(reads))
    reads, sample_of = [reads[i] for i in order], [sample_of[i] for i in order]
    R = gpu_ctx.upload(reads)
    k1 = db.realign_reads(R)
    genes = list(range(len(fx.genes)))
    cohort, is1 = db.diplotype_cohort(n_samples, sample_of, genes, R, k1)
    same = lambda a, b: a == b or (a >= 0 and b >= 0 and fx.cdna[a] == fx.cdna[b] and fx.dna[a] == fx.dna[b])
    for s in range(n_samples):
        mine = [i for i, x in enumerate(sample_of) if x == s]
        Rs = gpu_ctx.upload([reads[i] for i in mine])
        alone, is1_s = db.diplotype_genes(genes, Rs, db.realign_reads(Rs))
        for g in genes:
            c, c1, c2 = cohort[s][g]
            a, a1, a2 = alone[g]
            assert (c1, c2) == (a1, a2) and bytes(c) == bytes(a)
            got, want = sorted([c.allele1, c.allele2]), truth[s][g]
            assert all(same(x, y) for x, y in zip(got, want)) or all(same(x, y) for x, y in zip(got, want[::-1])), (s, g, got, want)
        assert is1[mine].tolist() == is1_s.tolist()


def test_hla_debug_file(oracle, pkg, gpu_ctx, tmp_path):
    """hla_debug.json (HlaDebug, /root/reference/src/hla/debug.rs:7-221) of one sample: the DualPassingStats of every gene are the call's, and
    the consensus entries carry DetailedMappingStats built from the library's alignment -- CIGAR / MD equal to the ones derived from the
    oracle's alignment of the same pair, and consistent with the sequences"""
    import json
    import re
    from pb_starphase_amd import synth
    D = pkg.database
    fx = synth.HlaFixture(max_alleles_per_gene=150, seed=4)
    db = fx.make_db(pkg, gpu_ctx)
    rng = np.random.default_rng(21)
    reads, truth = [], {}
    for g in range(len(fx.genes)):
        truth[g] = rng.choice(fx.full_length_alleles(g), 2, replace=False).tolist()
        reads += simulate(fx, synth, rng, g, truth[g], 14)
    R = gpu_ctx.upload(reads)
    k1 = db.realign_reads(R)
    dbg = D.HlaDebug()
    expected_dual = {}
    for g, gene in enumerate(fx.genes):
        call, c1, c2, _is1 = db.diplotype_gene(g, R, k1)
        assert call.status == 0 and call.is_dual
        dbg.add_dual_stats(gene, call)
        expected_dual[gene] = {"is_passing": bool(call.dual_passed), "is_dual": True, "counts1": call.counts1, "counts2": call.counts2,
                               "maf": call.maf, "cdf": call.cdf}
        for name, cons, typed in (("consensus1", c1, call.typed1), ("consensus2", c2, call.typed2)):
            hap, _s = fx.haplotype(g, typed)                                  # hg38-forward haplotype that carries the typed allele
            qs, ts = gpu_ctx.upload([cons]), gpu_ctx.upload([hap])
            diag, votes = gpu_ctx.anchor_batch(qs, ts, [0], [0])
            aln, ev = gpu_ctx.align_batch(qs, ts, [0], [0], diag, [255], events=True)
            assert votes[0] > 0 and aln[0]["ok"]
            m = D.detailed_mapping(aln[0], ev[0], hap)
            # the same strings from the oracle's alignment of the pair
            o_al, o_ev = oracle.wfa(cons, hap, int(diag[0]), 255)
            assert (o_al.nm, o_al.b_start, o_al.b_end) == (aln[0]["nm"], aln[0]["b_start"], aln[0]["b_end"])
            merged = []
            for n, op in oracle.cigar(o_al, o_ev):
                op = "M" if op in (7, 8) else "I" if op == 1 else "D"
                if merged and merged[-1][1] == op:
                    merged[-1][0] += n
                else:
                    merged.append([n, op])
            assert m["cigar"] == "".join(f"{n}{op}" for n, op in merged)
            ops = [(int(n), op) for n, op in re.findall(r"(\d+)([MID])", m["cigar"])]
            assert sum(n for n, op in ops if op in "MI") == aln[0]["a_end"] - aln[0]["a_start"]
            assert sum(n for n, op in ops if op in "MD") == aln[0]["b_end"] - aln[0]["b_start"]
            md_ref = sum(int(x) if x.isdigit() else len(x.lstrip("^")) for x in re.findall(r"\d+|\^[ACGT]+|[ACGT]", m["md"]))
            assert md_ref == aln[0]["b_end"] - aln[0]["b_start"]
            assert m["match_len"] == sum(int(x) for x in re.findall(r"\d+", m["md"])) and m["query_len"] == len(cons) and m["target_len"] == len(hap)
            dbg.add_read(gene, name, fx.ids[typed] if hasattr(fx, "ids") else f"allele{typed}", fx.names[typed] if hasattr(fx, "names") else str(typed))
            dbg.add_mapping(gene, name, f"allele{typed}", cdna=None, dna=m)
    out = tmp_path / "hla_debug.json"
    dbg.save(str(out))
    obj = json.load(open(out))
    assert obj["dual_passing_stats"] == expected_dual                        # f64 text round-trips to the same doubles
    assert list(obj["read_mapping_stats"]) == sorted(fx.genes)
    for gene in fx.genes:
        assert list(obj["read_mapping_stats"][gene]) == ["consensus1", "consensus2"]
        for entry in obj["read_mapping_stats"][gene].values():
            (stats,) = entry["mapping_stats"].values()
            assert stats["cdna_mapping"] is None and list(stats["dna_mapping"]) == ["query_len", "target_len", "match_len", "nm", "query_unmapped",
                                                                                  "target_unmapped", "cigar", "md"]


def test_genes_side_by_side_equal_one_stream(pkg, gpu_ctx):
    """sp_hla_diplotype_genes on a sample of >= 1,000 realigned reads solves the genes on two streams (sp_ctx_set_option "hla_split_genes");
    calls, consensuses and read groups are those of the one-stream run"""
    from pb_starphase_amd import synth
    fx = synth.HlaFixture(max_alleles_per_gene=150, seed=4)
    db = fx.make_db(pkg, gpu_ctx)
    wl = synth.Config2Workload(fx, n_reads=1400, seed=9)
    R = gpu_ctx.upload(wl.reads)
    k1 = db.realign_reads(R)
    assert int((k1["status"] == 0).sum()) >= 1000
    genes = list(range(len(fx.genes)))
    runs = []
    for split in (1, 0, 1):
        gpu_ctx.set_option("hla_split_genes", split)
        calls, is1 = db.diplotype_genes(genes, R, k1)
        runs.append(([(c.status, c.allele1, c.allele2, c.typed1, c.typed2, c.n_reads, c.counts1, c.counts2, c.is_dual, c.dual_passed, c.used_dna_dual, c.maf, c.cdf, c1, c2)
                      for c, c1, c2 in calls], is1.tolist()))
    gpu_ctx.set_option("hla_split_genes", 1)
    assert runs[0] == runs[1] == runs[2]
    assert all(r[0] == 0 for r in runs[0][0])
    truth = {g: sorted(a for gg, a in wl.truth if gg == g) for g in genes}
    for g in genes:
        assert sorted(runs[0][0][g][1:3]) == truth[g]
    # the same reads as a cohort of two samples (every other read): (sample, gene) units side by side == on one stream
    sample_of = np.arange(len(wl.reads)) % 2
    cohort_runs = []
    for split in (1, 0):
        gpu_ctx.set_option("hla_split_genes", split)
        cohort, is1 = db.diplotype_cohort(2, sample_of, genes, R, k1)
        cohort_runs.append(([[(c.status, c.allele1, c.allele2, c.typed1, c.typed2, c.n_reads, c.counts1, c.counts2, c.is_dual, c.dual_passed, c1, c2) for c, c1, c2 in row]
                             for row in cohort], is1.tolist()))
    gpu_ctx.set_option("hla_split_genes", 1)
    assert cohort_runs[0] == cohort_runs[1]
    assert all(u[0] == 0 and sorted(u[1:3]) == truth[g] for row in cohort_runs[0][0] for g, u in enumerate(row))
    with pytest.raises(pkg.StarphaseError):
        gpu_ctx.set_option("no_such_switch", 1)
