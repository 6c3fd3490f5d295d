"""BASELINE configs[3] (full CPIC panel, one 30x-WGS-like sample) and configs[1] at its stated size, through the C ABI on the GPU.

configs[3]: every variant gene of the bundled v0.14.1 database (gene_entries: 18 CPIC / PharmVar genes, up to 340 haplotypes x 314
variants) gets synthetic VCF observations drawn from two of its own haplotypes (hom / het, phased / unphased mixes, at most 8 hets), the
K6 search (sp_variant_solve) must equal oracle/variant.c cell for cell and the call must contain the simulated pair; the same sample
carries HLA-A/-B (~45 reads per gene, full IMGT/HLA database) and CYP2D6 (~100 reads, real 39 templates / variant table).
configs[1]: 10,000 reads; the pruned K1 search equals the exhaustive one on 1,000 of them and the oracle's whole-read search on 64."""
import ctypes as C
import gzip
import json
import os

import numpy as np
import pytest

import cyp_cases_real as cr
import variant_glue as vg

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def synthetic_observations(rng, haps, max_hets=8):
    """observed variants of a sample that carries haplotypes h1 / h2 of the gene: the first non-None alternative of every slot"""
    usable = [h for h in haps if all(any(nv is not None for nv in s) for s in h["slots"])]
    for _ in range(50):
        h1, h2 = rng.choice(len(usable), 2, replace=True).tolist()
        v1 = {next(nv for nv in s if nv is not None) for s in usable[h1]["slots"]}
        v2 = {next(nv for nv in s if nv is not None) for s in usable[h2]["slots"]}
        hets = sorted((v1 | v2) - (v1 & v2))
        if len(hets) <= max_hets:
            break
    obs = {v: (4, None) for v in v1 & v2}
    phased = rng.random() < 0.6
    ps = int(rng.integers(1000, 2000))
    for v in hets:
        if phased and rng.random() < 0.85:
            obs[v] = (2 if v in v2 else 3, ps)             # 0|1 when on the second haplotype, 1|0 on the first
        else:
            obs[v] = (1, None)
    return usable[h1]["name"], usable[h2]["name"], obs, phased and all(obs[v][0] != 1 for v in hets)


def test_full_panel_sample(oracle, pkg, gpu_ctx):
    from pb_starphase_amd import synth
    from test_gpu_variant import gpu_struct
    rng = np.random.default_rng(30)
    genes = json.load(gzip.open(os.path.join(GOLDEN, "gene_entries_v0.14.1.json.gz")))["gene_entries"]
    assert len(genes) == 18
    n_cells = 0
    for name in sorted(genes):
        gene = genes[name]
        vh, haps = vg.load_database_haplotypes(oracle, gene, None)
        assert len(haps) >= 2
        for rep in range(3):
            h1, h2, obs, fully_phased = synthetic_observations(rng, haps)
            prob = vg.Problem(vh, haps, obs, gene.get("structural_variants"))
            exp = vg.oracle_solve(oracle, prob)
            got = gpu_ctx.variant_solve(gpu_struct(pkg, prob))
            assert got == exp, (name, rep)
            called = vg.call_gene(oracle, prob, solver=lambda pr: gpu_ctx.variant_solve(gpu_struct(pkg, pr)))
            if fully_phased or h1 == h2:
                pairs = [frozenset(d) for d in called["diplotypes"]]
                assert exp[0] == (0, 0, 0, 0) and frozenset((h1, h2)) in pairs, (name, rep, h1, h2, called["diplotypes"][:4])
            n_cells += len(haps)
    assert n_cells > 2500
    # ---- HLA-A / -B of the same sample
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
    calls, _ = db.diplotype_genes(list(range(len(fx.genes))), R, db.realign_reads(R))
    same = lambda a, b: a == b or (a >= 0 and b >= 0 and fx.cdna[a] == fx.cdna[b] and fx.dna[a] == fx.dna[b])
    for g, (call, _c1, _c2) in enumerate(calls):
        got = sorted([call.allele1, call.allele2])
        assert call.status == 0 and call.is_dual and call.dual_passed
        assert all(same(a, b) for a, b in zip(got, truth[g])) or all(same(a, b) for a, b in zip(got, truth[g][::-1])), (g, got, truth[g])
    # ---- CYP2D6 of the same sample: ~100 reads on the real-shape locus
    cfg, gene_def = cr.load_db()
    locus = synth.Chr22Locus(cfg, gene_def, seed=3)
    cdb = pkg.ffi.CypDb(gpu_ctx, cfg, gene_def, locus.sequence, locus.start)
    name, haps, expected = cr.scenarios(locus)[0]
    creads = locus.sample(rng, haps, 100, lo=8000, hi=16000)
    call, _cons, _labels = cdb.diplotype(gpu_ctx.upload(creads))
    assert call.status == 0 and sorted([call.hap1.decode(), call.hap2.decode()]) == sorted(expected)


def test_config1_stated_size(oracle, pkg, gpu_ctx, k1_exhaustive):
    from pb_starphase_amd import synth
    fx = synth.HlaFixture()
    wl = synth.Config2Workload(fx, n_reads=10000, seed=1000)
    db = fx.make_db(pkg, gpu_ctx)
    R = gpu_ctx.upload(wl.reads)
    out = db.realign_reads(R)
    assert float(np.mean(out["status"] == 0)) == 1.0
    assert all(out[r]["gene"] == wl.read_truth[r][0] for r in range(len(wl.reads)))
    # pruned == exhaustive on 1,000 reads (the exhaustive pass also hands back its cell matrix)
    pick = np.sort(np.random.default_rng(1).choice(len(wl.reads), 1000, replace=False))
    sub = gpu_ctx.upload([wl.reads[i] for i in pick])
    full, cells = db.realign_reads(sub, cells=True)
    assert full.tolist() == db.realign_reads(sub).tolist()
    assert full.tolist() == out[pick].tolist()
    assert int((cells != 0xFFFFFFFF).sum()) > 1000
    # 64 reads against the oracle's whole-read search (osp_hla_k1_read: anchor, every allele cell, acceptance)
    import hla_expected as hx
    tb = hx.K1Tables(oracle, fx)
    L = oracle.L
    L.osp_hla_k1_read.restype = C.c_int32
    refs = tb.refs
    n_all = len(fx.ids)
    enc = [e if e is not None else np.zeros(0, np.uint8) for e in tb.fwd_e]
    off = np.array([(-2 ** 31 if o is None else o) for o in tb.off], np.int32)
    ref_ptr = (C.c_void_p * len(refs))(*[r.ctypes.data for r in refs]); ref_len = np.array([len(r) for r in refs], np.int32)
    al_ptr = (C.c_void_p * n_all)(*[(e.ctypes.data if len(e) else None) for e in enc]); al_len = np.array([len(e) for e in enc], np.int32)
    gene_of = fx.gene_of.astype(np.int32)
    for r in pick[:64]:
        re = oracle.encode(wl.reads[r])
        ncell = C.c_int64(0)
        b = L.osp_hla_k1_read(re.ctypes.data_as(C.c_void_p), len(re), len(refs), ref_ptr, ref_len.ctypes.data_as(C.c_void_p), n_all, al_ptr,
                              al_len.ctypes.data_as(C.c_void_p), gene_of.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), None, C.byref(ncell))
        assert b == int(out[r]["best_allele"]), r
    # all 1,000 of them against what that search found on the CPU when the fixture was made (tests/golden/make_fullsize.py: eight CPU-minutes)
    import gzip, json, os
    gold = json.load(gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize_oracle.json.gz"), "rt"))["k1"]
    assert gold["reads"] == pick.tolist()
    assert [int(out[r]["best_allele"]) for r in pick] == gold["best_allele"]
    # reads -> diplotype at the stated size equals the simulated truth
    calls, _ = db.diplotype_genes(list(range(len(fx.genes))), R, out)
    truth = {g: sorted(a for (gg, _c, _d, a) in wl.consensus if gg == g) for g in range(len(fx.genes))}
    same = lambda a, b: a == b or (a >= 0 and b >= 0 and fx.cdna[a] == fx.cdna[b] and fx.dna[a] == fx.dna[b])
    for g, (call, _c1, _c2) in enumerate(calls):
        got = sorted([call.allele1, call.allele2])
        assert all(same(a, b) for a, b in zip(got, truth[g])) or all(same(a, b) for a, b in zip(got, truth[g][::-1]))


def test_sample_from_database_files_to_result_file(oracle, pkg, gpu_ctx, tmp_path):
    """The f3 route end to end: the database FILES are read and flattened by the library (sp_database_*, sp_variant_gene_*), the calls run
    on the GPU, the result goes out through sp_result_*.  Every call equals the one made from the Python-flattened inputs / the oracle."""
    from pb_starphase_amd import synth
    D = pkg.database
    rng = np.random.default_rng(31)
    result = D.Result(D.Database(os.path.join(GOLDEN, "gene_entries_v0.14.1.json.gz")), "test")
    # ---- variant genes: problem built by the library from VCF-like alleles, solved on the GPU == oracle on the glue's problem
    path = os.path.join(GOLDEN, "gene_entries_v0.14.1.json.gz")
    raw = json.load(gzip.open(path))["gene_entries"]
    db = D.Database(path)
    for name, _chrom in db.gene_entries():
        vh, haps = vg.load_database_haplotypes(oracle, raw[name], None)
        h1, h2, obs, _phased = synthetic_observations(rng, haps)
        want = vg.Problem(vh, haps, obs, raw[name].get("structural_variants"))
        gene = db.variant_gene(name)
        alleles = [(v[1], v[2], v[3], gt, ps) for v, (gt, ps) in sorted(obs.items())]
        rng.shuffle(alleles)                                                    # record order does not matter
        prob = gene.problem(alleles)
        res = pkg.ffi.sp_variant_result()
        gpu_ctx.check(pkg.ffi.lib().sp_variant_solve(gpu_ctx._h, C.byref(prob), C.byref(res)))
        got = (tuple(res.score), [(res.dip[i][0], res.dip[i][1], res.dip_comb[i]) for i in range(res.n_dip)])
        assert got == vg.oracle_solve(oracle, want), name
        hap_names = [h[0] for h in gene.haplotypes()]
        d = D.GeneDetails()
        for a, b, _c in got[1][:4]:
            d.add_diplotype(hap_names[a], hap_names[b])
        meta = gene.variants()
        arr = D.problem_arrays(prob)
        for o in range(prob.n_obs):
            k, _label, _s, _e = gene.problem_variant(arr["obs_var"][o])
            m = meta[k]
            d.add_variant(m["variant_id"], m["name"], m["dbsnp_id"], raw[name]["chromosome"], m["position"], m["ref"], m["alt"], arr["obs_gt"][o],
                          None if arr["obs_ps"][o] < 0 else arr["obs_ps"][o], m["is_core_variant"])
        result.insert(name, d, D.SUBALLELE_MATCH if got[0] == (0, 0, 0, 0) else D.INEXACT_DIPLOTYPES)
    # ---- HLA: database file -> sp_hla_db, same calls as from the Python-flattened description
    fx = synth.HlaFixture()
    hdb_file = D.Database(os.path.join(GOLDEN, "hla_db_v0.14.1.json.gz"))
    regions = hdb_file.hla_genes()
    assert [(r["start"], r["end"]) for r in regions] == fx.coords
    # the reference bases come out of a FASTA file through its .fai: chr6 with the gene islands, chr22 with the CYP2D6 window (N elsewhere)
    cfg, gene_def = cr.load_db()
    locus = synth.Chr22Locus(cfg, gene_def, seed=3)
    buffer = (len(fx.gene_ref[0]) - (fx.coords[0][1] - fx.coords[0][0])) // 2
    chr6 = bytearray(b"N" * (max(e for _s, e in fx.coords) + buffer + 1000))
    for (s0, _e0), ref in zip(fx.coords, fx.gene_ref):
        chr6[s0 - buffer:s0 - buffer + len(ref)] = ref.encode()
    chr22 = bytearray(b"N" * (locus.start + len(locus.sequence) + 1000))
    chr22[locus.start:locus.start + len(locus.sequence)] = locus.sequence.encode()
    import test_io
    fasta_path = tmp_path / "reference.fa"
    test_io.write_fasta(fasta_path, {"chr6": chr6.decode(), "chr22": chr22.decode()}, 60, index=True)
    del chr6, chr22
    fasta = D.Fasta(str(fasta_path))
    gene_ref = [fasta.fetch(r["chrom"], r["start"] - buffer, r["end"] + buffer) for r in regions]
    assert gene_ref == fx.gene_ref
    hdb, alleles = hdb_file.hla_db(gpu_ctx, gene_ref)
    ref_db = fx.make_db(pkg, gpu_ctx)
    reads = []
    for g in range(len(fx.genes)):
        for a in rng.choice(fx.full_length_alleles(g), 2, replace=False).tolist():
            hap, s = fx.haplotype(g, a)
            reads += synth.simulate_reads(rng, hap, s, len(fx.dna[a]), 23, mean_len=6000, sd_len=1500, min_overlap=2500)
    # the reads come out of a BAM file (written by the test's own encoder): one region fetch per gene, QNAMEs handed out once
    order = rng.permutation(len(reads)).tolist()
    recs = sorted(((0, int(regions[i % 2]["start"]) + int(rng.integers(0, 2000)), f"m84/{i}/ccs", 0, 60, [("M", len(reads[i]))], reads[i]) for i in order),
                  key=lambda r: r[1])
    bam_path = str(tmp_path / "sample.bam")
    test_io.write_bam(bam_path, [("chr6", 170805979)], recs, 65280)
    bam = D.Bam(bam_path)
    fetched = [r for g in regions for r in bam.fetch(g["chrom"], g["start"], g["end"], exclude_flags=0x900, dedupe=True)]
    assert sorted(r["qname"] for r in fetched) == sorted(r[2] for r in recs)
    by_name = {r[2]: r[6] for r in recs}
    assert all(r["seq"] == by_name[r["qname"]] for r in fetched)
    reads = [r["seq"] for r in fetched]
    R = gpu_ctx.upload(reads)
    ra, rb = hdb.realign_reads(R), ref_db.realign_reads(R)
    assert ra.tolist() == rb.tolist()
    ca, _ = hdb.diplotype_genes(list(range(len(fx.genes))), R, ra)
    cb, _ = ref_db.diplotype_genes(list(range(len(fx.genes))), R, rb)
    for g, ((x, x1, x2), (y, y1, y2)) in enumerate(zip(ca, cb)):
        assert (x.status, x.allele1, x.allele2, x1, x2) == (y.status, y.allele1, y.allele2, y1, y2) and x.status == 0
        d = D.GeneDetails().add_diplotype(alleles[x.allele1][2], alleles[x.allele2][2])
        for r in np.nonzero(ra["gene"] == g)[0][:5]:
            d.add_mapping(f"read/{r}", alleles[ra[r]["best_allele"]][0], alleles[ra[r]["best_allele"]][2], cdna=None,
                          dna=(int(ra[r]["target_len"]), int(ra[r]["nm"]), int(ra[r]["unmapped"])))
        result.insert(fx.genes[g], d, D.FROM_MAPPINGS)
    # ---- CYP2D6: database file -> sp_cyp_db -> sp_cyp_diplotype
    cyp_file = D.Database(os.path.join(GOLDEN, "cyp2d6_db_v0.14.1.json.gz"))
    w_chrom, w_start, w_end = cyp_file.cyp_window()
    assert w_chrom == "chr22" and locus.start <= w_start and w_end <= locus.start + len(locus.sequence)
    cdb = cyp_file.cyp_db(gpu_ctx, fasta.fetch(w_chrom, locus.start, locus.start + len(locus.sequence)), locus.start)
    _name, haps, expected = cr.scenarios(locus)[1]
    call, _cons, _labels = cdb.diplotype(gpu_ctx.upload(locus.sample(rng, haps, 120, lo=8000, hi=16000)))
    assert call.status == 0 and sorted([call.hap1.decode(), call.hap2.decode()]) == sorted(expected)
    # the entry as diplotype_cyp2d6 builds it (src/cyp2d6/caller.rs:709-737): sub-allele diplotype, collapsed one, the deep one without haplotypes
    d6 = D.GeneDetails().add_diplotype(call.hap1.decode(), call.hap2.decode()).add_simple_diplotype(call.core1.decode(), call.core2.decode()) \
        .add_diplotype_only(call.deep1.decode(), call.deep2.decode())
    result.insert("CYP2D6", d6, D.FROM_MULTI_MAPPINGS)
    out = tmp_path / "sample.json.gz"
    result.save(str(out))
    obj = json.load(gzip.open(out))
    assert sorted(obj["gene_details"]) == sorted(list(raw) + fx.genes + ["CYP2D6"]) and len(obj["gene_details"]) == 21
    assert obj["gene_details"]["CYP2D6"]["diplotypes"][0]["diplotype"] == f"{call.hap1.decode()}/{call.hap2.decode()}"
    assert obj["gene_details"]["CYP2D6"]["simple_diplotypes"] == [{"hap1": "*4", "hap2": "*4", "diplotype": "*4/*4"}]
    deep = obj["gene_details"]["CYP2D6"]["inexact_diplotypes"]
    assert len(deep) == 1 and deep[0]["haplotype_1"] is None and deep[0]["haplotype_2"] is None
    assert sorted(h.split("_", 1)[1] for h in (deep[0]["basic_diplotype"]["hap1"], deep[0]["basic_diplotype"]["hap2"])) == ["CYP2D6*4.001)"] * 2
