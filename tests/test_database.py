"""SURVEY.md 8 row f3 through the C ABI (host only, no GPU): the database file -> the kernels' inputs, and the result file.

* sp_database_load / _parse on the bundled v0.14.1 / v0.9.0 data and the reference's small test databases, against Python's json;
* sp_database_hla_flatten against the Python flattening the GPU tests have used so far (synth.HlaFixture);
* sp_database_cyp_flatten + sp_cyp_db_create against the tables built from the Python-flattened structs, and the reference vector of
  test_load_variant_database (src/cyp2d6/haplotyper.rs:918-933) on a file without cyp2d6_config (the defaults apply);
* sp_variant_gene_* against the test-side restatement of load_database_haplotypes / load_vcf_variants / load_sv_vcf_variants
  (tests/variant_glue.py) on every scenario of the reference's diplotyper tests (src/diplotyper.rs:1653-2304), and on every CPIC gene of
  the bundled database;
* sp_result_* : the constructor tests of src/data_types/starphase_json.rs:327-489, and the text against Python's json layout (which is
  serde_json's pretty layout: two-space indent, ": " and ",\\n" separators, [] / {} for empty containers)."""
import gzip
import json
import os

import numpy as np
import pytest

import variant_glue as vg
from test_oracle_variant import CASES, SV_CASES

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GZ = ["hla_db_v0.14.1.json.gz", "cyp2d6_db_v0.14.1.json.gz", "cyp2d6_gene_def_v0.9.0.json.gz", "gene_entries_v0.14.1.json.gz"]


@pytest.fixture(scope="module")
def D(pkg):
    return pkg.database


@pytest.mark.parametrize("name", GZ)
def test_load_bundled_data(D, name):
    path = os.path.join(GOLDEN, name)
    want = json.load(gzip.open(path))
    db = D.Database(path)
    st = db.stats
    assert st.n_gene_entries == len(want.get("gene_entries", {}))
    assert st.n_hla_sequences == len(want.get("hla_sequences", {}))
    assert st.n_cyp2d6_alleles == len(want.get("cyp2d6_gene_def", {}))
    assert (st.has_hla_config, st.has_cyp2d6_config) == (int("hla_config" in want), int("cyp2d6_config" in want))
    assert db.metadata == want["database_metadata"]
    assert db.gene_entries() == [(g, want["gene_entries"][g]["chromosome"]) for g in sorted(want.get("gene_entries", {}))]
    # the same bytes, unzipped, through sp_database_parse
    db2 = D.Database(gzip.open(path).read())
    assert db2.metadata == db.metadata and db2.stats.n_hla_sequences == st.n_hla_sequences


def test_load_errors(D, pkg, tmp_path):
    with pytest.raises(pkg.StarphaseError, match="cannot open"):
        D.Database(str(tmp_path / "missing.json"))
    with pytest.raises(pkg.StarphaseError, match="JSON"):
        D.Database(b'{"database_metadata": {')
    with pytest.raises(pkg.StarphaseError, match="database_metadata"):
        D.Database(b'{"gene_entries": {}}')
    with pytest.raises(pkg.StarphaseError, match="256 levels"):
        D.Database(b'{"database_metadata": ' + b"[" * 400 + b"]" * 400 + b"}")
    bad = gzip.compress(b'{"database_metadata": {}}')[:-9]
    with pytest.raises(pkg.StarphaseError, match="gzip"):
        D.Database(bad)
    p = tmp_path / "mini.json"
    p.write_text(json.dumps({"database_metadata": {k: "x" for k in ("pbstarphase_version", "cpic_version", "hla_version", "pharmvar_version", "build_time")},
                             "gene_entries": {}, "hla_sequences": {}, "cyp2d6_gene_def": {}}))
    db = D.Database(str(p))
    # neither configuration in the file: HlaConfig / Cyp2d6Config defaults (src/hla/alleles.rs:232-318, src/cyp2d6/definitions.rs:128-301)
    assert [g["name"] for g in db.hla_genes()] == ["HLA-A", "HLA-B"]
    assert db.hla_genes()[0]["start"] == 29942254 - 1 and db.hla_genes()[1]["is_forward_strand"] is False
    assert db.cyp_window() == ("chr22", 42123192 - 1, 42145903)


def test_hla_config_both_schemas(D):
    old = json.load(gzip.open(os.path.join(GOLDEN, "hla_db_v0.14.1.json.gz")))
    cfg = old["hla_config"]
    md = old["database_metadata"]
    gene_dict = {g: {"gene_name": g, "coordinates": cfg["hla_coordinates"][g], "is_forward_strand": cfg["hla_is_forward_strand"][g],
                     "transcript_id": None, "exons": cfg["hla_exons"][g], "is_absent_capable": g == "HLA-B"} for g in cfg["hla_coordinates"]}
    new = {"database_metadata": md, "gene_entries": {}, "hla_sequences": {}, "cyp2d6_gene_def": {},
           "hla_config": {"gene_collection": {"version": "t", "gene_dict": gene_dict}}}
    a = D.Database(json.dumps({"database_metadata": md, "gene_entries": {}, "hla_sequences": {}, "cyp2d6_gene_def": {}, "hla_config": cfg}).encode()).hla_genes()
    b = D.Database(json.dumps(new).encode()).hla_genes()
    assert [g["is_absent_capable"] for g in b] == [False, True]
    for g in b:
        g["is_absent_capable"] = False
    assert a == b and len(a) == 2 and len(a[0]["exons"]) == 8
    # validate_config (src/hla/alleles.rs:84-102)
    gene_dict["HLA-A"]["exons"] = []
    with pytest.raises(Exception, match="Found 0 exons"):
        D.Database(json.dumps(new).encode())


def _strings(blob_ptr, off_ptr, n):
    import ctypes as C
    off = np.ctypeslib.as_array(C.cast(off_ptr, C.POINTER(C.c_uint64)), (n + 1,))
    blob = C.string_at(blob_ptr, int(off[-1]))
    return [blob[off[i]:off[i + 1]].decode() for i in range(n)]


@pytest.mark.parametrize("genes", [None, ["HLA-B"], ["HLA-B", "HLA-A"]])
def test_hla_flatten_equals_python_flattening(D, pkg, genes):
    import ctypes as C
    from pb_starphase_amd import synth
    fx = synth.HlaFixture(genes=genes)
    db = D.Database(os.path.join(GOLDEN, "hla_db_v0.14.1.json.gz"))
    desc, alleles = db.hla_flatten(fx.gene_ref, genes=genes, ref_buffer=fx.buffer)
    assert desc.n_alleles == len(fx.ids) and desc.n_genes == len(fx.genes)
    assert [a[0] for a in alleles] == fx.ids and [a[2] for a in alleles] == fx.star
    assert [a[1] for a in alleles] == [fx.genes[g] for g in fx.gene_of]
    as_arr = lambda ptr, n, t: np.ctypeslib.as_array(C.cast(ptr, C.POINTER(t)), (n,))
    assert (as_arr(desc.gene_of, desc.n_alleles, C.c_uint32) == fx.gene_of).all()
    assert _strings(desc.dna, desc.dna_off, desc.n_alleles) == fx.dna
    assert _strings(desc.cdna, desc.cdna_off, desc.n_alleles) == fx.cdna
    assert _strings(desc.gene_ref, desc.gene_ref_off, desc.n_genes) == fx.gene_ref
    assert list(as_arr(desc.gene_fwd, desc.n_genes, C.c_uint8)) == fx.gene_fwd
    eo = as_arr(desc.exon_off, desc.n_genes + 1, C.c_uint32)
    es, ee = as_arr(desc.exon_start, eo[-1], C.c_int32), as_arr(desc.exon_end, eo[-1], C.c_int32)
    assert [[(int(es[k]), int(ee[k])) for k in range(eo[g], eo[g + 1])] for g in range(desc.n_genes)] == fx.exons
    assert desc.ref_buffer == fx.buffer
    with pytest.raises(pkg.StarphaseError, match="must hold"):
        db.hla_flatten([r[:-1] for r in fx.gene_ref], genes=genes, ref_buffer=fx.buffer)
    with pytest.raises(pkg.StarphaseError, match="no gene"):
        db.hla_flatten(fx.gene_ref[:1], genes=["HLA-Q"], ref_buffer=fx.buffer)


@pytest.mark.parametrize("name", ["cyp2d6_db_v0.14.1.json.gz", "cyp2d6_gene_def_v0.9.0.json.gz"])
def test_cyp_flatten_equals_python_flattening(D, pkg, name):
    from pb_starphase_amd import synth
    cfg = json.load(gzip.open(os.path.join(GOLDEN, "cyp2d6_db_v0.14.1.json.gz")))["cyp2d6_config"]
    gene_def = json.load(gzip.open(os.path.join(GOLDEN, name)))["cyp2d6_gene_def"]
    locus = synth.Chr22Locus(cfg, gene_def, seed=5)
    db = D.Database(os.path.join(GOLDEN, name))
    chrom, lo, hi = db.cyp_window()
    assert chrom == "chr22" and locus.start <= lo and hi <= locus.start + len(locus.sequence)
    got = db.cyp_db(None, locus.sequence, locus.start)
    want = pkg.ffi.CypDb(None, cfg, gene_def, locus.sequence, locus.start)
    assert got.templates() == want.templates() and len(got.templates()) == 39
    assert got.variants() == want.variants()
    ga, wa = got.alleles(), want.alleles()
    assert ga[0] == wa[0] and (ga[1] == wa[1]).all()
    assert got.cfg == want.cfg
    if "v0.9.0" in name:                # test_load_variant_database (src/cyp2d6/haplotyper.rs:918-933), cyp2d6_config from the defaults
        st = got.stats
        assert (st.first_variant_pos, st.last_variant_pos, st.n_variants, st.n_vi) == (42126309, 42132374, 387, 144)
        assert got.index_label("rs12169962") == 0 and got.index_label("rs1080985") == 386


# ------------------------------------------------------------------ variant-typed genes
def vcf_alleles(vcf):
    """decoded rows -> one (position0, ref, alt, SP_GT_*, ps) per ALT allele (what a VCF reader in front of the library hands over)"""
    sample = [c for c in vcf["columns"] if c not in ("CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO", "FORMAT")][0]
    out = []
    for row in vcf["rows"]:
        fmt = dict(zip(row["FORMAT"].split(":"), row[sample].split(":")))
        gt = fmt["GT"]
        a = gt.replace("|", "/").split("/")
        if len(a) != 2 or "." in a:
            continue
        g1, g2 = int(a[0]), int(a[1])
        phased = "|" in gt
        ps = int(fmt["PS"]) if phased and fmt.get("PS", ".") != "." else None
        for ai, alt in enumerate(row["ALT"].split(","), start=1):
            if ai == g1 and ai == g2:
                code = 4
            elif ai == g1 and phased:
                code = 3
            elif ai == g2 and phased:
                code = 2
            elif ai in (g1, g2):
                code = 1
            else:
                code = 0
            out.append((int(row["POS"]) - 1, row["REF"], alt, code, ps))
    return out


def vcf_deletions(vcf):
    sample = [c for c in vcf["columns"] if c not in ("CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO", "FORMAT")][0]
    out = []
    for row in vcf["rows"]:
        info = dict(kv.split("=", 1) for kv in row["INFO"].split(";") if "=" in kv)
        if len(row["ALT"].split(",")) != 1 or info.get("SVTYPE") != "DEL":
            continue
        fmt = dict(zip(row["FORMAT"].split(":"), row[sample].split(":")))
        gt = fmt["GT"]
        a = gt.replace("|", "/").split("/")
        if len(a) != 2 or "." in a:
            continue
        g1, g2 = int(a[0]), int(a[1])
        phased = "|" in gt
        ps = int(fmt["PS"]) if phased and fmt.get("PS", ".") != "." else None
        code = (0 if g1 == 0 else 4) if g1 == g2 else ((2 if g1 == 0 else 3) if phased else 1)
        out.append((int(row["POS"]) - 1, int(info["END"]), code, ps))
    return out


def check_problem(D, gene, got, want):
    """got: sp_variant_problem from the library; want: variant_glue.Problem"""
    arr = D.problem_arrays(got)
    n_obs = len(want.obs)
    assert arr["n_haps"] == len(want.haps) and arr["n_vars"] == len(want.var_list) and arr["n_obs"] == n_obs
    assert arr["hap_is_sv"] == want.hap_is_sv.tolist() and arr["hap_is_core"] == want.hap_is_core.tolist()
    assert arr["slot_off"] == want.slot_off.tolist() and arr["alt_off"] == want.alt_off.tolist()
    assert arr["alt_var"] == want.alt_var.tolist()[:arr["alt_off"][-1]]
    assert arr["var_is_core"] == want.var_is_core.tolist()[:len(want.var_list)]
    assert arr["obs_var"] == want.obs_var.tolist()[:n_obs] and arr["obs_gt"] == want.obs_gt.tolist()[:n_obs]
    assert arr["obs_ps"] == want.obs_ps.tolist()[:n_obs] and arr["obs_sv_label"] == want.obs_sv.tolist()[:n_obs]
    assert [h[0] for h in gene.haplotypes()] == [h["name"] for h in want.haps]
    assert [h[1] for h in gene.haplotypes()] == [h["core_allele"] for h in want.haps]
    db_vars = gene.variants()
    for vid, v in enumerate(want.var_list):
        k, label, s, e = gene.problem_variant(vid)
        if len(v) == 5:
            assert k == -1 and (label, s, e) == (v[4][3], v[4][1], v[4][2])
        else:
            m = db_vars[k]
            assert (m["position"], m["ref"], m["alt"]) == v[1:4]
            meta = want.var_meta[vid]
            assert (m["variant_id"], m["name"], m["is_core_variant"]) == (meta["variant_id"], meta["name"], meta["is_core_variant"])
    assert [gene.problem_sv_label(i) for i in range(len(want.sv_labels))] == want.sv_labels


@pytest.mark.parametrize("case", CASES, ids=lambda c: c[1])
def test_variant_gene_problem_equals_glue(D, oracle, case):
    db_name, vcf_key, with_ref, _dips, _inexact = case
    _name, want = vg.load_case(oracle, db_name, vcf_key, with_ref)
    db = D.Database(os.path.join(GOLDEN, "variant_dbs", db_name + ".json"))
    gene_name, chrom = db.gene_entries()[0]
    genome = json.load(open(os.path.join(GOLDEN, "test_reference.json"))) if with_ref else None
    gene = db.variant_gene(gene_name, genome[chrom] if genome else None)
    vcf = json.load(open(os.path.join(GOLDEN, "variant_vcfs.json")))[vcf_key]
    check_problem(D, gene, gene.problem(vcf_alleles(vcf)), want)


def test_variant_gene_homozygous_with_phase_set(D, pkg):
    """load_vcf_variants bails on "1|1" with a PS (src/diplotyper.rs:674-678; test_load_vcf_variants' bad_hom_ps.vcf.gz)"""
    db = D.Database(os.path.join(GOLDEN, "variant_dbs", "CACNA1S.json"))
    gene = db.variant_gene("CACNA1S")
    vcf = json.load(open(os.path.join(GOLDEN, "variant_vcfs.json")))["CACNA1S/bad_hom_ps.vcf.gz"]
    with pytest.raises(pkg.StarphaseError, match="Homozygous record detected with a phase set"):
        gene.problem(vcf_alleles(vcf))


@pytest.mark.parametrize("case", SV_CASES, ids=lambda c: c[0])
def test_variant_gene_structural_variants(D, oracle, pkg, case):
    sv_vcf, _dips, _inexact = case
    _name, want = vg.load_case(oracle, "DPYD-sv-test", "DPYD-sv-test/empty_small.vcf.gz", True, sv_vcf_key=sv_vcf)
    db = D.Database(os.path.join(GOLDEN, "variant_dbs", "DPYD-sv-test.json"))
    raw = json.load(open(os.path.join(GOLDEN, "variant_dbs", "DPYD-sv-test.json")))
    genome = json.load(open(os.path.join(GOLDEN, "test_reference.json")))
    gene = db.variant_gene("DPYD", genome[raw["gene_entries"]["DPYD"]["chromosome"]])
    vcfs = json.load(open(os.path.join(GOLDEN, "variant_vcfs.json")))
    got = gene.problem(vcf_alleles(vcfs["DPYD-sv-test/empty_small.vcf.gz"]), vcf_deletions(vcfs[sv_vcf]))
    check_problem(D, gene, got, want)
    # is_deletion through the library-built definitions == through the Python-flattened ones, over a sweep of intervals
    defs = pkg.ffi.SvDefinitions(raw["gene_collection"]["gene_dict"], raw["gene_entries"]["DPYD"]["structural_variants"])
    co = raw["gene_collection"]["gene_dict"]["DPYD"]["coordinates"]
    rng = np.random.default_rng(4)
    seen = set()
    spans = [(d[0], d[1]) for k in ("DPYD-sv-test/multi_del.vcf.gz", "DPYD-sv-test/hom_del.vcf.gz") for d in vcf_deletions(vcfs[k])]
    spans += [(co["start"] - 10, co["end"] + 10), (co["start"], co["end"])]
    for _ in range(400):
        a = int(rng.integers(co["start"] - 20000, co["end"] + 20000))
        spans.append((a, a + int(rng.integers(1, 400000))))
    for a, b in spans:
        label = defs.is_deletion(a, b)
        assert gene.is_deletion(a, b) == label
        seen.add(label)
    assert len(seen) >= 2, seen
    st = gene.stats
    svs = raw["gene_entries"]["DPYD"]["structural_variants"]
    assert (st.n_full_deletions, st.n_partial_deletions) == (len(svs.get("full_gene_deletions", {})), len(svs.get("partial_gene_deletions", {})))


def test_every_cpic_gene_of_the_bundled_database(D, oracle):
    path = os.path.join(GOLDEN, "gene_entries_v0.14.1.json.gz")
    raw = json.load(gzip.open(path))["gene_entries"]
    db = D.Database(path)
    assert len(raw) == 18
    for name in sorted(raw):
        vh, haps = vg.load_database_haplotypes(oracle, raw[name], None)
        gene = db.variant_gene(name)
        st = gene.stats
        assert (st.n_haplotypes, st.n_variants) == (len(haps), len(vh)), name
        assert st.n_skipped_haplotypes == len(raw[name]["defined_haplotypes"]) - len(haps)
        want = vg.Problem(vh, haps, {})
        check_problem(D, gene, gene.problem(), want)


# ------------------------------------------------------------------ the result file
def dip(a, b):
    return {"hap1": a, "hap2": b, "diplotype": f"{a}/{b}"}


def details(diplotypes, simple=None, inexact=None, variants=None, mappings=None, multi=None):
    return {"diplotypes": diplotypes, "simple_diplotypes": simple, "inexact_diplotypes": inexact, "variant_details": variants,
            "mapping_details": mappings, "multi_mapping_details": multi}


MD0 = {"pbstarphase_version": "", "cpic_version": "", "hla_version": "", "pharmvar_version": "", "build_time": "1970-01-01T00:00:00Z"}


def parsed(result):
    text = result.json()
    obj = json.loads(text)
    assert text == json.dumps(obj, indent=2, ensure_ascii=False)           # serde_json's pretty layout, byte for byte
    return obj


def test_starphase_json(D, pkg):
    """test_starphase_json + test_duplicate_diplotype (src/data_types/starphase_json.rs:333-358)"""
    r = D.Result(None, "1.2.3")
    d = D.GeneDetails().add_diplotype("B", "A")
    r.insert("CACNA1S", d, D.SUBALLELE_MATCH)
    obj = parsed(r)
    assert list(obj) == ["pbstarphase_version", "database_metadata", "gene_details"]
    assert obj["pbstarphase_version"] == "1.2.3" and obj["database_metadata"] == MD0 and list(obj["database_metadata"]) == list(MD0)
    assert obj["gene_details"] == {"CACNA1S": details([dip("B", "A")], variants=[])}
    assert list(obj["gene_details"]["CACNA1S"]) == ["diplotypes", "simple_diplotypes", "inexact_diplotypes", "variant_details", "mapping_details",
                                                    "multi_mapping_details"]
    with pytest.raises(pkg.StarphaseError, match="Entry for CACNA1S is already occupied."):
        r.insert("CACNA1S", d, D.SUBALLELE_MATCH)
    assert len(parsed(r)["gene_details"]) == 1


def test_new_from_mappings(D):
    """test_new_from_mappings (:360-375): variant_details is None"""
    r = D.Result()
    d = D.GeneDetails().add_diplotype("B", "A")
    d.add_mapping("read/1", "HLA:HLA00001", "A*01:01:01:01", cdna=(1098, 2, 0), dna=(3503, 7, 12, 3, 4), is_ignored=False)
    d.add_mapping("read/2", "HLA:HLA00005", "A*02:01:01:01", cdna=(1098, 0, 0), dna=None, is_ignored=True)
    r.insert("HLA-A", d, D.FROM_MAPPINGS)
    g = parsed(r)["gene_details"]["HLA-A"]
    assert g["variant_details"] is None and g["diplotypes"] == [dip("B", "A")] and g["simple_diplotypes"] is None
    assert g["mapping_details"] == [
        {"read_qname": "read/1", "best_hla_id": "HLA:HLA00001", "best_star_allele": "A*01:01:01:01", "best_mapping_stats": {
            "cdna_stats": {"seq_len": 1098, "nm": 2, "unmapped": 0, "clipped_start": None, "clipped_end": None},
            "dna_stats": {"seq_len": 3503, "nm": 7, "unmapped": 12, "clipped_start": 3, "clipped_end": 4}}, "is_ignored": False},
        {"read_qname": "read/2", "best_hla_id": "HLA:HLA00005", "best_star_allele": "A*02:01:01:01", "best_mapping_stats": {
            "cdna_stats": {"seq_len": 1098, "nm": 0, "unmapped": 0, "clipped_start": None, "clipped_end": None}, "dna_stats": None}, "is_ignored": True}]


REL = {"Match": 1, "Unexpected": 2, "Missing": 3}


def two_variants(d):
    d.add_variant(12345, "test_variant_1", "rs123456", "chr1", 1000, "A", "T", genotype=1)
    d.add_variant(67890, "test_variant_2", None, "chr1", 2000, "C", "G", genotype=4)
    return [{"variant_id": 12345, "variant_name": "test_variant_1", "dbsnp": "rs123456",
             "normalized_variant": {"chrom": "chr1", "position": 1000, "reference": "A", "alternate": "T", "sv_stats": None},
             "normalized_genotype": {"genotype": "0/1", "phase_set": None}, "is_core_variant": True},
            {"variant_id": 67890, "variant_name": "test_variant_2", "dbsnp": None,
             "normalized_variant": {"chrom": "chr1", "position": 2000, "reference": "C", "alternate": "G", "sv_stats": None},
             "normalized_genotype": {"genotype": "1/1", "phase_set": None}, "is_core_variant": True}]


def inexact_hap(base, rel, match_type):
    return {"base_haplotype": base, "match_type": match_type,
            "variant_relationships": [{"label": l, "is_vi": vi, "variant_state": st} for l, vi, st in sorted(rel, key=lambda r: (r[0], r[1], REL[r[2]]))]}


def test_new_inexact_diplotypes(D):
    """test_new_inexact_diplotypes (:377-435): one NO_MATCH diplotype, no simple diplotypes, the inexact ones and the variants kept"""
    d = D.GeneDetails()
    for a, b in (("1", "2"), ("3", "4")):
        d.add_inexact_diplotype((f"*{a}", [(f"test_variant_{a}", True, 1)]), (f"*{b}", [(f"test_variant_{b}", True, 1)]))
    d.add_diplotype("ignored", "ignored").add_simple_diplotype("x", "y")       # the constructor takes neither
    want_variants = two_variants(d)
    r = D.Result()
    r.insert("G", d, D.INEXACT_DIPLOTYPES)
    g = parsed(r)["gene_details"]["G"]
    assert g["diplotypes"] == [dip("NO_MATCH", "NO_MATCH")] and g["simple_diplotypes"] is None
    assert g["inexact_diplotypes"] == [
        {"basic_diplotype": dip(f"*{a}", f"*{b}"), "haplotype_1": inexact_hap(f"*{a}", [(f"test_variant_{a}", True, "Match")], "SubAlleleMatch"),
         "haplotype_2": inexact_hap(f"*{b}", [(f"test_variant_{b}", True, "Match")], "SubAlleleMatch")} for a, b in (("1", "2"), ("3", "4"))]
    assert g["variant_details"] == want_variants and g["mapping_details"] is None and g["multi_mapping_details"] is None


def test_new_core_match(D, pkg):
    """test_new_core_match (:437-489) + the two length checks of the constructor (:101-111)"""
    d = D.GeneDetails().add_diplotype("*1", "*2").add_simple_diplotype("*1", "*2")
    d.add_inexact_diplotype(("*1", [("test_variant_1", True, 1), ("test_variant_2", True, 2)]), ("*2", [("test_variant_2", True, 1)]))
    want_variants = two_variants(d)
    r = D.Result()
    r.insert("G", d, D.CORE_MATCH)
    g = parsed(r)["gene_details"]["G"]
    assert g["diplotypes"] == [dip("*1", "*2")] and g["simple_diplotypes"] == [dip("*1", "*2")]
    assert g["inexact_diplotypes"] == [{
        "basic_diplotype": dip("(*1 +test_variant_2)", "*2"),
        "haplotype_1": inexact_hap("*1", [("test_variant_1", True, "Match"), ("test_variant_2", True, "Unexpected")], "NoMatch"),
        "haplotype_2": inexact_hap("*2", [("test_variant_2", True, "Match")], "SubAlleleMatch")}]
    assert g["variant_details"] == want_variants and g["mapping_details"] is None and g["multi_mapping_details"] is None
    d.add_diplotype("*3", "*4")
    with pytest.raises(pkg.StarphaseError, match="diplotypes and simple_diplotypes must be the same length"):
        r.insert("H", d, D.CORE_MATCH)
    d.add_simple_diplotype("*3", "*4")
    with pytest.raises(pkg.StarphaseError, match="diplotypes and inexact_diplotypes must be the same length"):
        r.insert("H", d, D.CORE_MATCH)
    r.insert("H", d, D.SUBALLELE_MATCH)                                        # no inexact list in that constructor
    assert parsed(r)["gene_details"]["H"]["inexact_diplotypes"] is None


def test_multi_mappings_no_match_and_structural_variant(D, tmp_path):
    db = D.Database(os.path.join(GOLDEN, "cyp2d6_db_v0.14.1.json.gz"))
    r = D.Result(db, "0.14.1-test")
    d = D.GeneDetails().add_diplotype("*4.001 + *68", "*1.001").add_simple_diplotype("*4+*68", "*1")
    d.add_diplotype_only("*4.001 + *68", "(*1.001 +rs1)")
    d.add_multi_mapping("m84/1/ccs", 100, 4500, 0, "*4.001").add_multi_mapping("m84/2/ccs", 0, 3000, 2, "*68")
    r.insert("CYP2D6", d, D.FROM_MULTI_MAPPINGS)
    r.insert("ZZZ", None, D.NO_MATCH)
    sv = D.GeneDetails().add_diplotype("generic exon del", "Reference")
    sv.add_variant(0, "structural_variant", None, "chr1", 97515786, "", "", genotype=2, phase_set=77, sv=(97515786, 97771853, "generic exon del"))
    r.insert("DPYD", sv, D.SUBALLELE_MATCH)
    obj = parsed(r)
    assert list(obj["gene_details"]) == ["CYP2D6", "DPYD", "ZZZ"]             # BTreeMap order
    assert obj["database_metadata"] == db.metadata
    g = obj["gene_details"]["CYP2D6"]
    assert g["inexact_diplotypes"] == [{"basic_diplotype": dip("*4.001 + *68", "(*1.001 +rs1)"), "haplotype_1": None, "haplotype_2": None}]
    assert g["multi_mapping_details"] == [
        {"read_qname": "m84/1/ccs", "read_position": {"start": 100, "end": 4500}, "consensus_id": 0, "consensus_star_allele": "*4.001"},
        {"read_qname": "m84/2/ccs", "read_position": {"start": 0, "end": 3000}, "consensus_id": 2, "consensus_star_allele": "*68"}]
    assert obj["gene_details"]["ZZZ"] == details([dip("NO_MATCH", "NO_MATCH")])
    v = obj["gene_details"]["DPYD"]["variant_details"][0]
    assert v["normalized_variant"]["sv_stats"] == {"sv_type": "Deletion", "start": 97515786, "end": 97771853, "haplotype_label": "generic exon del"}
    assert v["normalized_genotype"] == {"genotype": "0|1", "phase_set": 77}
    for name in ("out.json", "out.json.gz"):
        p = tmp_path / name
        r.save(str(p))
        data = gzip.open(p).read() if name.endswith(".gz") else p.read_bytes()
        assert data.decode() == r.json()


def test_pharmcat_tsv(D, pkg, tmp_path):
    """save_pharmcat_tsv (src/main.rs:190-241): key order, de-duplicated simple diplotypes, "Multiple", brackets around '+' haplotypes,
    MT-RNR1 as one haplotype"""
    r = D.Result(None, "x")
    r.insert("CYP2D6", D.GeneDetails().add_diplotype("*4.001 + *68", "*1.001").add_simple_diplotype("*4 + *68", "*1"), D.FROM_MULTI_MAPPINGS)
    r.insert("CYP2C19", D.GeneDetails().add_diplotype("*1", "*2").add_diplotype("*2", "*1").set_simple_diplotypes(False), D.SUBALLELE_MATCH)   # the same pair twice
    r.insert("UGT1A1", D.GeneDetails().add_diplotype("*1", "*80+*28").add_diplotype("*28", "*80").add_simple_diplotype("*1", "*80+*28").add_simple_diplotype("*28", "*80"),
             D.SUBALLELE_MATCH)
    r.insert("MT-RNR1", D.GeneDetails().add_diplotype("961T>del", "961T>del"), D.SUBALLELE_MATCH)
    r.insert("ABCG2", None, D.NO_MATCH)
    want = ("#gene\tdiplotype\n" "ABCG2\tNO_MATCH/NO_MATCH\n" "CYP2C19\t*1/*2\n" "CYP2D6\t[*4 + *68]/*1\n" "MT-RNR1\t961T>del\n" "UGT1A1\tMultiple/Multiple\n")
    assert r.pharmcat_tsv() == want
    r.save_pharmcat_tsv(str(tmp_path / "p.tsv"))
    assert (tmp_path / "p.tsv").read_text() == want
    r2 = D.Result()
    r2.insert("MT-RNR1", D.GeneDetails().add_diplotype("961T>del", "961T>del+Cn"), D.SUBALLELE_MATCH)
    r2.insert("G\t1", D.GeneDetails().add_diplotype('a"b', "c"), D.SUBALLELE_MATCH)
    assert r2.pharmcat_tsv() == '#gene\tdiplotype\n"G\t1"\t"a""b/c"\nMT-RNR1\tUnknown\n'


def test_duplicate_json_keys_resolve_to_the_last_one(D, tmp_path):
    """serde fills its maps by insertion: of two members with one name the later one stays.  hla_sequences of a real database has ~40,000 members:
    they are looked up through a sorted index built when the object is read (sp_json.h), the few members of a small object by a scan -- both
    resolve a repeated key the same way."""
    raw = json.load(open(os.path.join(GOLDEN, "variant_dbs", "CACNA1S.json")))
    good = raw["gene_entries"]["CACNA1S"]
    n_good = len(good["defined_haplotypes"])
    damaged = dict(good, defined_haplotypes={})                               # a first copy of the entry without haplotypes: must be replaced by the second
    text = json.dumps(raw)
    dup = text.replace('"gene_entries": {', '"gene_entries": {"CACNA1S": ' + json.dumps(damaged) + ', ', 1)
    assert dup != text
    # many members in front so that the object is one of the indexed ones (>= 32 members)
    filler = ", ".join(f'"ZZ{k}": ' + json.dumps(damaged) for k in range(40))
    big = dup.replace('"gene_entries": {', '"gene_entries": {' + filler + ", ", 1)
    for name, body in (("small", dup), ("big", big)):
        path = tmp_path / f"{name}.json"
        path.write_text(body)
        gene = D.Database(str(path)).variant_gene("CACNA1S")
        assert gene.stats.n_haplotypes == n_good > 0, name
