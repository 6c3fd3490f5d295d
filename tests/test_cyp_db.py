"""SURVEY.md 8(a) row a14 through the C ABI (host tables only, no GPU): generate_cyp_hybrids, load_variant_database,
haplotype_lookup.  Checks: the reference's own test_load_variant_database (src/cyp2d6/haplotyper.rs:918-933), the library against
the independent Python restatement oracle/cyp_db.py on two databases, and the template inventory of SURVEY.md 8(a)."""
import gzip
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cyp_db as oracle_db  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def dbs():
    new = json.load(gzip.open(os.path.join(GOLDEN, "cyp2d6_db_v0.14.1.json.gz")))
    old = json.load(gzip.open(os.path.join(GOLDEN, "cyp2d6_gene_def_v0.9.0.json.gz")))
    return new["cyp2d6_config"], new["cyp2d6_gene_def"], old["cyp2d6_gene_def"]


def test_oracle_load_variant_database_reference_vector(dbs):
    """test_load_variant_database (haplotyper.rs:918-933) against the oracle"""
    _cfg, _new, old = dbs
    lv = oracle_db.load_variant_database(old)
    assert lv["variants"][0][0] == 42126309 and lv["variants"][-1][0] == 42132374
    assert len(lv["variants"]) == 387 and sum(lv["vi"]) == 144
    assert lv["label_lookup"]["rs12169962"] == 0 and lv["label_lookup"]["rs1080985"] == 386


def test_library_load_variant_database_reference_vector(pkg, dbs):
    """the same vector through sp_cyp_db_create (the v0.9.0 database has no cyp2d6_config: the default coordinates apply)"""
    from pb_starphase_amd import synth
    cfg, _new, old = dbs
    locus = synth.Chr22Locus(cfg, old, seed=5)
    db = pkg.ffi.CypDb(None, cfg, old, locus.sequence, locus.start)
    st = db.stats
    assert (st.first_variant_pos, st.last_variant_pos, st.n_variants, st.n_vi) == (42126309, 42132374, 387, 144)
    assert db.index_label("rs12169962") == 0 and db.index_label("rs1080985") == 386
    with pytest.raises(KeyError):
        db.index_label("no such label")
    with pytest.raises(KeyError):
        db.index_variant(1, "A", "C")


@pytest.mark.parametrize("which", ["v0.14.1", "v0.9.0"])
def test_library_equals_oracle(pkg, dbs, which):
    from pb_starphase_amd import synth
    cfg, new, old = dbs
    gene_def = new if which == "v0.14.1" else old
    locus = synth.Chr22Locus(cfg, gene_def, seed=11)
    db = pkg.ffi.CypDb(None, cfg, gene_def, locus.sequence, locus.start)
    hyb = oracle_db.generate_cyp_hybrids(locus.slice, cfg)
    order = oracle_db.template_order(hyb)
    got = db.templates()
    assert len(got) == len(order) == 39
    for (t, sub, full, seq, deep), key in zip(got, order):
        assert oracle_db.TYPES[t] == key[0] and sub == key[1]
        assert full == oracle_db.full_allele(*key)
        assert seq == hyb[key]
        assert deep == (key in oracle_db.MAPPED_HYBRIDS)
    lens = {full: len(seq) for (_t, _s, full, seq, _d) in got}
    assert lens["CYP2D6"] == 6165 and lens["CYP2D7"] == 5938 and lens["CYP2D6*5"] == 3500
    assert lens["REP6"] == lens["REP7"] == 2772 and lens["spacer"] == 1564 and lens["link_region"] == 2919
    assert sum(1 for (t, *_r) in got if oracle_db.TYPES[t] == "Hybrid") == 32
    lv = oracle_db.load_variant_database(gene_def)
    mine = db.variants()
    assert [(p, r, a) for (p, r, a, _l, _v) in mine] == lv["variants"]
    assert [l for (_p, _r, _a, l, _v) in mine] == lv["labels"]
    assert [v for (*_x, v) in mine] == lv["vi"]
    for k, i in list(lv["lookup"].items())[::7]:
        assert db.index_variant(*k) == i
    for l, i in lv["label_lookup"].items():
        assert db.index_label(l) == i
    names, rows = oracle_db.haplotype_lookup(gene_def, lv)
    got_names, got_rows = db.alleles()
    assert got_names == names
    assert np.array_equal(got_rows, np.array(rows, np.uint8))
    assert db.stats.backbone_len == 6165


def test_bad_inputs(pkg, dbs):
    from pb_starphase_amd import synth
    cfg, new, _old = dbs
    locus = synth.Chr22Locus(cfg, new, seed=11)
    with pytest.raises(pkg.StarphaseError):                                    # a coordinate outside the window
        pkg.ffi.CypDb(None, cfg, new, locus.sequence[:20000], locus.start)
    moved = json.loads(json.dumps(cfg))
    moved["cyp_coordinates"]["CYP2D6"]["start"] += 200                         # the region no longer holds the first variant
    with pytest.raises(pkg.StarphaseError):
        pkg.ffi.CypDb(None, moved, new, locus.sequence, locus.start)


def test_star_alleles_apply_cleanly(dbs):
    """every star allele of the database can be realised on the synthetic locus (reference alleles agree with the window)"""
    from pb_starphase_amd import synth
    cfg, new, _old = dbs
    locus = synth.Chr22Locus(cfg, new, seed=3)
    for d in list(new.values())[::9]:
        s = locus.star_allele(d["star_allele"])
        assert abs(len(s) - 6165) < 200
