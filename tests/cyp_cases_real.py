"""BASELINE configs[2] scenarios on the synthetic chr22 locus: truth haplotypes and the strings the caller
should report for them (convert_chain_to_hap reverses the chain: the gene copy furthest from REP6 comes first)."""
import gzip
import json
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_db():
    d = json.load(gzip.open(os.path.join(GOLDEN, "cyp2d6_db_v0.14.1.json.gz")))
    return d["cyp2d6_config"], d["cyp2d6_gene_def"]


def scenarios(locus):
    a = locus.star_allele
    h = locus.haplotype
    hyb68 = locus.hybrid("CYP2D6::CYP2D7::exon2")
    return [
        ("*1/*2", [h([a("1.001")]), h([a("2.001")])], ["*1.001", "*2.001"]),
        ("*4/*4", [h([a("4.001")]), h([a("4.001")])], ["*4.001", "*4.001"]),
        ("*5/*1", [h(None), h([a("1.001")])], ["*5", "*1.001"]),
        ("*4+*68/*1", [h([a("4.001"), hyb68]), h([a("1.001")])], ["*68 + *4.001", "*1.001"]),
        ("*10+*36/*10", [h([a("10.001"), a("36.001")]), h([a("10.001")])], ["*36.001 + *10.001", "*10.001"]),
        ("*2x2/*1", [h([a("2.001"), a("2.001")]), h([a("1.001")])], ["*2.001x2", "*1.001"]),
    ]
