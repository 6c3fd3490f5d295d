"""cyp_db.py -- CPU ORACLE (test infrastructure only): the CYP2D6 search templates and typing tables.

Pure-Python restatement (small, runs once per database) of
    generate_cyp_hybrids                    src/cyp2d6/definitions.rs:346-464
    LoadedVariants::load_variant_database   src/cyp2d6/haplotyper.rs:650-773
    Cyp2d6Extractor::new (tables only)      src/cyp2d6/haplotyper.rs:45-132
    Cyp2d6RegionLabel::full_allele          src/cyp2d6/region_label.rs:131-170
Pinned by the reference's own test_load_variant_database (src/cyp2d6/haplotyper.rs:918-933: 387 variants, 144 VI, first / last
position, label indices) on the v0.9.0 gene definitions (tests/golden/cyp2d6_gene_def_v0.9.0.json.gz); generate_cyp_hybrids has no
test in the reference (haplotyper.rs:908-915 says so), the template count / lengths of SURVEY.md 8(a) row a14 are checked instead.
Only tests/ may import this module.
"""

STAR5_PRE_BUFFER = 500      # definitions.rs:13
STAR5_POST_BUFFER = 3000    # definitions.rs:14

# Cyp2d6RegionType in declaration order (region_label.rs:5-34): the derived Ord of the enum
TYPES = ["UNKNOWN", "REP6", "CYP2D6", "link_region", "REP7", "spacer", "CYP2D7", "CYP2D6*5", "Hybrid", "FalseAllele"]


def full_allele(type_name, subtype):
    """Cyp2d6RegionLabel::full_allele"""
    if type_name == "CYP2D6":
        return "CYP2D6*" + subtype if subtype is not None else "CYP2D6"
    if type_name == "Hybrid":
        return subtype if subtype is not None else "Hybrid"
    if type_name == "FalseAllele":
        return "FalseAllele_" + subtype if subtype is not None else "FalseAllele"
    return type_name


def generate_cyp_hybrids(get_slice, config):
    """-> {(type_name, subtype|None): sequence}.  get_slice(start, end) reads the chromosome (0-based, half open)."""
    cc, regions = config["cyp_coordinates"], config["cyp_regions"]
    g1s, g1e = cc["CYP2D6"]["start"], cc["CYP2D6"]["end"]
    g2s, g2e = cc["CYP2D7"]["start"], cc["CYP2D7"]["end"]
    ret = {("CYP2D6", None): get_slice(g1s, g1e), ("CYP2D7", None): get_slice(g2s, g2e)}
    s5 = config["cyp2d6_star5_del"]
    ret[("CYP2D6*5", None)] = get_slice(s5["start"] - STAR5_PRE_BUFFER, s5["start"]) + get_slice(s5["end"], s5["end"] + STAR5_POST_BUFFER)
    for exon_index in range(1, 10):
        e1, e2 = regions["CYP2D6"][f"exon{exon_index}"], regions["CYP2D7"][f"exon{exon_index}"]
        cuts = []
        if exon_index != 1:
            cuts.append((f"exon{exon_index}", e1["end"], e2["end"]))          # start of the exon on the coding strand
        if exon_index != 9:
            cuts.append((f"intron{exon_index}", e1["start"], e2["start"]))    # end of the exon = start of the intron
        for name, bp1, bp2 in cuts:
            ret[("Hybrid", f"CYP2D6::CYP2D7::{name}")] = get_slice(g2s, bp2) + get_slice(bp1, g1e)
            ret[("Hybrid", f"CYP2D7::CYP2D6::{name}")] = get_slice(g1s, bp1) + get_slice(bp2, g2e)
    for key, type_name in (("REP6", "REP6"), ("REP7", "REP7"), ("spacer", "spacer"), ("link_region", "link_region")):
        ret[(type_name, None)] = get_slice(cc[key]["start"], cc[key]["end"])
    return ret


def template_order(hybrids):
    """the visiting order of find_base_type_in_sequence: keys sorted by full_allele() (haplotyper.rs:175-183)"""
    return sorted(hybrids, key=lambda k: full_allele(*k).encode())


MAPPED_HYBRIDS = {("CYP2D6", None), ("Hybrid", "CYP2D6::CYP2D7::exon9")}        # haplotyper.rs:117-123


def load_variant_database(gene_def):
    """-> dict(variants=[(pos, ref, alt)], labels=[..], vi=[bool], lookup={(pos, ref, alt): i}, label_lookup={label: i})"""
    inserted, unsorted, vi_set = set(), [], {}
    for allele_id in sorted(gene_def, key=lambda k: k.encode()):                # BTreeMap<String, AlleleDefinition>
        for v in gene_def[allele_id]["variants"]:
            key = (v["position"], v["reference"], v["alternate"])
            if "VI" in v.get("extras", {}):
                vi_set[key] = v["extras"]["VI"]
            if key not in inserted:
                label = v["id"] if v.get("id") is not None else f'{v["chrom"]}:{v["position"] + 1}{v["reference"]}>{v["alternate"]}'
                unsorted.append((key, label))
                inserted.add(key)
    unsorted.sort(key=lambda x: x[0][0])                                        # stable, by position only
    variants = [k for k, _ in unsorted]
    labels = [l for _, l in unsorted]
    return dict(variants=variants, labels=labels, vi=[k in vi_set for k in variants], lookup={k: i for i, k in enumerate(variants)},
                label_lookup={l: i for i, l in enumerate(labels)})


def haplotype_lookup(gene_def, loaded):
    """-> ([star_allele in BTreeMap<Cyp2d6RegionLabel, _> order], rows of 0/1)"""
    rows = {}
    for allele_id in sorted(gene_def, key=lambda k: k.encode()):
        d = gene_def[allele_id]
        assert d["gene_name"] == "CYP2D6"
        row = [0] * len(loaded["variants"])
        for v in d["variants"]:
            row[loaded["lookup"][(v["position"], v["reference"], v["alternate"])]] = 1
        rows[d["star_allele"]] = row
    names = sorted(rows, key=lambda s: s.encode())
    return names, [rows[n] for n in names]
