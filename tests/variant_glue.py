"""Host prep of the variant-gene path restated for the tests (Python, small inputs only): load_database_haplotypes
(src/diplotyper.rs:437-538), load_vcf_variants on decoded VCF rows (:551-737), the call_diplotypes packaging (:130-204),
and the flattening into the integer problem that oracle/variant.c and sp_variant_solve consume."""
import ctypes as C
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GT = {"0/1": 1, "0|1": 2, "1|0": 3, "1/1": 4}
MAXLEN, MAXTIES, MAXDIP = 4096, 64, 4096


class NormVariant(C.Structure):
    _fields_ = [("chrom", C.c_char * 64), ("position", C.c_int64), ("ref", C.c_char * MAXLEN), ("alt", C.c_char * MAXLEN)]


class VariantProblem(C.Structure):
    _fields_ = [("n_haps", C.c_int32), ("hap_is_sv", C.c_void_p), ("hap_is_core", C.c_void_p), ("slot_off", C.c_void_p),
                ("alt_off", C.c_void_p), ("alt_var", C.c_void_p), ("n_vars", C.c_int32), ("var_is_core", C.c_void_p),
                ("n_obs", C.c_int32), ("obs_var", C.c_void_p), ("obs_gt", C.c_void_p), ("obs_ps", C.c_void_p), ("obs_sv_label", C.c_void_p)]


class VariantResult(C.Structure):
    _fields_ = [("score", C.c_int64 * 4), ("n_dip", C.c_int32), ("overflow", C.c_int32), ("dip", (C.c_int32 * 2) * MAXDIP),
                ("dip_comb", C.c_int32 * MAXDIP)]


def normalize(oracle, chrom, pos, ref, alt, genome):
    """NormalizedVariant::new -> (chrom, position, ref, alt) tuple (its derived Ord) or raises ValueError"""
    out = NormVariant()
    err = C.create_string_buffer(256)
    seq = genome.get(chrom) if genome else None
    if genome is not None and seq is None:
        raise ValueError(f"Reference genome does not contain contig {chrom!r}")
    if hasattr(oracle, "normalize_variant"):          # the product's host routine (ProductNormalizer below)
        return oracle.normalize_variant(chrom, pos, ref, alt, seq)
    oracle.L.osp_normalize_variant.argtypes = [C.c_char_p, C.c_int64, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int64, C.POINTER(NormVariant),
                                                C.c_char_p, C.c_size_t]
    rc = oracle.L.osp_normalize_variant(chrom.encode(), int(pos), ref.encode(), alt.encode(), seq.encode() if seq else None,
                                        len(seq) if seq else 0, C.byref(out), err, 256)
    if rc != 0:
        raise ValueError(err.value.decode())
    return (out.chrom.decode(), int(out.position), out.ref.decode(), out.alt.decode())


IUPAC = {"K": ["G", "T"], "M": ["A", "C"], "R": ["A", "G"], "S": ["C", "G"], "W": ["A", "T"], "Y": ["C", "T"],
         "B": ["C", "G", "T"], "D": ["A", "G", "T"], "H": ["A", "C", "T"], "V": ["A", "C", "G"]}


class ProductNormalizer:
    """Routes the string preparation through the library's sp_variant_normalize / sp_variant_multi_normalize."""
    def __init__(self, pkg):
        self.pkg = pkg

    def normalize_variant(self, chrom, pos, ref, alt, seq):
        try:
            p, r, a = self.pkg.ffi.normalize_variant(seq, pos, ref, alt)
        except self.pkg.StarphaseError as e:
            raise ValueError(str(e))
        return (chrom, p, r, a)

    def multi_new(self, chrom, pos, ref, alt, seq):
        try:
            got = self.pkg.ffi.multi_normalize_variant(seq, pos, ref, alt)
        except self.pkg.StarphaseError as e:
            raise ValueError(str(e))
        return [None if g is None else (chrom,) + g for g in got]


def multi_new(oracle, chrom, pos, ref, alt, genome):
    """NormalizedVariant::multi_new (src/data_types/normalized_variant.rs:174-214)"""
    if hasattr(oracle, "multi_new"):
        seq = genome.get(chrom) if genome else None
        if genome is not None and seq is None:
            raise ValueError(f"Reference genome does not contain contig {chrom!r}")
        return oracle.multi_new(chrom, pos, ref, alt, seq)
    alts = IUPAC.get(alt) or alt.split("; ")
    return [None if a == ref else normalize(oracle, chrom, pos, ref, a, genome) for a in alts]


def load_database_haplotypes(oracle, gene, genome):
    variant_hash, haps = {}, []
    for name in sorted(gene["defined_haplotypes"]):
        h = gene["defined_haplotypes"][name]
        slots, metas, ok = [], [], True
        for vid in sorted(h["haplotype"], key=int):
            var = gene["variants"][vid]
            ref, alt = var["alleles"][0], h["haplotype"][vid]
            if ref == alt:
                continue
            try:
                slots.append(multi_new(oracle, gene["chromosome"], var["position"] - 1, ref, alt, genome))
                metas.append({"variant_id": int(vid), "name": var["name"], "is_core_variant": var.get("is_core_variant", True)})
            except ValueError:
                ok = False
                break
        if ok:
            for slot, meta in zip(slots, metas):
                for nv in slot:
                    if nv is not None:
                        variant_hash.setdefault(nv, meta)
            haps.append({"name": name, "core_allele": h.get("core_allele"), "slots": slots})
    return variant_hash, haps


def load_vcf_variants(oracle, vcf, variant_hash, genome):
    """rows of one decoded test VCF; the sample is the first (only) sample column"""
    sample = [c for c in vcf["columns"] if c not in ("CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO", "FORMAT")][0]
    found = {}
    for variant in sorted(variant_hash):
        chrom, position = variant[0], variant[1]
        lo, hi = max(0, position - 50), position + 50
        genotype = None
        for row in vcf["rows"]:
            pos0 = int(row["POS"]) - 1
            if row["CHROM"] != chrom or not (pos0 < hi and pos0 + len(row["REF"]) > lo):
                continue
            fmt = dict(zip(row["FORMAT"].split(":"), row[sample].split(":")))
            gt = fmt["GT"]
            phased = "|" in gt
            a = gt.replace("|", "/").split("/")
            if len(a) != 2 or "." in a:
                continue
            gt1, gt2 = int(a[0]), int(a[1])
            ps = None
            if phased:
                ps = int(fmt["PS"]) if fmt.get("PS", ".") != "." else None
                if ps is None:
                    phased = False
            elif fmt.get("PS", ".") != ".":
                ps = None            # PS on an unphased record is only read when phased
            for ai, alt in enumerate(row["ALT"].split(","), start=1):
                try:
                    nv = normalize(oracle, chrom, pos0, row["REF"], alt, genome)
                except ValueError:
                    continue
                if nv != variant:
                    continue
                has_ps = fmt.get("PS", ".") != "." if phased or (ai == gt1 and ai == gt2) else False
                if ai == gt1 and ai == gt2:
                    if fmt.get("PS", ".") != "." and "|" in gt:
                        raise ValueError("Homozygous record detected with a phase set ID (PS)")
                    genotype = (4, None)
                elif ai == gt1 and phased:
                    genotype = (3, ps)
                elif ai == gt2 and phased:
                    genotype = (2, ps)
                elif (ai == gt1 or ai == gt2) and not phased:
                    genotype = (1, None)
        if genotype is not None:
            found[variant] = genotype
    return found


class SvDefsStruct(C.Structure):
    """osp_sv_definitions (same layout as the library's sp_sv_definitions)"""
    _fields_ = [("n_genes", C.c_int32), ("gene_start", C.c_void_p), ("gene_end", C.c_void_p), ("gene_forward", C.c_void_p),
                ("exon_off", C.c_void_p), ("exon_start", C.c_void_p), ("exon_end", C.c_void_p),
                ("n_full", C.c_int32), ("full_generic", C.c_void_p), ("full_off", C.c_void_p), ("full_gene", C.c_void_p),
                ("n_partial", C.c_int32), ("partial_generic", C.c_void_p), ("partial_off", C.c_void_p), ("partial_gene", C.c_void_p),
                ("partial_first", C.c_void_p), ("partial_end", C.c_void_p)]


def oracle_is_deletion(oracle, defs, start, end):
    """defs: the package's SvDefinitions (a pure-Python flattening); the decision is oracle/variant.c's osp_is_deletion"""
    kind, index = C.c_int32(0), C.c_int32(-1)
    d = defs.struct(SvDefsStruct)
    oracle.L.osp_is_deletion.restype = C.c_int
    oracle.L.osp_is_deletion.argtypes = [C.POINTER(SvDefsStruct), C.c_uint64, C.c_uint64, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    if oracle.L.osp_is_deletion(C.byref(d), int(start), int(end), C.byref(kind), C.byref(index)) != 0:
        raise ValueError("Gene collection does not contain a definition for a deletable gene")
    return defs.label(kind.value, index.value)


def load_sv_vcf_variants(is_deletion, vcf, structural_variants, gene_dict, max_sv_length=1000000):
    """load_sv_vcf_variants (src/diplotyper.rs:739-857) on the rows of one decoded test VCF.  Returns {sv variant: (genotype, ps)};
    an SV variant is the 5-tuple (chrom, start, "", "", ("Deletion", start, end, label)) -- NormalizedVariant::new_sv's fields, so
    tuple order = the derived Ord (sv_stats None sorts before Some: shorter tuples first)."""
    if not structural_variants:
        return {}
    genes = set()
    for fd in structural_variants.get("full_gene_deletions", {}).values():
        genes |= set(fd["full_genes_deleted"])
    for pd in structural_variants.get("partial_gene_deletions", {}).values():
        genes |= set(pd["exons_deleted"])
    chrom, lo, hi = None, None, 0
    for g in sorted(genes):
        if g not in gene_dict:
            raise ValueError(f"Missing gene definition ({g}) for structural variant")
        c = gene_dict[g]["coordinates"]
        if chrom is not None and chrom != c["chrom"]:
            raise ValueError("Structural variant gene set is not all on the same chromosome")
        chrom = c["chrom"]
        lo, hi = c["start"] if lo is None else min(lo, c["start"]), max(hi, c["end"])
    if chrom is None:
        return {}
    sample = [c for c in vcf["columns"] if c not in ("CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO", "FORMAT")][0]
    found = {}
    for row in vcf["rows"]:
        info = dict(kv.split("=", 1) for kv in row["INFO"].split(";") if "=" in kv)
        start = int(row["POS"]) - 1
        if row["CHROM"] != chrom or len(row["ALT"].split(",")) != 1:
            continue
        if "SVTYPE" not in info:
            raise ValueError("No INFO:SVTYPE in record")
        if info["SVTYPE"] != "DEL":
            continue
        if "END" not in info:
            raise ValueError("No INFO:END in record")
        end = int(info["END"])
        if not (start < hi and end > lo):            # the indexed fetch of the gene span
            continue
        if end - start > max_sv_length:
            continue
        label = is_deletion(start, end)
        if label is None:
            continue
        fmt = dict(zip(row["FORMAT"].split(":"), row[sample].split(":")))
        gt = fmt["GT"]
        a = gt.replace("|", "/").split("/")
        if len(a) != 2 or "." in a:
            continue
        phased = "|" in gt
        ps = None
        if phased:
            if fmt.get("PS", ".") != ".":
                ps = int(fmt["PS"])
            else:
                phased = False
        g1, g2 = int(a[0]), int(a[1])
        assert g1 < 2 and g2 < 2
        if g1 == g2:
            if g1 == 0:
                continue
            code = 4
        elif phased:
            code = 2 if g1 == 0 else 3
        else:
            code = 1
        key = (chrom, start, "", "", ("Deletion", start, end, label))
        if key in found:
            raise ValueError("Detected duplicate entry for normalized SV")
        found[key] = (code, ps)
    return found


def build_core_allele_lookup(haps, structural_variants):
    """src/diplotyper.rs:378-399"""
    lookup = {h["name"]: (h["core_allele"] or h["name"]) for h in haps}
    for cls in ("full_gene_deletions", "partial_gene_deletions"):
        for key in (structural_variants or {}).get(cls, {}):
            lookup[key] = key.split(".")[0]
    return lookup


def simplify_diplotypes(diplotypes, lookup):
    """src/diplotyper.rs:408-421 (KeyError where the reference reports a missing core allele)"""
    return [(lookup[a], lookup[b]) for a, b in diplotypes]


class Problem:
    def __init__(self, variant_hash, haps, observed, structural_variants=None):
        # observed SVs are variants too (they reach quant_match as "extra" ones); all SVs are core variants (:156-164)
        sv_obs = [v for v in observed if len(v) == 5]
        self.var_list = sorted(set(variant_hash) | set(sv_obs))
        self.var_id = {v: i for i, v in enumerate(self.var_list)}
        self.var_meta = [variant_hash.get(v) or {"variant_id": None, "name": "structural_variant", "is_core_variant": True} for v in self.var_list]
        self.sv_labels = sorted({v[4][3] for v in sv_obs})
        self.core_lookup = build_core_allele_lookup(haps, structural_variants)
        self.haps = haps
        slot_off, alt_off, alt_var = [0], [0], []
        for h in haps:
            for slot in h["slots"]:
                alt_var += [(-1 if nv is None else self.var_id[nv]) for nv in slot]
                alt_off.append(len(alt_var))
            slot_off.append(len(alt_off) - 1)
        self.hap_is_sv = np.zeros(len(haps), np.uint8)
        self.hap_is_core = np.array([1 if h["core_allele"] is None else 0 for h in haps], np.uint8)
        self.slot_off = np.array(slot_off, np.int32)
        self.alt_off = np.array(alt_off, np.int32)
        self.alt_var = np.array(alt_var if alt_var else [0], np.int32)
        self.var_is_core = np.array([1 if m["is_core_variant"] else 0 for m in self.var_meta] or [0], np.uint8)
        self.obs = sorted(observed)
        self.obs_var = np.array([self.var_id[v] for v in self.obs] or [0], np.int32)
        self.obs_gt = np.array([observed[v][0] for v in self.obs] or [0], np.int32)
        self.obs_ps = np.array([(-1 if observed[v][1] is None else observed[v][1]) for v in self.obs] or [0], np.int64)
        self.obs_sv = np.array([(self.sv_labels.index(v[4][3]) if len(v) == 5 else -1) for v in self.obs] or [-1], np.int32)

    def struct(self):
        p = VariantProblem()
        p.n_haps = len(self.haps)
        p.hap_is_sv, p.hap_is_core = self.hap_is_sv.ctypes.data, self.hap_is_core.ctypes.data
        p.slot_off, p.alt_off, p.alt_var = self.slot_off.ctypes.data, self.alt_off.ctypes.data, self.alt_var.ctypes.data
        p.n_vars, p.var_is_core = len(self.var_list), self.var_is_core.ctypes.data
        p.n_obs = len(self.obs)
        p.obs_var, p.obs_gt, p.obs_ps, p.obs_sv_label = self.obs_var.ctypes.data, self.obs_gt.ctypes.data, self.obs_ps.ctypes.data, self.obs_sv.ctypes.data
        return p


def oracle_solve(oracle, prob):
    res = VariantResult()
    p = prob.struct()
    oracle.L.osp_solve_diplotype.restype = C.c_int32
    rc = oracle.L.osp_solve_diplotype(C.byref(p), C.byref(res))
    assert rc == 0 and not res.overflow
    return tuple(res.score), [(res.dip[i][0], res.dip[i][1], res.dip_comb[i]) for i in range(res.n_dip)]


def het_split(prob, combination):
    """the two observed haplotypes of a het assignment (solve_diplotype, src/diplotyper.rs:1263-1317), as variant id lists"""
    h1 = [int(prob.obs_var[o]) for o in range(len(prob.obs)) if prob.obs_gt[o] == 4]
    h2 = list(h1)
    combo_index, lookup = 0, {}
    for o in range(len(prob.obs)):
        if prob.obs_gt[o] == 4:
            continue
        ps = int(prob.obs_ps[o])
        if ps >= 0:
            if ps not in lookup:
                lookup[ps] = (combination >> combo_index) & 1
                combo_index += 1
            is_h1 = lookup[ps]
        else:
            is_h1 = (combination >> combo_index) & 1
            combo_index += 1
        (h1 if bool(is_h1) == (prob.obs_gt[o] != 3) else h2).append(int(prob.obs_var[o]))
    return h1, h2


def quant_match(oracle, prob, h, obs_ids):
    n = len(obs_ids) + 1
    ns = int(prob.slot_off[h + 1] - prob.slot_off[h]) + 1
    m, mi, e = np.zeros(n, np.int32), np.zeros(ns, np.int32), np.zeros(n, np.int32)
    nm, nmi, ne = C.c_int32(), C.c_int32(), C.c_int32()
    obs = np.array(list(obs_ids) or [0], np.int32)
    p = prob.struct()
    oracle.L.osp_quant_match(C.byref(p), int(h), obs.ctypes.data_as(C.c_void_p), len(obs_ids), m.ctypes.data_as(C.c_void_p), C.byref(nm),
                             mi.ctypes.data_as(C.c_void_p), C.byref(nmi), e.ctypes.data_as(C.c_void_p), C.byref(ne))
    return m[:nm.value].tolist(), mi[:nmi.value].tolist(), e[:ne.value].tolist()


def inexact_haplotype(oracle, prob, h, obs_ids):
    """derive_inexact_haplotype (src/diplotyper.rs:1516-1550): (name, sorted set of (variant name, is_core, relationship))"""
    m, mi, e = quant_match(oracle, prob, h, obs_ids)
    rel = set()
    for ids, state in ((m, "Match"), (mi, "Missing"), (e, "Unexpected")):
        for v in ids:
            meta = prob.var_meta[v]
            rel.add((meta["name"], meta["is_core_variant"], state))
    return (prob.haps[h]["name"], frozenset(rel))


def call_gene(oracle, prob, solver=None):
    """solve + the packaging of call_diplotypes (src/diplotyper.rs:130-204): returns dict(diplotypes, simple, inexact)"""
    score, dips = (solver or oracle_solve)(oracle, prob) if solver is None else solver(prob)
    hap_name = lambda h: prob.sv_labels[-h - 2] if h < 0 else prob.haps[h]["name"]       # an SV label is encoded as -(label + 2)
    names = lambda d: (hap_name(d[0]), hap_name(d[1]))
    core = prob.core_lookup
    main = [names(d) for d in dips]

    def side(h, obs_ids):
        if h >= 0:
            return inexact_haplotype(oracle, prob, h, obs_ids)
        # the SV short-circuit (:1414-1431): the first label names the haplotype, the other labels are unexpected core variants
        labels = [prob.var_list[v][4][3] for v in obs_ids if len(prob.var_list[v]) == 5]
        assert labels[0] == hap_name(h)
        return (labels[0], frozenset((l, True, "Unexpected") for l in labels[1:]))
    ext = []
    for d in dips:
        h1, h2 = het_split(prob, d[2]) if any(g != 4 for g in prob.obs_gt[:len(prob.obs)]) else (
            [int(v) for v in prob.obs_var[:len(prob.obs)]], [int(v) for v in prob.obs_var[:len(prob.obs)]])
        ext.append((side(d[0], h1), side(d[1], h2)))
    if score == (0, 0, 0, 0):
        return {"score": score, "diplotypes": main, "simple": [(core[a], core[b]) for a, b in main], "inexact": None}
    if score[:2] == (0, 0):
        simple = [(core[a], core[b]) for a, b in main]
        return {"score": score, "diplotypes": simple, "simple": simple, "inexact": ext}
    return {"score": score, "diplotypes": [("NO_MATCH", "NO_MATCH")], "simple": [("NO_MATCH", "NO_MATCH")], "inexact": ext}


def load_case(oracle, db_name, vcf_key, with_reference, sv_vcf_key=None, is_deletion=None):
    """is_deletion(defs, start, end) -> label | None decides the SV records of sv_vcf_key (default: the oracle's osp_is_deletion)"""
    db = json.load(open(os.path.join(GOLDEN, "variant_dbs", db_name + ".json")))
    gene_name = sorted(db["gene_entries"])[0]
    gene = db["gene_entries"][gene_name]
    genome = json.load(open(os.path.join(GOLDEN, "test_reference.json"))) if with_reference else None
    vh, haps = load_database_haplotypes(oracle, gene, genome)
    vcf = json.load(open(os.path.join(GOLDEN, "variant_vcfs.json")))[vcf_key]
    obs = load_vcf_variants(oracle, vcf, vh, genome)
    svs = gene.get("structural_variants")
    if sv_vcf_key is not None and svs:
        import __graft_entry__ as ge
        defs = ge.load_package().ffi.SvDefinitions(db["gene_collection"]["gene_dict"], svs)       # pure-Python flattening
        decide = (lambda s, e: is_deletion(defs, s, e)) if is_deletion else (lambda s, e: oracle_is_deletion(oracle, defs, s, e))
        sv_vcf = json.load(open(os.path.join(GOLDEN, "variant_vcfs.json")))[sv_vcf_key]
        obs.update(load_sv_vcf_variants(decide, sv_vcf, svs, db["gene_collection"]["gene_dict"]))
    return gene_name, Problem(vh, haps, obs, svs)
