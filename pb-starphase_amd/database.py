"""ctypes binding of the database / result-file part of include/starphase_hip.h (sp_database_*, sp_variant_gene_*, sp_result_*,
sp_gene_details_*).  Host only: none of these calls needs a device.  No logic lives here."""
import ctypes as C

import numpy as np

from . import ffi
from .ffi import SP_OK, StarphaseError, sp_hla_db_desc, sp_cyp_locus, sp_cyp_gene_def, sp_cyp_config, sp_variant_problem, sp_sv_definitions

_vp, _u32, _i32, _u64, _i64, _s = C.c_void_p, C.c_uint32, C.c_int32, C.c_uint64, C.c_int64, C.c_char_p


class sp_database_metadata(C.Structure):
    _fields_ = [(k, _s) for k in ("pbstarphase_version", "cpic_version", "hla_version", "pharmvar_version", "build_time")]


class sp_database_stats(C.Structure):
    _fields_ = [(k, _u32) for k in ("n_gene_entries", "n_hla_sequences", "n_hla_genes", "n_cyp2d6_alleles", "n_collection_genes")] + \
               [(k, _i32) for k in ("has_hla_config", "has_cyp2d6_config", "reserved")]


class sp_gene_region(C.Structure):
    _fields_ = [("name", _s), ("chrom", _s), ("start", _u64), ("end", _u64), ("is_forward_strand", _i32), ("is_absent_capable", _i32),
                ("n_exons", _u32), ("reserved", _u32), ("exon_start", C.POINTER(_u64)), ("exon_end", C.POINTER(_u64))]


class sp_variant_gene_stats(C.Structure):
    _fields_ = [(k, _u32) for k in ("n_haplotypes", "n_variants", "n_skipped_haplotypes", "n_full_deletions", "n_partial_deletions", "reserved")]


class sp_vcf_allele(C.Structure):
    _fields_ = [("position", _u64), ("ref", _s), ("alt", _s), ("gt", _i32), ("reserved", _i32), ("ps", _i64)]


class sp_vcf_deletion(C.Structure):
    _fields_ = [("start", _u64), ("end", _u64), ("gt", _i32), ("reserved", _i32), ("ps", _i64)]


class sp_variant_detail(C.Structure):
    _fields_ = [("variant_id", _u64), ("variant_name", _s), ("dbsnp", _s), ("chrom", _s), ("position", _u64), ("reference", _s), ("alternate", _s),
                ("sv_label", _s), ("sv_start", _u64), ("sv_end", _u64), ("genotype", _i32), ("is_core_variant", _i32), ("phase_set", _i64)]


class sp_mapping_stats(C.Structure):
    _fields_ = [("present", _i32), ("has_clips", _i32), ("seq_len", _u64), ("nm", _u64), ("unmapped", _u64), ("clipped_start", _u64), ("clipped_end", _u64)]


class sp_detailed_mapping(C.Structure):
    _fields_ = [("present", _i32), ("reserved", _i32), ("query_len", _u64), ("target_len", _u64), ("match_len", _u64), ("nm", _u64),
                ("query_unmapped", _u64), ("target_unmapped", _u64), ("cigar", _s), ("md", _s)]


SUBALLELE_MATCH, CORE_MATCH, INEXACT_DIPLOTYPES, FROM_MAPPINGS, FROM_MULTI_MAPPINGS, NO_MATCH = range(6)
_bound = False


def _lib():
    global _bound
    L = ffi.lib()
    if _bound:
        return L
    P = C.POINTER
    sigs = {
        "sp_database_load": (_i32, [_s, P(_vp), _s, _u32]),
        "sp_database_parse": (_i32, [_s, _u64, P(_vp), _s, _u32]),
        "sp_database_free": (None, [_vp]),
        "sp_database_last_error": (_s, [_vp]),
        "sp_database_get_metadata": (_i32, [_vp, P(sp_database_metadata)]),
        "sp_database_info": (_i32, [_vp, P(sp_database_stats)]),
        "sp_database_hla_gene": (_i32, [_vp, _u32, P(sp_gene_region)]),
        "sp_database_gene_entry": (_i32, [_vp, _u32, P(_s), P(_s)]),
        "sp_database_hla_flatten": (_i32, [_vp, _u32, P(_s), P(_s), _i32, P(sp_hla_db_desc)]),
        "sp_database_hla_allele": (_i32, [_vp, _u32, P(_s), P(_s), P(_s)]),
        "sp_database_cyp_window": (_i32, [_vp, P(_s), P(_u64), P(_u64)]),
        "sp_database_cyp_flatten": (_i32, [_vp, _s, _u64, _u64, P(sp_cyp_locus), P(sp_cyp_gene_def), P(sp_cyp_config)]),
        "sp_variant_gene_create": (_i32, [_vp, _s, _s, _u64, P(_vp)]),
        "sp_variant_gene_free": (None, [_vp]),
        "sp_variant_gene_info": (_i32, [_vp, P(sp_variant_gene_stats)]),
        "sp_variant_gene_haplotype": (_i32, [_vp, _u32, P(_s), P(_s)]),
        "sp_variant_gene_variant": (_i32, [_vp, _u32, P(_u64), P(_s), P(_s), P(_s), P(_s), P(_i64), P(_i32)]),
        "sp_variant_gene_sv_definitions": (_i32, [_vp, P(sp_sv_definitions)]),
        "sp_variant_gene_sv_label": (_i32, [_vp, _i32, _i32, P(_s)]),
        "sp_variant_gene_problem": (_i32, [_vp, _u32, P(sp_vcf_allele), _u32, P(sp_vcf_deletion), _u64, P(sp_variant_problem)]),
        "sp_variant_gene_problem_variant": (_i32, [_vp, _i32, P(_i32), P(_s), P(_u64), P(_u64)]),
        "sp_variant_gene_problem_sv_label": (_i32, [_vp, _i32, P(_s)]),
        "sp_variant_gene_last_error": (_s, [_vp]),
        "sp_result_create": (_i32, [_vp, _s, P(_vp)]),
        "sp_result_free": (None, [_vp]),
        "sp_result_last_error": (_s, [_vp]),
        "sp_gene_details_create": (_i32, [P(_vp)]),
        "sp_gene_details_free": (None, [_vp]),
        "sp_gene_details_add_diplotype": (_i32, [_vp, _s, _s]),
        "sp_gene_details_add_simple_diplotype": (_i32, [_vp, _s, _s]),
        "sp_gene_details_set_simple_diplotypes": (_i32, [_vp, _i32]),
        "sp_gene_details_add_inexact_diplotype": (_i32, [_vp, _s, _u32, P(_s), _vp, _vp, _s, _u32, P(_s), _vp, _vp]),
        "sp_gene_details_add_diplotype_only": (_i32, [_vp, _s, _s]),
        "sp_gene_details_add_variant": (_i32, [_vp, P(sp_variant_detail)]),
        "sp_gene_details_add_mapping": (_i32, [_vp, _s, _s, _s, P(sp_mapping_stats), P(sp_mapping_stats), _i32]),
        "sp_gene_details_add_multi_mapping": (_i32, [_vp, _s, _u64, _u64, _u64, _s]),
        "sp_result_insert": (_i32, [_vp, _s, _vp, _i32]),
        "sp_result_json": (_i32, [_vp, P(_s), P(_u64)]),
        "sp_result_save": (_i32, [_vp, _s]),
        "sp_result_pharmcat_tsv": (_i32, [_vp, P(_s), P(_u64)]),
        "sp_result_save_pharmcat_tsv": (_i32, [_vp, _s]),
        "sp_aln_strings": (_i32, [P(ffi.sp_aln), _vp, _s, _u64, _s, _u32, _s, _u32, P(_u64)]),
        "sp_hla_debug_create": (_i32, [P(_vp)]),
        "sp_hla_debug_free": (None, [_vp]),
        "sp_hla_debug_last_error": (_s, [_vp]),
        "sp_hla_debug_add_read": (_i32, [_vp, _s, _s, _s, _s]),
        "sp_hla_debug_add_mapping": (_i32, [_vp, _s, _s, _s, P(sp_detailed_mapping), P(sp_detailed_mapping)]),
        "sp_hla_debug_add_dual_stats": (_i32, [_vp, _s, P(ffi.sp_hla_call)]),
        "sp_hla_debug_json": (_i32, [_vp, P(_s), P(_u64)]),
        "sp_hla_debug_save": (_i32, [_vp, _s]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _bound = True
    return L


def _b(x):
    return None if x is None else (x if isinstance(x, bytes) else str(x).encode())


def _d(x):
    return None if x is None else x.decode()


class Database:
    """sp_database: one database file (``.json`` / ``.json.gz`` path, or the bytes themselves)."""

    def __init__(self, source):
        self._h = _vp()
        err = C.create_string_buffer(512)
        if isinstance(source, (bytes, bytearray)):
            rc = _lib().sp_database_parse(bytes(source), len(source), C.byref(self._h), err, 512)
        else:
            rc = _lib().sp_database_load(_b(source), C.byref(self._h), err, 512)
        if rc != SP_OK:
            raise StarphaseError(rc, err.value.decode())

    def close(self):
        if self._h:
            _lib().sp_database_free(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != SP_OK:
            raise StarphaseError(rc, _lib().sp_database_last_error(self._h).decode())

    @property
    def metadata(self):
        m = sp_database_metadata()
        self._check(_lib().sp_database_get_metadata(self._h, C.byref(m)))
        return {k: getattr(m, k).decode() for k, _ in m._fields_}

    @property
    def stats(self):
        s = sp_database_stats()
        self._check(_lib().sp_database_info(self._h, C.byref(s)))
        return s

    def hla_genes(self):
        out = []
        for g in range(self.stats.n_hla_genes):
            r = sp_gene_region()
            self._check(_lib().sp_database_hla_gene(self._h, g, C.byref(r)))
            out.append(dict(name=r.name.decode(), chrom=r.chrom.decode(), start=r.start, end=r.end, is_forward_strand=bool(r.is_forward_strand),
                            is_absent_capable=bool(r.is_absent_capable), exons=[(r.exon_start[e], r.exon_end[e]) for e in range(r.n_exons)]))
        return out

    def gene_entries(self):
        out = []
        for i in range(self.stats.n_gene_entries):
            a, b = _s(), _s()
            self._check(_lib().sp_database_gene_entry(self._h, i, C.byref(a), C.byref(b)))
            out.append((a.value.decode(), b.value.decode()))
        return out

    def hla_flatten(self, gene_refs, genes=None, ref_buffer=100):
        """gene_refs: the hg38 bases of [start - ref_buffer, end + ref_buffer) per gene (in the order of `genes`, default: every gene of
        hla_config in name order).  Returns the sp_hla_db_desc (pointing into this database) and the allele names."""
        n = len(gene_refs)
        names = ffi._strs(genes) if genes is not None else None
        refs = ffi._strs(gene_refs)
        desc = sp_hla_db_desc()
        self._check(_lib().sp_database_hla_flatten(self._h, n, names, refs, ref_buffer, C.byref(desc)))
        alleles = []
        for i in range(desc.n_alleles):
            a, b, c = _s(), _s(), _s()
            self._check(_lib().sp_database_hla_allele(self._h, i, C.byref(a), C.byref(b), C.byref(c)))
            alleles.append((a.value.decode(), b.value.decode(), c.value.decode()))
        return desc, alleles

    def hla_db(self, ctx, gene_refs, genes=None, ref_buffer=100):
        """sp_database_hla_flatten + sp_hla_db_create -> (HlaDb, allele names)"""
        desc, alleles = self.hla_flatten(gene_refs, genes, ref_buffer)
        return ffi.HlaDb.from_desc(ctx, desc), alleles

    def cyp_window(self):
        c, a, b = _s(), _u64(), _u64()
        self._check(_lib().sp_database_cyp_window(self._h, C.byref(c), C.byref(a), C.byref(b)))
        return c.value.decode(), a.value, b.value

    def cyp_flatten(self, chrom_seq, window_start):
        self._cyp_seq = _b(chrom_seq)
        L, G, K = sp_cyp_locus(), sp_cyp_gene_def(), sp_cyp_config()
        self._check(_lib().sp_database_cyp_flatten(self._h, self._cyp_seq, int(window_start), len(self._cyp_seq), C.byref(L), C.byref(G), C.byref(K)))
        return L, G, K

    def cyp_db(self, ctx, chrom_seq, window_start):
        """sp_database_cyp_flatten + sp_cyp_db_create -> CypDb (ctx None: host tables only)"""
        L, G, K = self.cyp_flatten(chrom_seq, window_start)
        return ffi.CypDb.from_structs(ctx, L, G, K, keep=[self])

    def variant_gene(self, gene_name, chrom_seq=None):
        return VariantGene(self, gene_name, chrom_seq)


class VariantGene:
    """sp_variant_gene: one gene entry, normalised (load_database_haplotypes)"""

    def __init__(self, db, gene_name, chrom_seq=None):
        self.db = db
        self._seq = _b(chrom_seq)
        self._h = _vp()
        db._check(_lib().sp_variant_gene_create(db._h, _b(gene_name), self._seq, len(self._seq) if self._seq else 0, C.byref(self._h)))
        self._keep = None

    def close(self):
        if self._h:
            _lib().sp_variant_gene_free(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != SP_OK:
            raise StarphaseError(rc, _lib().sp_variant_gene_last_error(self._h).decode())

    @property
    def stats(self):
        s = sp_variant_gene_stats()
        self._check(_lib().sp_variant_gene_info(self._h, C.byref(s)))
        return s

    def haplotypes(self):
        out = []
        for h in range(self.stats.n_haplotypes):
            a, b = _s(), _s()
            self._check(_lib().sp_variant_gene_haplotype(self._h, h, C.byref(a), C.byref(b)))
            out.append((a.value.decode(), _d(b.value)))
        return out

    def variants(self):
        out = []
        for v in range(self.stats.n_variants):
            pos, ref, alt, name, rs, vid, core = _u64(), _s(), _s(), _s(), _s(), _i64(), _i32()
            self._check(_lib().sp_variant_gene_variant(self._h, v, C.byref(pos), C.byref(ref), C.byref(alt), C.byref(name), C.byref(rs), C.byref(vid), C.byref(core)))
            out.append(dict(position=pos.value, ref=ref.value.decode(), alt=alt.value.decode(), name=name.value.decode(), dbsnp_id=_d(rs.value),
                            variant_id=vid.value, is_core_variant=bool(core.value)))
        return out

    def sv_definitions(self):
        d = sp_sv_definitions()
        self._check(_lib().sp_variant_gene_sv_definitions(self._h, C.byref(d)))
        return d

    def is_deletion(self, start, end):
        d = self.sv_definitions()
        kind, index = _i32(0), _i32(-1)
        rc = ffi.lib().sp_variant_is_deletion(C.byref(d), int(start), int(end), C.byref(kind), C.byref(index))
        if rc != SP_OK:
            raise StarphaseError(rc, "sp_variant_is_deletion")
        if kind.value == 0:
            return None
        s = _s()
        self._check(_lib().sp_variant_gene_sv_label(self._h, kind.value, index.value, C.byref(s)))
        return s.value.decode()

    def problem(self, alleles=(), deletions=(), max_sv_length=0):
        """alleles: (position0, ref, alt, gt, ps|None) per ALT allele of a record; deletions: (start, end, gt, ps|None).
        Returns the sp_variant_problem (valid until the next call)."""
        A = (sp_vcf_allele * max(1, len(alleles)))()
        keep = []
        for i, (pos, ref, alt, gt, ps) in enumerate(alleles):
            keep += [_b(ref), _b(alt)]
            A[i] = sp_vcf_allele(int(pos), keep[-2], keep[-1], int(gt), 0, -1 if ps is None else int(ps))
        D = (sp_vcf_deletion * max(1, len(deletions)))()
        for i, (s, e, gt, ps) in enumerate(deletions):
            D[i] = sp_vcf_deletion(int(s), int(e), int(gt), 0, -1 if ps is None else int(ps))
        p = sp_variant_problem()
        self._check(_lib().sp_variant_gene_problem(self._h, len(alleles), A, len(deletions), D, int(max_sv_length), C.byref(p)))
        return p

    def problem_variant(self, vid):
        k, s, a, b = _i32(), _s(), _u64(), _u64()
        self._check(_lib().sp_variant_gene_problem_variant(self._h, int(vid), C.byref(k), C.byref(s), C.byref(a), C.byref(b)))
        return k.value, _d(s.value), a.value, b.value

    def problem_sv_label(self, label_id):
        s = _s()
        self._check(_lib().sp_variant_gene_problem_sv_label(self._h, int(label_id), C.byref(s)))
        return s.value.decode()


def problem_arrays(p):
    """the arrays of an sp_variant_problem as Python lists (tests compare them)"""
    def arr(ptr, n, t):
        return list(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(t)), (n,))) if n else []
    n_slots = arr(p.slot_off, p.n_haps + 1, C.c_int32)[-1] if p.n_haps >= 0 else 0
    alt_off = arr(p.alt_off, n_slots + 1, C.c_int32)
    return dict(n_haps=p.n_haps, hap_is_sv=arr(p.hap_is_sv, p.n_haps, C.c_uint8), hap_is_core=arr(p.hap_is_core, p.n_haps, C.c_uint8),
                slot_off=arr(p.slot_off, p.n_haps + 1, C.c_int32), alt_off=alt_off, alt_var=arr(p.alt_var, alt_off[-1], C.c_int32),
                n_vars=p.n_vars, var_is_core=arr(p.var_is_core, p.n_vars, C.c_uint8), n_obs=p.n_obs, obs_var=arr(p.obs_var, p.n_obs, C.c_int32),
                obs_gt=arr(p.obs_gt, p.n_obs, C.c_int32), obs_ps=arr(p.obs_ps, p.n_obs, C.c_int64), obs_sv_label=arr(p.obs_sv_label, p.n_obs, C.c_int32))


class GeneDetails:
    """sp_gene_details: the parts of one PgxGeneDetails"""

    def __init__(self):
        self._h = _vp()
        _lib().sp_gene_details_create(C.byref(self._h))

    def __del__(self):
        try:
            if self._h:
                _lib().sp_gene_details_free(self._h)
                self._h = _vp()
        except Exception:
            pass

    def add_diplotype(self, h1, h2):
        _lib().sp_gene_details_add_diplotype(self._h, _b(h1), _b(h2)); return self

    def add_simple_diplotype(self, h1, h2):
        _lib().sp_gene_details_add_simple_diplotype(self._h, _b(h1), _b(h2)); return self

    def set_simple_diplotypes(self, some):
        _lib().sp_gene_details_set_simple_diplotypes(self._h, 1 if some else 0); return self

    def add_inexact_diplotype(self, hap1, hap2):
        """hap = (base haplotype, [(label, is_vi, state), ...])"""
        args = []
        for base, rel in (hap1, hap2):
            labels = ffi._strs([r[0] for r in rel])
            vi = np.array([1 if r[1] else 0 for r in rel] or [0], np.uint8)
            st = np.array([r[2] for r in rel] or [0], np.int32)
            args += [_b(base), len(rel), labels, vi, st]
        a = args
        rc = _lib().sp_gene_details_add_inexact_diplotype(self._h, a[0], a[1], a[2], ffi._ptr(a[3]), ffi._ptr(a[4]), a[5], a[6], a[7], ffi._ptr(a[8]), ffi._ptr(a[9]))
        assert rc == SP_OK
        return self

    def add_diplotype_only(self, h1, h2):
        _lib().sp_gene_details_add_diplotype_only(self._h, _b(h1), _b(h2)); return self

    def add_variant(self, variant_id, name, dbsnp, chrom, position, ref, alt, genotype, phase_set=None, is_core=True, sv=None):
        v = sp_variant_detail(int(variant_id), _b(name), _b(dbsnp), _b(chrom), int(position), _b(ref), _b(alt), _b(sv[2]) if sv else None,
                              int(sv[0]) if sv else 0, int(sv[1]) if sv else 0, int(genotype), 1 if is_core else 0, -1 if phase_set is None else int(phase_set))
        rc = _lib().sp_gene_details_add_variant(self._h, C.byref(v))
        if rc != SP_OK:
            raise StarphaseError(rc, "sp_gene_details_add_variant")
        return self

    @staticmethod
    def _stats(s):
        if s is None:
            return None
        seq_len, nm, unmapped = s[:3]
        clips = s[3:] if len(s) > 3 and s[3] is not None else None
        return sp_mapping_stats(1, 1 if clips else 0, int(seq_len), int(nm), int(unmapped), int(clips[0]) if clips else 0, int(clips[1]) if clips else 0)

    def add_mapping(self, qname, hla_id, star, cdna=None, dna=None, is_ignored=False):
        c, d = self._stats(cdna), self._stats(dna)
        _lib().sp_gene_details_add_mapping(self._h, _b(qname), _b(hla_id), _b(star), C.byref(c) if c else None, C.byref(d) if d else None, 1 if is_ignored else 0)
        return self

    def add_multi_mapping(self, qname, start, end, consensus_id, star):
        _lib().sp_gene_details_add_multi_mapping(self._h, _b(qname), int(start), int(end), int(consensus_id), _b(star)); return self


class Result:
    """sp_result: StarphaseJson"""

    def __init__(self, db=None, version=""):
        self._h = _vp()
        _lib().sp_result_create(db._h if db is not None else None, _b(version), C.byref(self._h))

    def __del__(self):
        try:
            if self._h:
                _lib().sp_result_free(self._h)
                self._h = _vp()
        except Exception:
            pass

    def _check(self, rc):
        if rc != SP_OK:
            raise StarphaseError(rc, _lib().sp_result_last_error(self._h).decode())

    def insert(self, gene, details, constructor):
        self._check(_lib().sp_result_insert(self._h, _b(gene), details._h if details is not None else None, int(constructor)))

    def json(self):
        s, n = _s(), _u64()
        self._check(_lib().sp_result_json(self._h, C.byref(s), C.byref(n)))
        return s.value.decode()

    def save(self, path):
        self._check(_lib().sp_result_save(self._h, _b(path)))

    def pharmcat_tsv(self):
        s, n = _s(), _u64()
        self._check(_lib().sp_result_pharmcat_tsv(self._h, C.byref(s), C.byref(n)))
        return s.value.decode()

    def save_pharmcat_tsv(self, path):
        self._check(_lib().sp_result_save_pharmcat_tsv(self._h, _b(path)))


# ------------------------------------------------------------------ debug files (sp_hla_debug_*, sp_aln_strings)
def aln_strings(aln, events, target):
    """sp_aln_strings: one row of Context.align_batch(..., events=True) (A = query, B = target) -> (cigar, md, match_len)"""
    a = ffi.sp_aln(*[int(aln[k]) for k in ("ok", "nm", "a_start", "a_end", "b_start", "b_end", "a_len", "b_len")])
    ev = np.ascontiguousarray(events, np.uint32)
    cap = 16 * (int(aln["nm"]) + 2) + 32
    cg, md, ml = C.create_string_buffer(cap), C.create_string_buffer(cap), _u64()
    rc = _lib().sp_aln_strings(C.byref(a), ev.ctypes.data, _b(target), len(target), cg, cap, md, cap, C.byref(ml))
    if rc != SP_OK:
        raise StarphaseError(rc, "sp_aln_strings")
    return cg.value.decode(), md.value.decode(), ml.value


def detailed_mapping(aln, events, target):
    """DetailedMappingStats::from_mapping for one alignment row: a dict with the fields of sp_detailed_mapping"""
    cg, md, ml = aln_strings(aln, events, target)
    return dict(query_len=int(aln["a_len"]), target_len=int(aln["b_len"]), match_len=ml, nm=int(aln["nm"]),
                query_unmapped=int(aln["a_len"]) - (int(aln["a_end"]) - int(aln["a_start"])),
                target_unmapped=int(aln["b_len"]) - (int(aln["b_end"]) - int(aln["b_start"])), cigar=cg, md=md)


class HlaDebug:
    """sp_hla_debug: hla_debug.json"""

    def __init__(self):
        self._h = _vp()
        _lib().sp_hla_debug_create(C.byref(self._h))

    def close(self):
        if self._h:
            _lib().sp_hla_debug_free(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != SP_OK:
            raise StarphaseError(rc, _lib().sp_hla_debug_last_error(self._h).decode())

    @staticmethod
    def _dm(m):
        if m is None:
            return None, None
        keep = (_b(m["cigar"]), _b(m["md"]))
        return sp_detailed_mapping(1, 0, m["query_len"], m["target_len"], m["match_len"], m["nm"], m["query_unmapped"], m["target_unmapped"], *keep), keep

    def add_read(self, gene, qname, best_id=None, best_star=None):
        self._check(_lib().sp_hla_debug_add_read(self._h, _b(gene), _b(qname), _b(best_id), _b(best_star))); return self

    def add_mapping(self, gene, qname, hla_id, cdna=None, dna=None):
        c, _k1 = self._dm(cdna)
        d, _k2 = self._dm(dna)
        self._check(_lib().sp_hla_debug_add_mapping(self._h, _b(gene), _b(qname), _b(hla_id), C.byref(c) if c else None, C.byref(d) if d else None)); return self

    def add_dual_stats(self, gene, call):
        self._check(_lib().sp_hla_debug_add_dual_stats(self._h, _b(gene), C.byref(call))); return self

    def json(self):
        s, n = _s(), _u64()
        self._check(_lib().sp_hla_debug_json(self._h, C.byref(s), C.byref(n)))
        return s.value.decode()

    def save(self, path):
        self._check(_lib().sp_hla_debug_save(self._h, _b(path)))


# ------------------------------------------------------------------ input files (sp_bam_*, sp_vcf_*)
class sp_bam_read(C.Structure):
    _fields_ = [("qname", _s), ("flag", _u32), ("mapq", _u32), ("ref_id", _i32), ("reserved", _i32), ("pos", _i64), ("end", _i64),
                ("l_seq", _u32), ("n_cigar", _u32), ("cigar", C.POINTER(_u32))]


_io_bound = False


def _io():
    global _io_bound
    L = _lib()
    if _io_bound:
        return L
    P = C.POINTER
    sigs = {
        "sp_bam_open": (_i32, [_s, P(_vp), _s, _u32]),
        "sp_bam_free": (None, [_vp]),
        "sp_bam_last_error": (_s, [_vp]),
        "sp_bam_references": (_i32, [_vp, P(_u32), P(P(_s)), P(P(_u64))]),
        "sp_bam_fetch": (_i32, [_vp, _s, _u64, _u64, _u32, _i32, P(P(sp_bam_read)), P(_u32), P(_vp), P(P(_u64))]),
        "sp_bam_forget": (_i32, [_vp]),
        "sp_bam_last_seq4": (_i32, [_vp, P(_vp), P(P(_u64)), P(P(_u32)), P(_u32)]),
        "sp_vcf_open": (_i32, [_s, P(_vp), _s, _u32]),
        "sp_vcf_free": (None, [_vp]),
        "sp_vcf_index_info": (_i32, [_vp, P(_i32), P(_u64)]),
        "sp_vcf_last_error": (_s, [_vp]),
        "sp_vcf_samples": (_i32, [_vp, P(_u32), P(P(_s))]),
        "sp_vcf_alleles": (_i32, [_vp, _s, _s, _u64, _u64, P(P(sp_vcf_allele)), P(_u32)]),
        "sp_vcf_deletions": (_i32, [_vp, _s, _s, _u64, _u64, P(P(sp_vcf_deletion)), P(_u32)]),
        "sp_fasta_open": (_i32, [_s, P(_vp), _s, _u32]),
        "sp_fasta_free": (None, [_vp]),
        "sp_fasta_last_error": (_s, [_vp]),
        "sp_fasta_sequences": (_i32, [_vp, P(_u32), P(P(_s)), P(P(_u64))]),
        "sp_fasta_fetch": (_i32, [_vp, _s, _u64, _u64, P(_s), P(_u64)]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _io_bound = True
    return L


class Bam:
    """sp_bam: an (indexed) BAM file"""

    def __init__(self, path):
        self._h = _vp()
        err = C.create_string_buffer(512)
        rc = _io().sp_bam_open(_b(path), C.byref(self._h), err, 512)
        if rc != SP_OK:
            raise StarphaseError(rc, err.value.decode())

    def __del__(self):
        try:
            if self._h:
                _io().sp_bam_free(self._h)
                self._h = _vp()
        except Exception:
            pass

    def references(self):
        n, names, lens = _u32(), C.POINTER(_s)(), C.POINTER(_u64)()
        _io().sp_bam_references(self._h, C.byref(n), C.byref(names), C.byref(lens))
        return [(names[i].decode(), int(lens[i])) for i in range(n.value)]

    def fetch(self, chrom, start, end, exclude_flags=0, dedupe=False):
        """-> list of dict(qname, flag, mapq, pos, end, cigar [(op, len)], seq)"""
        reads, n, bases, offs = C.POINTER(sp_bam_read)(), _u32(), _vp(), C.POINTER(_u64)()
        rc = _io().sp_bam_fetch(self._h, _b(chrom), int(start), int(end), int(exclude_flags), 1 if dedupe else 0, C.byref(reads), C.byref(n), C.byref(bases),
                                C.byref(offs))
        if rc != SP_OK:
            raise StarphaseError(rc, _io().sp_bam_last_error(self._h).decode())
        blob = C.string_at(bases.value, int(offs[n.value])) if n.value else b""
        out = []
        for i in range(n.value):
            r = reads[i]
            out.append(dict(qname=r.qname.decode(), flag=r.flag, mapq=r.mapq, pos=r.pos, end=r.end,
                            cigar=[(r.cigar[k] & 15, r.cigar[k] >> 4) for k in range(r.n_cigar)], seq=blob[offs[i]:offs[i + 1]].decode()))
        return out

    def last_seq4(self):
        """the SEQ fields of the last fetch as the file stores them -> (bytes as uint8 array, byte offsets[n + 1], lengths[n]): the arguments
        of Context.upload_format(SP_SEQ_BAM4, ...)"""
        import numpy as np
        seq, off, ln, n = _vp(), C.POINTER(_u64)(), C.POINTER(_u32)(), _u32()
        rc = _io().sp_bam_last_seq4(self._h, C.byref(seq), C.byref(off), C.byref(ln), C.byref(n))
        if rc != SP_OK:
            raise StarphaseError(rc, "sp_bam_last_seq4")
        k = n.value
        offs = np.array([off[i] for i in range(k + 1)], np.uint64) if k else np.zeros(1, np.uint64)
        blob = np.frombuffer(C.string_at(seq.value, int(offs[-1])), np.uint8).copy() if k and offs[-1] else np.zeros(0, np.uint8)
        return blob, offs, np.array([ln[i] for i in range(k)], np.uint32)

    def forget(self):
        _io().sp_bam_forget(self._h)


class Vcf:
    """sp_vcf: a VCF file (plain, gzip or bgzip)"""

    def __init__(self, path):
        self._h = _vp()
        err = C.create_string_buffer(512)
        rc = _io().sp_vcf_open(_b(path), C.byref(self._h), err, 512)
        if rc != SP_OK:
            raise StarphaseError(rc, err.value.decode())

    def __del__(self):
        try:
            if self._h:
                _io().sp_vcf_free(self._h)
                self._h = _vp()
        except Exception:
            pass

    def _check(self, rc):
        if rc != SP_OK:
            raise StarphaseError(rc, _io().sp_vcf_last_error(self._h).decode())

    def index_info(self):
        """-> (read through a tabix / CSI index?, record lines the region fetches have parsed so far)"""
        ix, n = _i32(), _u64()
        _io().sp_vcf_index_info(self._h, C.byref(ix), C.byref(n))
        return bool(ix.value), int(n.value)

    def samples(self):
        n, names = _u32(), C.POINTER(_s)()
        _io().sp_vcf_samples(self._h, C.byref(n), C.byref(names))
        return [names[i].decode() for i in range(n.value)]

    def alleles(self, chrom, start=0, end=2 ** 62, sample=None):
        """-> [(position0, ref, alt, gt, ps|None)], the rows VariantGene.problem takes"""
        out, n = C.POINTER(sp_vcf_allele)(), _u32()
        self._check(_io().sp_vcf_alleles(self._h, _b(sample), _b(chrom), int(start), int(end), C.byref(out), C.byref(n)))
        return [(out[i].position, out[i].ref.decode(), out[i].alt.decode(), out[i].gt, None if out[i].ps < 0 else out[i].ps) for i in range(n.value)]

    def deletions(self, chrom, start=0, end=2 ** 62, sample=None):
        out, n = C.POINTER(sp_vcf_deletion)(), _u32()
        self._check(_io().sp_vcf_deletions(self._h, _b(sample), _b(chrom), int(start), int(end), C.byref(out), C.byref(n)))
        return [(out[i].start, out[i].end, out[i].gt, None if out[i].ps < 0 else out[i].ps) for i in range(n.value)]


class Fasta:
    """sp_fasta: the reference FASTA (plain with or without .fai, gzip)"""

    def __init__(self, path):
        self._h = _vp()
        err = C.create_string_buffer(512)
        rc = _io().sp_fasta_open(_b(path), C.byref(self._h), err, 512)
        if rc != SP_OK:
            raise StarphaseError(rc, err.value.decode())

    def __del__(self):
        try:
            if self._h:
                _io().sp_fasta_free(self._h)
                self._h = _vp()
        except Exception:
            pass

    def sequences(self):
        n, names, lens = _u32(), C.POINTER(_s)(), C.POINTER(_u64)()
        _io().sp_fasta_sequences(self._h, C.byref(n), C.byref(names), C.byref(lens))
        return [(names[i].decode(), lens[i]) for i in range(n.value)]

    def fetch(self, chrom, start, end):
        b, n = _s(), _u64()
        rc = _io().sp_fasta_fetch(self._h, _b(chrom), int(start), int(end), C.byref(b), C.byref(n))
        if rc != SP_OK:
            raise StarphaseError(rc, _io().sp_fasta_last_error(self._h).decode())
        return C.string_at(b, n.value).decode()
