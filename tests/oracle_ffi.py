"""ctypes access to oracle/liboracle.so -- the CPU oracle (TEST INFRASTRUCTURE ONLY: tests/, smoke(), bench
cpu_baseline).  Builds it on demand with gcc."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")


class Aln(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("ok", "nm", "a_start", "a_end", "b_start", "b_end", "a_len", "b_len")]


ALN_DTYPE = np.dtype([(n, np.int32) for n in ("ok", "nm", "a_start", "a_end", "b_start", "b_end", "a_len", "b_len")])


class Mapping(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("query_len", "query_start", "query_end", "target_len", "target_start",
                                          "target_end", "nm", "strand_fwd")]


class HlaLevel(C.Structure):
    _fields_ = [("present", C.c_int32), ("range_start", C.c_int32), ("range_end", C.c_int32),
                ("len", C.c_int32), ("nm", C.c_int32), ("unmapped", C.c_int32), ("pc", C.POINTER(C.c_uint64))]


class HlaScoreProblem(C.Structure):
    _fields_ = [("cons", C.c_void_p * 2), ("cons_len", C.c_int32 * 2), ("n_alleles", C.c_int32),
                ("seq", C.c_void_p * 2), ("seq_len", C.c_void_p * 2), ("diag", C.c_void_p * 2), ("max_ed", C.c_int32)]


class Oracle:
    def __init__(self, lib):
        self.L = lib
        L = lib
        vp, i32, u64, dbl = C.c_void_p, C.c_int32, C.c_uint64, C.c_double
        L.osp_encode.argtypes = [C.c_char_p, C.c_size_t, vp]
        L.osp_wfa.restype = i32
        L.osp_wfa.argtypes = [vp, i32, vp, i32, i32, i32, C.POINTER(Aln), vp, C.POINTER(i32)]
        L.osp_anchor.restype = i32
        L.osp_anchor.argtypes = [vp, i32, vp, i32, C.POINTER(i32)]
        L.osp_events_to_cigar.restype = i32
        L.osp_events_to_cigar.argtypes = [C.POINTER(Aln), vp, i32, vp, i32]
        L.osp_score_value.restype = dbl
        L.osp_score_value.argtypes = [u64, u64, u64]
        L.osp_custom_score.restype = dbl
        L.osp_custom_score.argtypes = [u64, u64, u64, i32]
        L.osp_select_best_mapping.restype = i32
        L.osp_select_best_mapping.argtypes = [C.POINTER(Mapping), i32, i32, i32, C.c_int64, C.POINTER(u64 * 3)]
        L.osp_process_mm_cigar.restype = i32
        L.osp_process_mm_cigar.argtypes = [vp, vp, i32, u64, u64, u64, u64, vp]
        L.osp_is_better_match.restype = i32
        L.osp_is_better_match.argtypes = [C.POINTER(HlaLevel * 2), C.POINTER(HlaLevel * 2)]
        L.osp_hla_score_read.restype = i32
        L.osp_hla_score_read.argtypes = [C.POINTER(HlaScoreProblem), vp, vp]
        L.osp_hla_pick_allele.restype = i32
        L.osp_hla_pick_allele.argtypes = [vp, i32, i32]
        L.osp_is_passing_dual.restype = i32
        L.osp_is_passing_dual.argtypes = [u64, u64, dbl, dbl, dbl, C.POINTER(dbl), C.POINTER(dbl)]
        L.osp_is_hemizygous_better.restype = i32
        L.osp_is_hemizygous_better.argtypes = [vp, vp, vp, i32, i32, u64, i32, dbl, C.POINTER(dbl), C.POINTER(dbl)]
        L.osp_hpc.restype = C.c_size_t
        L.osp_hpc.argtypes = [C.c_char_p, C.c_size_t, vp]
        L.osp_hpc_pos.restype = C.c_size_t
        L.osp_hpc_pos.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t]
        L.osp_revcomp.restype = i32
        L.osp_revcomp.argtypes = [C.c_char_p, C.c_size_t, vp]
        L.osp_ln_factorial.restype = dbl
        L.osp_ln_factorial.argtypes = [u64]
        L.osp_multinomial_ln_pmf.restype = dbl
        L.osp_multinomial_ln_pmf.argtypes = [vp, vp, i32]
        for f in ("osp_binomial_cdf", "osp_binomial_ln_pmf"):
            getattr(L, f).restype = dbl
            getattr(L, f).argtypes = [dbl, u64, u64]
        L.osp_normal_ln_pdf.restype = dbl
        L.osp_normal_ln_pdf.argtypes = [dbl, dbl, dbl]

    # ---- helpers
    def encode(self, s):
        if isinstance(s, str):
            s = s.encode()
        a = np.zeros(max(1, len(s)), np.uint8)
        self.L.osp_encode(s, len(s), a.ctypes.data_as(C.c_void_p))
        return a[:len(s)] if len(s) else a[:0]

    def wfa(self, a, b, diag, max_ed=255, events=True):
        A = a if isinstance(a, np.ndarray) else self.encode(a)
        B = b if isinstance(b, np.ndarray) else self.encode(b)
        al = Aln()
        ev = np.zeros(max_ed + 1, np.uint32)
        ne = C.c_int32(0)
        self.L.osp_wfa(A.ctypes.data_as(C.c_void_p), len(A), B.ctypes.data_as(C.c_void_p), len(B), int(diag), int(max_ed),
                       C.byref(al), ev.ctypes.data_as(C.c_void_p) if events else None, C.byref(ne))
        return al, ev[:ne.value].copy()

    def anchor(self, a, b):
        A = a if isinstance(a, np.ndarray) else self.encode(a)
        B = b if isinstance(b, np.ndarray) else self.encode(b)
        d = C.c_int32(0)
        v = self.L.osp_anchor(A.ctypes.data_as(C.c_void_p), len(A), B.ctypes.data_as(C.c_void_p), len(B), C.byref(d))
        return d.value, v

    def cigar(self, al, ev):
        cg = np.zeros(2 * len(ev) + 4, np.uint32)
        evc = np.ascontiguousarray(ev, np.uint32)
        n = self.L.osp_events_to_cigar(C.byref(al), evc.ctypes.data_as(C.c_void_p), len(evc), cg.ctypes.data_as(C.c_void_p), len(cg))
        return [(int(x >> 4), int(x & 15)) for x in cg[:n]]

    def process_mm_cigar(self, cigar, target_offset, target_len, clip_start, clip_end):
        ln = np.array([c[0] for c in cigar], np.uint32)
        op = np.array([c[1] for c in cigar], np.uint8)
        out = np.zeros(target_len + 1, np.uint64)
        rc = self.L.osp_process_mm_cigar(ln.ctypes.data_as(C.c_void_p), op.ctypes.data_as(C.c_void_p), len(cigar),
                                         target_offset, target_len, clip_start, clip_end, out.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise ValueError(f"process_mm_cigar rc={rc}")
        return out.tolist()

    def hla_score_read(self, cons_cdna, cons_dna, cdna_list, dna_list, diag_cdna, diag_dna, max_ed=255):
        """cdna_list/dna_list: per allele str or None; diag_*: per allele int or None. Returns best, stats[n,2,3], alns[n,2]"""
        n = len(cdna_list)
        keep = []
        p = HlaScoreProblem()
        cons = [self.encode(cons_cdna), self.encode(cons_dna)]
        keep += cons
        for lv in range(2):
            p.cons[lv] = cons[lv].ctypes.data if len(cons[lv]) else None
            p.cons_len[lv] = len(cons[lv])
        p.n_alleles = n
        p.max_ed = max_ed
        for lv, (seqs, diags) in enumerate(((cdna_list, diag_cdna), (dna_list, diag_dna))):
            ptrs = (C.c_void_p * n)()
            lens = np.zeros(n, np.int32)
            dg = np.full(n, -2 ** 31, np.int32)
            for i, s in enumerate(seqs):
                if s:
                    e = self.encode(s)
                    keep.append(e)
                    ptrs[i] = e.ctypes.data
                    lens[i] = len(e)
                if diags[i] is not None:
                    dg[i] = diags[i]
            keep += [ptrs, lens, dg]
            p.seq[lv] = C.cast(ptrs, C.c_void_p)
            p.seq_len[lv] = lens.ctypes.data
            p.diag[lv] = dg.ctypes.data
        stats = np.zeros((n, 2, 3), np.int64)
        alns = np.zeros((n, 2), ALN_DTYPE)
        best = self.L.osp_hla_score_read(C.byref(p), stats.ctypes.data_as(C.c_void_p), alns.ctypes.data_as(C.c_void_p))
        return best, stats, alns

    def pick_allele(self, alns, read_len):
        a = np.ascontiguousarray(alns)
        return self.L.osp_hla_pick_allele(a.ctypes.data_as(C.c_void_p), len(a), int(read_len))

    def splice_read(self, pos, cigar, exons):
        """cigar: list of (len, op) BAM ops; exons: list of (start, end). Returns (segments, offset)"""
        cg = np.array([(l << 4) | op for l, op in cigar], np.uint32)
        es = np.array([e[0] for e in exons], np.int64)
        ee = np.array([e[1] for e in exons], np.int64)
        ss = np.zeros(len(exons), np.int32)
        se = np.zeros(len(exons), np.int32)
        ns = C.c_int32(0)
        off = C.c_int64(0)
        self.L.osp_splice_read.argtypes = [C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                           C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
        self.L.osp_splice_read(int(pos), cg.ctypes.data_as(C.c_void_p), len(cg), es.ctypes.data_as(C.c_void_p), ee.ctypes.data_as(C.c_void_p),
                               len(exons), ss.ctypes.data_as(C.c_void_p), se.ctypes.data_as(C.c_void_p), C.byref(ns), C.byref(off))
        return [(int(ss[i]), int(se[i])) for i in range(ns.value)], off.value

    def hpc(self, s):
        out = np.zeros(max(1, len(s)), np.uint8)
        n = self.L.osp_hpc(s.encode(), len(s), out.ctypes.data_as(C.c_void_p))
        return out[:n].tobytes().decode()

    def hpc_pos(self, s, pos):
        return self.L.osp_hpc_pos(s.encode(), len(s), pos)

    def revcomp(self, s):
        out = C.create_string_buffer(len(s) + 1)
        rc = self.L.osp_revcomp(s.encode(), len(s), out)
        if rc != 0:
            raise ValueError("bad character")
        return out.raw[:len(s)].decode()


_cached = None


def load():
    global _cached
    if _cached is None:
        so = os.path.join(ODIR, "liboracle.so")
        srcs = [os.path.join(ODIR, f) for f in os.listdir(ODIR) if f.endswith((".c", ".h"))]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call(["make", "-s", "-C", ODIR])
        _cached = Oracle(C.CDLL(so))
    return _cached
