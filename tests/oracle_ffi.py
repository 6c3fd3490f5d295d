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

    def wfa(self, a, b, diag, max_ed=255, events=True, retry=False):
        """retry: the rules of the library's generic cell launcher -- True / 1: a cell that finds nothing on 64 diagonals runs again on
        256; 2: also when it needed more than 32 edits, the wide result is kept when it has fewer (sp_align_batch, placements)"""
        A = a if isinstance(a, np.ndarray) else self.encode(a)
        B = b if isinstance(b, np.ndarray) else self.encode(b)
        al = Aln()
        ev = np.zeros(max_ed + 1, np.uint32)
        ne = C.c_int32(0)
        fn = self.L.osp_wfa_retry2 if retry == 2 else self.L.osp_wfa_retry if retry else self.L.osp_wfa
        fn.restype = C.c_int32
        fn.argtypes = self.L.osp_wfa.argtypes
        fn(A.ctypes.data_as(C.c_void_p), len(A), B.ctypes.data_as(C.c_void_p), len(B), int(diag), int(max_ed),
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


class AffineOpts(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("a", "b", "q", "e", "q2", "e2", "sc_ambi")]


class AffineOut(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("score", "nm", "t_start", "t_end", "q_start", "q_end")]


def oracle_affine(oracle, target, query, k0, band=64, a=1):
    """osp_affine_local (oracle/affine.c): target / query as strings or code arrays, k0 = q_pos - t_pos -> (score, nm, t_start, t_end, q_start, q_end)"""
    t = target if isinstance(target, np.ndarray) else oracle.encode(target)
    q = query if isinstance(query, np.ndarray) else oracle.encode(query)
    op, out = AffineOpts(a, 4, 6, 2, 26, 1, 1), AffineOut()
    oracle.L.osp_affine_local.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(AffineOpts), C.POINTER(AffineOut)]
    oracle.L.osp_affine_local.restype = None
    oracle.L.osp_affine_local(t.ctypes.data, len(t), q.ctypes.data, len(q), int(k0), int(band), C.byref(op), C.byref(out))
    return (out.score, out.nm, out.t_start, out.t_end, out.q_start, out.q_end)


def load(path=None):
    """the shipped oracle, or (bench.py's CPU leg) another build of the same sources"""
    global _cached
    if path is not None:
        return Oracle(C.CDLL(path))
    if _cached is None:
        so = os.path.join(ODIR, "liboracle.so")
        srcs = [os.path.join(ODIR, f) for f in os.listdir(ODIR) if f.endswith((".c", ".h"))]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call(["make", "-s", "-C", ODIR])
        _cached = Oracle(C.CDLL(so))
    return _cached


# ------------------------------------------------------------------ CYP2D6 chain search (oracle/cyp.c)
REGION_TYPES = {"UNKNOWN": 0, "REP6": 1, "CYP2D6": 2, "link_region": 3, "REP7": 4, "spacer": 5, "CYP2D7": 6,
                "CYP2D6*5": 7, "Hybrid": 8, "FalseAllele": 9}
OSP_MAX_CHAIN = 64


class CypConfig(C.Structure):
    _fields_ = [("n_translate", C.c_int32), ("tr_key", C.POINTER(C.c_char_p)), ("tr_val", C.POINTER(C.c_char_p)),
                ("n_conn", C.c_int32), ("conn_a", C.POINTER(C.c_char_p)), ("conn_b", C.POINTER(C.c_char_p)),
                ("n_single", C.c_int32), ("singles", C.POINTER(C.c_char_p))]


class ChainProblem(C.Structure):
    _fields_ = [("n_haps", C.c_int32), ("type", C.c_void_p), ("subtype", C.POINTER(C.c_char_p)), ("cfg", CypConfig),
                ("n_reads", C.c_int32), ("read_chain_off", C.c_void_p), ("chain_off", C.c_void_p), ("chain_items", C.c_void_p),
                ("read_w_off", C.c_void_p), ("w_ed", C.c_void_p), ("w_ov", C.c_void_p),
                ("infer_connections", C.c_int32), ("normalize_all_alleles", C.c_int32), ("ignore_chain_label_limits", C.c_int32),
                ("lasso", C.c_double), ("ln_ed", C.c_double), ("unexpected", C.c_double), ("inferred", C.c_double)]


class ChainResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("n_possible", C.c_int32), ("index1", C.c_int32), ("index2", C.c_int32),
                ("n1", C.c_int32), ("n2", C.c_int32), ("chain1", C.c_int32 * OSP_MAX_CHAIN), ("chain2", C.c_int32 * OSP_MAX_CHAIN),
                ("score", C.c_double), ("ln_ed_penalty", C.c_double), ("mn_llh_penalty", C.c_double),
                ("allele_expected_penalty", C.c_double), ("unexpected_chain_penalty", C.c_double), ("inferred_chain_penalty", C.c_double),
                ("edit_distance", C.c_uint64), ("unmet_observations", C.c_uint64)]


def default_cyp_config():
    """cyp_translate / inferred_connections / unexpected_singletons of the bundled database (data fixture);
    identical to Cyp2d6Config::default() (src/cyp2d6/definitions.rs:242-301)."""
    import gzip, json
    d = json.load(gzip.open(os.path.join(ROOT, "tests", "golden", "cyp2d6_db_v0.14.1.json.gz")))["cyp2d6_config"]
    return {"translate": sorted(d["cyp_translate"].items()), "connections": sorted(tuple(x) for x in d["inferred_connections"]),
            "singletons": sorted(d["unexpected_singletons"])}


def _strs(items):
    arr = (C.c_char_p * max(1, len(items)))()
    for i, s in enumerate(items):
        arr[i] = s.encode() if s is not None else None
    return arr


class ChainInputs:
    """flattened find_best_chain_pair inputs shared by the oracle and the product binding"""

    def __init__(self, hap_labels, obs_chains, chain_scores, infer, normalize_all, penalties, ignore_limits, cfg=None):
        cfg = cfg or default_cyp_config()
        self.cfg = cfg
        self.H = len(hap_labels)
        self.types = np.array([REGION_TYPES[t] if isinstance(t, str) else t for t, _ in hap_labels], np.int32)
        self.subtypes = [s for _, s in hap_labels]
        names = sorted(set(obs_chains) | set(chain_scores))          # BTreeMap order; the reference iterates the two maps separately
        self.chain_names = sorted(obs_chains)
        self.score_names = sorted(chain_scores)
        rco, co, items = [0], [0], []
        for n in self.chain_names:
            for ch in obs_chains[n]:
                items += list(ch)
                co.append(len(items))
            rco.append(len(co) - 1)
        self.read_chain_off = np.array(rco, np.int32)
        self.chain_off = np.array(co, np.int32)
        self.chain_items = np.array(items if items else [0], np.int32)
        rwo, ed, ov = [0], [], []
        for n in self.score_names:
            for row in chain_scores[n]:
                assert len(row) == self.H
                ed.append([r[0] for r in row])
                ov.append([r[1] for r in row])
            rwo.append(len(ed))
        self.read_w_off = np.array(rwo, np.int32)
        self.w_ed = np.array(ed if ed else [[0] * max(1, self.H)], np.uint64)
        self.w_ov = np.array(ov if ov else [[0.0] * max(1, self.H)], np.float64)
        self.infer, self.normalize_all, self.ignore = int(infer), int(normalize_all), int(ignore_limits)
        self.penalties = penalties          # (lasso, ln_ed, unexpected, inferred)
        assert len(self.chain_names) == len(self.score_names) or not self.chain_names or not self.score_names


DEFAULT_PENALTIES = (4.0, 2.0, 10.0, 2.0)


def _cyp_cfg_struct(cfg, keep):
    c = CypConfig()
    k, v = _strs([a for a, _ in cfg["translate"]]), _strs([b for _, b in cfg["translate"]])
    ca, cb = _strs([a for a, _ in cfg["connections"]]), _strs([b for _, b in cfg["connections"]])
    sg = _strs(cfg["singletons"])
    keep += [k, v, ca, cb, sg]
    c.n_translate, c.tr_key, c.tr_val = len(cfg["translate"]), k, v
    c.n_conn, c.conn_a, c.conn_b = len(cfg["connections"]), ca, cb
    c.n_single, c.singles = len(cfg["singletons"]), sg
    return c


def oracle_chain_pair(oracle, inp):
    L = oracle.L
    keep = []
    p = ChainProblem()
    p.n_haps = inp.H
    p.type = inp.types.ctypes.data
    st = _strs(inp.subtypes)
    keep.append(st)
    p.subtype = st
    p.cfg = _cyp_cfg_struct(inp.cfg, keep)
    p.n_reads = max(len(inp.chain_names), len(inp.score_names))
    # the two maps normally share their keys; tests with an empty side get empty ranges
    rco = inp.read_chain_off if len(inp.read_chain_off) == p.n_reads + 1 else np.zeros(p.n_reads + 1, np.int32)
    rwo = inp.read_w_off if len(inp.read_w_off) == p.n_reads + 1 else np.zeros(p.n_reads + 1, np.int32)
    keep += [rco, rwo]
    p.read_chain_off, p.chain_off, p.chain_items = rco.ctypes.data, inp.chain_off.ctypes.data, inp.chain_items.ctypes.data
    p.read_w_off, p.w_ed, p.w_ov = rwo.ctypes.data, inp.w_ed.ctypes.data, inp.w_ov.ctypes.data
    p.infer_connections, p.normalize_all_alleles, p.ignore_chain_label_limits = inp.infer, inp.normalize_all, inp.ignore
    p.lasso, p.ln_ed, p.unexpected, p.inferred = inp.penalties
    res = ChainResult()
    L.osp_cyp_find_best_chain_pair.restype = C.c_int32
    L.osp_cyp_find_best_chain_pair(C.byref(p), C.byref(res))
    return res


def chain_hap_string(oracle, chain, hap_labels, detail, cfg=None):
    cfg = cfg or default_cyp_config()
    keep = []
    c = _cyp_cfg_struct(cfg, keep)
    types = np.array([REGION_TYPES[t] if isinstance(t, str) else t for t, _ in hap_labels], np.int32)
    st = _strs([s for _, s in hap_labels])
    ch = np.array(chain, np.int32)
    out = C.create_string_buffer(1024)
    oracle.L.osp_cyp_convert_chain_to_hap(ch.ctypes.data_as(C.c_void_p), len(ch), types.ctypes.data_as(C.c_void_p), st, int(detail),
                                          C.byref(c), out, 1024)
    return out.value.decode()


REGION_HIT_DTYPE = np.dtype([(n, np.int32) for n in ("template_idx", "start", "end", "seq_len", "nm", "unmapped", "clip_start", "clip_end")])


def oracle_find_base_type(oracle, seq, templates, template_type, max_missing_frac, rescore=True):
    """find_base_type_in_sequence (src/cyp2d6/haplotyper.rs:142-315) on the alignment contract; rescore (the library's default, context option "mm2_rescore"): the placements
    that pass max_ed_frac are re-scored the reference's way and the rules run on minimap2's numbers (osp_cyp_find_base_type_ex)"""
    L = oracle.L
    enc = [oracle.encode(t) for t in templates]
    ptrs = (C.c_void_p * len(enc))(*[e.ctypes.data for e in enc])
    lens = np.array([len(e) for e in enc], np.int32)
    tt = np.ascontiguousarray(template_type, np.int32)
    s = oracle.encode(seq)
    out = np.zeros(64, REGION_HIT_DTYPE)
    L.osp_cyp_find_base_type_ex.restype = C.c_int32
    L.osp_cyp_find_base_type_ex.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int32, C.c_void_p, C.c_int32]
    n = L.osp_cyp_find_base_type_ex(s.ctypes.data_as(C.c_void_p), len(s), len(enc), ptrs, lens.ctypes.data_as(C.c_void_p),
                                    tt.ctypes.data_as(C.c_void_p), float(max_missing_frac), 1 if rescore else 0, out.ctypes.data_as(C.c_void_p), len(out))
    return out[:n]


def oracle_weight_sequence(oracle, seq, consensus, allowed):
    """weight_sequence (src/cyp2d6/chaining.rs:28-103) on the alignment contract; returns (ed, ov, kept)"""
    L = oracle.L
    enc = [oracle.encode(t) for t in consensus]
    ptrs = (C.c_void_p * len(enc))(*[e.ctypes.data for e in enc])
    lens = np.array([len(e) for e in enc], np.int32)
    al = np.ascontiguousarray(allowed, np.uint8)
    s = oracle.encode(seq)
    ed = np.zeros(len(enc), np.uint64)
    ov = np.zeros(len(enc), np.float64)
    L.osp_cyp_weight_sequence.restype = C.c_int32
    L.osp_cyp_weight_sequence.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    kept = L.osp_cyp_weight_sequence(s.ctypes.data_as(C.c_void_p), len(s), len(enc), ptrs, lens.ctypes.data_as(C.c_void_p),
                                     al.ctypes.data_as(C.c_void_p), ed.ctypes.data_as(C.c_void_p), ov.ctypes.data_as(C.c_void_p))
    return ed, ov, kept


def oracle_build_chains(oracle, hap_type, read_seg_off, ed, kept):
    """osp_cyp_build_chains -> same dict as pb_starphase_amd.ffi.build_chains; returns None on 'chain collapse'"""
    hap_type = np.ascontiguousarray(hap_type, np.int32)
    read_seg_off = np.ascontiguousarray(read_seg_off, np.uint32)
    ed = np.ascontiguousarray(ed, np.uint64)
    kept = np.ascontiguousarray(kept, np.uint8)
    n_haps, n_reads, n_seg = len(hap_type), len(read_seg_off) - 1, len(kept)
    chain_cap, item_cap = 1 << 16, 1 << 20
    read_index = np.zeros(max(1, n_reads), np.uint32)
    rco, rwo = np.zeros(n_reads + 1, np.uint32), np.zeros(n_reads + 1, np.uint32)
    co, items = np.zeros(chain_cap + 1, np.uint32), np.zeros(item_cap, np.uint32)
    w_seg = np.zeros(max(1, n_seg), np.uint32)
    uniq, false_allele = np.zeros(n_haps, np.uint64), np.zeros(n_haps, np.uint8)
    nk = C.c_uint32(0)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    oracle.L.osp_cyp_build_chains.restype = C.c_int
    rc = oracle.L.osp_cyp_build_chains(n_haps, p(hap_type), n_reads, p(read_seg_off), p(ed), p(kept), p(read_index), C.byref(nk), p(rco), p(co),
                                       C.c_uint32(chain_cap), p(items), C.c_uint32(item_cap), p(rwo), p(w_seg), p(uniq), p(false_allele))
    if rc == 2:
        return None
    assert rc == 0
    nk = nk.value
    chains = [[[int(x) for x in items[co[c]:co[c + 1]]] for c in range(rco[k], rco[k + 1])] for k in range(nk)]
    rows = [[int(x) for x in w_seg[rwo[k]:rwo[k + 1]]] for k in range(nk)]
    return dict(read_index=[int(x) for x in read_index[:nk]], chains=chains, w_rows=rows, unique_counts=uniq, false_allele=false_allele)


class ConsConfig(C.Structure):
    _fields_ = [("min_count", C.c_int32), ("dual_max_ed_delta", C.c_int32), ("allow_early_termination", C.c_int32), ("allow_dual", C.c_int32),
                ("offset_window", C.c_int32), ("offset_compare_length", C.c_int32), ("min_af", C.c_double),
                ("max_queue_size", C.c_int32), ("max_capacity_per_size", C.c_int32), ("max_nodes_wo_constraint", C.c_int32), ("pad", C.c_int32)]


class ConsResult(C.Structure):
    _fields_ = [("is_dual", C.c_int32), ("len1", C.c_int32), ("len2", C.c_int32), ("split_at", C.c_int32), ("gave_up", C.c_int64), ("best_total", C.c_int64),
                ("nodes_expanded", C.c_int64)]


def cons_config(min_count=3, min_af=0.10, dual_max_ed_delta=100, early_termination=True, dual=True, offset_window=400, offset_compare_length=50,
                max_queue_size=0, max_capacity_per_size=0, max_nodes_wo_constraint=0):
    """the three search bounds default (0) to the values of dwfa_config_from_cli / waffle_con: 20, 10, 1000"""
    return ConsConfig(min_count, dual_max_ed_delta, int(early_termination), int(dual), offset_window, offset_compare_length, min_af,
                      max_queue_size, max_capacity_per_size, max_nodes_wo_constraint, 0)


def with_dual(cfg, dual):
    return ConsConfig(cfg.min_count, cfg.dual_max_ed_delta, cfg.allow_early_termination, int(dual), cfg.offset_window, cfg.offset_compare_length, cfg.min_af,
                      cfg.max_queue_size, cfg.max_capacity_per_size, cfg.max_nodes_wo_constraint, 0)


def oracle_consensus(oracle, reads, offsets=None, cfg=None, cap=None):
    """osp_consensus -> dict(cons=[str, str|None], is_dual, is_cons1, score1, score2 (None = -1), split_at, nodes_expanded, gave_up)"""
    cfg = cfg or cons_config()
    enc = [oracle.encode(r) for r in reads]
    n = len(reads)
    lens = np.array([len(r) for r in reads], np.int32)
    offs = np.array([-1 if o is None else int(o) for o in (offsets or [None] * n)], np.int32)
    cap = cap or (int(lens.max() if n else 0) + int(max(0, offs.max() if n else 0)) + 64)
    c1, c2 = np.zeros(cap + 1, np.uint8), np.zeros(cap + 1, np.uint8)
    is1, s1, s2 = np.zeros(max(1, n), np.uint8), np.zeros(max(1, n), np.int32), np.zeros(max(1, n), np.int32)
    ptrs = (C.c_void_p * max(1, n))(*[e.ctypes.data for e in enc])
    res = ConsResult()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    oracle.L.osp_consensus.restype = C.c_int
    rc = oracle.L.osp_consensus(n, ptrs, p(lens), p(offs), C.byref(cfg), p(c1), p(c2), cap, p(is1), p(s1), p(s2), C.byref(res))
    assert rc == 0
    dec = lambda a, k: "".join("ACGT"[x] for x in a[:k])
    return dict(cons=[dec(c1, res.len1), dec(c2, res.len2) if res.is_dual else None], is_dual=bool(res.is_dual), is_cons1=is1[:n].astype(bool),
                score1=s1[:n].copy(), score2=s2[:n].copy(), split_at=res.split_at, nodes_expanded=res.nodes_expanded, gave_up=bool(res.gave_up))


def dual_consensus_two_pass(run, reads, offsets=None, cfg=None):
    """(round 1 ran a two-pass split policy here; the best-first search decides where to split by cost, so this is one dual run)"""
    return run(reads, offsets, with_dual(cfg or cons_config(), True))


PRIORITY_RETRY_MIN_AF = (0.15, 0.20, 0.30, 0.40)          # sp_consensus_priority's ladder for searches that give up


def oracle_priority_consensus(oracle, levels, cfg, offsets=None, seeds=None, retry_ladder=False):
    """The multi-way contract of sp_consensus_priority on top of the oracle's two-way consensus (include/starphase_hip.h):
    levels = list (per level) of lists of strings.  Returns (group_of, [[consensus per level] per group])."""
    n, nl = len(levels[0]), len(levels)
    half = cfg.offset_window // 2
    run = lambda rd, offs, c: oracle_consensus(oracle, rd, offs, c)
    single, dual = with_dual(cfg, False), with_dual(cfg, True)

    def rebased(members, level):
        if offsets is None or offsets[level] is None:
            return None
        vals = [0 if offsets[level][r] is None else int(offsets[level][r]) for r in members]
        mn = min(vals)
        return [None if v == mn else v - mn + (0 if mn == 0 else half) for v in vals]

    def solve(members, level, kept=None):
        """-> [(members, {level: consensus})]: `kept` holds the consensus of every level whose two-way search on exactly these members ended with ONE consensus at
        the configured fraction -- that search's consensus is the group's consensus at that level; only levels without one are solved again at the end"""
        kept = dict(kept or {})
        res = dual_consensus_two_pass(run, [levels[level][r] for r in members], rebased(members, level), dual)
        first_try = True
        # a search that gave up (no complete node: a mixture of more classes than a search holds consensuses can exhaust the queue and capacity
        # bounds) is run again with only the stronger differences as candidates; the split it finds is the split, the groups it leaves are
        # solved with the configured fraction again
        for af in PRIORITY_RETRY_MIN_AF:
            if not retry_ladder or not res["gave_up"] or af <= cfg.min_af:
                continue
            stricter = ConsConfig(dual.min_count, dual.dual_max_ed_delta, dual.allow_early_termination, 1, dual.offset_window, dual.offset_compare_length, af,
                                  dual.max_queue_size, dual.max_capacity_per_size, dual.max_nodes_wo_constraint, 0)
            res = run([levels[level][r] for r in members], rebased(members, level), stricter)
            first_try = False
        g1 = [r for r, f in zip(members, res["is_cons1"]) if f]
        g2 = [r for r, f in zip(members, res["is_cons1"]) if not f]
        if res["is_dual"] and g1 and g2:
            return solve(g1, level) + solve(g2, level)
        if first_try and not res["is_dual"] and not res["gave_up"]:
            kept[level] = res["cons"][0]
        if level + 1 < nl:
            return solve(members, level + 1, kept)
        return [(members, kept)]

    keys = sorted(set(-1 if (seeds is None or seeds[r] is None) else int(seeds[r]) for r in range(n)))
    groups = []
    for k in keys:
        members = [r for r in range(n) if (-1 if (seeds is None or seeds[r] is None) else int(seeds[r])) == k]
        groups += solve(members, 0)
    group_of = np.zeros(n, np.int32)
    cons = []
    for g, (members, kept) in enumerate(groups):
        for r in members:
            group_of[r] = g
        cons.append([kept[l] if l in kept else oracle_consensus(oracle, [levels[l][r] for r in members], rebased(members, l), single)["cons"][0] for l in range(nl)])
    return group_of, cons


def oracle_variant_states(oracle, seq, backbone, var_pos, var_ref, var_alt):
    """osp_cyp_variant_states -> (states uint8[n_variants], (a_start, a_end, b_start, b_end, nm) or None)"""
    nv = len(var_pos)
    se, be = oracle.encode(seq), oracle.encode(backbone)
    pos = np.ascontiguousarray(var_pos, np.int32)
    refs = (C.c_char_p * max(1, nv))(*[r.encode() for r in var_ref])
    alts = (C.c_char_p * max(1, nv))(*[a.encode() for a in var_alt])
    states = np.full(nv, 3, np.uint8)
    aln = np.zeros(5, np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    oracle.L.osp_cyp_variant_states.restype = C.c_int
    ok = oracle.L.osp_cyp_variant_states(p(se), len(se), p(be), len(be), nv, p(pos), refs, alts, p(states), p(aln))
    return states, (tuple(int(x) for x in aln) if ok else None)
