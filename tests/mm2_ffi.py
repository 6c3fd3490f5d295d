"""ctypes access to the minimap2 restatement of the oracle (oracle/mm2.c): TEST INFRASTRUCTURE ONLY (tests/, the audit scripts under
profiles/scripts and bench.py's cpu_baseline leg)."""
import ctypes as C

import numpy as np

import oracle_ffi


class Opts(C.Structure):
    _fields_ = [("k", C.c_int32), ("w", C.c_int32),
                ("a", C.c_int32), ("b", C.c_int32), ("q", C.c_int32), ("e", C.c_int32), ("q2", C.c_int32), ("e2", C.c_int32),
                ("sc_ambi", C.c_int32),
                ("zdrop", C.c_int32), ("zdrop_inv", C.c_int32), ("end_bonus", C.c_int32),
                ("bw", C.c_int32), ("max_gap", C.c_int32),
                ("min_cnt", C.c_int32), ("min_chain_score", C.c_int32),
                ("max_chain_skip", C.c_int32), ("max_chain_iter", C.c_int32),
                ("chain_gap_scale", C.c_float), ("mask_level", C.c_float), ("pri_ratio", C.c_float),
                ("best_n", C.c_int32), ("min_dp_max", C.c_int32), ("min_ksw_len", C.c_int32),
                ("min_mid_occ", C.c_int32), ("max_mid_occ", C.c_int32), ("max_max_occ", C.c_int32), ("occ_dist", C.c_int32),
                ("mid_occ_frac", C.c_float), ("forward_only", C.c_int32)]


HIT_FIELDS = ("rid", "rev", "q_start", "q_end", "q_len", "t_start", "t_end", "t_len", "nm", "mlen", "blen", "n_ambi",
              "dp_score", "dp_max", "chain_score", "n_seeds", "primary", "n_cigar", "cigar_off")
HIT_DTYPE = np.dtype([(n, np.int32) for n in HIT_FIELDS])

OPS = {0: "M", 1: "I", 2: "D", 7: "=", 8: "X"}

SEED_SEL = 16                                           # OMM_SEED_SEL
SEED_HIT_FIELDS = ("rid", "rev", "chain_score", "n_seeds", "t_len", "sel_rank", "diag", "ok", "cell_nm", "a_start", "a_end", "b_start", "b_end",
                   "dp_max", "nm", "t_start", "t_end", "q_start", "q_end", "primary")
SEED_HIT_DTYPE = np.dtype([(n, np.int32) for n in SEED_HIT_FIELDS])
REG_FIELDS = ("rid", "rev", "score", "cnt", "qs", "qe", "rs", "re", "parent", "kept")


class Mm2:
    def __init__(self, oracle=None):
        self.O = oracle or oracle_ffi.load()
        L = self.L = self.O.L
        vp, i32 = C.c_void_p, C.c_int32
        L.omm_default_opts.argtypes = [C.POINTER(Opts)]
        L.omm_index_build.restype = vp
        L.omm_index_build.argtypes = [vp, vp, i32, C.POINTER(Opts)]
        L.omm_index_free.argtypes = [vp]
        L.omm_index_mid_occ.restype = i32
        L.omm_index_mid_occ.argtypes = [vp]
        L.omm_index_n_minimizers.restype = C.c_int64
        L.omm_index_n_minimizers.argtypes = [vp]
        L.omm_map.restype = i32
        L.omm_map.argtypes = [vp, vp, i32, C.POINTER(Opts), vp, i32, vp, i32]
        L.omm_map_pair.restype = i32
        L.omm_map_pair.argtypes = [vp, i32, vp, i32, C.POINTER(Opts), vp, i32, vp, i32]
        L.omm_dp.argtypes = [vp, i32, vp, i32, C.POINTER(Opts), i32, i32, i32, vp, vp, i32, C.POINTER(i32)]
        L.omm_global_score_bruteforce.restype = i32
        L.omm_global_score_bruteforce.argtypes = [vp, i32, vp, i32, C.POINTER(Opts)]
        L.omm_sketch.restype = i32
        L.omm_sketch.argtypes = [vp, i32, C.POINTER(Opts), vp, vp, vp, i32]
        L.omm_chain_stage.restype = i32
        L.omm_chain_stage.argtypes = [vp, vp, i32, C.POINTER(Opts), vp, i32, vp]
        L.omm_anchors.restype = C.c_int64
        L.omm_anchors.argtypes = [vp, vp, i32, C.POINTER(Opts), vp, vp, C.c_int64]
        L.omm_hla_k1_seeded.restype = i32
        L.omm_hla_k1_seeded.argtypes = [vp, vp, i32, C.POINTER(Opts), vp, C.POINTER(i32), C.POINTER(i32)]

    def opts(self, **kw):
        o = Opts()
        self.L.omm_default_opts(C.byref(o))
        for k, v in kw.items():
            setattr(o, k, v)
        return o

    def codes(self, s):
        return s if isinstance(s, np.ndarray) else self.O.encode(s)

    def _hits(self, n, hits, pool):
        out = []
        for h in hits[:n]:
            d = {k: int(h[k]) for k in HIT_FIELDS}
            d["cigar"] = [(int(x >> 4), OPS[int(x & 15)]) for x in pool[d["cigar_off"]:d["cigar_off"] + d["n_cigar"]]]
            out.append(d)
        return out

    def map_pair(self, target, query, opts=None, max_hits=8):
        """aligner.with_seq(target) ; aligner.map(query)"""
        o = opts or self.opts()
        t, q = self.codes(target), self.codes(query)
        hits = np.zeros(max_hits, HIT_DTYPE)
        pool = np.zeros(4 * (len(t) + len(q)) + 64, np.uint32)
        n = self.L.omm_map_pair(t.ctypes.data, len(t), q.ctypes.data, len(q), C.byref(o), hits.ctypes.data, max_hits,
                                pool.ctypes.data, len(pool))
        return self._hits(n, hits, pool)

    def sketch(self, seq, opts=None):
        """the (w,k)-minimizers of a sequence in position order -> (hash u64[], end position i32[], strand u8[])"""
        o = opts or self.opts()
        q = self.codes(seq)
        cap = len(q) + 1
        h, p, st = np.zeros(cap, np.uint64), np.zeros(cap, np.int32), np.zeros(cap, np.uint8)
        n = self.L.omm_sketch(q.ctypes.data, len(q), C.byref(o), h.ctypes.data, p.ctypes.data, st.ctypes.data, cap)
        return h[:n].copy(), p[:n].copy(), st[:n].copy()

    def dp(self, target, query, opts=None, band=751, mode=0, right=0):
        o = opts or self.opts()
        t, q = self.codes(target), self.codes(query)
        out = np.zeros(8, np.int32)
        cg = np.zeros(2 * (len(t) + len(q)) + 8, np.uint32)
        n = C.c_int32(0)
        self.L.omm_dp(t.ctypes.data, len(t), q.ctypes.data, len(q), C.byref(o), band, mode, right, out.ctypes.data, cg.ctypes.data, len(cg),
                      C.byref(n))
        keys = ("score", "max", "max_t", "max_q", "zdropped", "reach_end", "t_end", "q_end")
        r = {k: int(v) for k, v in zip(keys, out)}
        r["cigar"] = [(int(x >> 4), OPS[int(x & 15)]) for x in cg[:n.value]]
        return r

    def brute(self, target, query, opts=None):
        o = opts or self.opts()
        t, q = self.codes(target), self.codes(query)
        return self.L.omm_global_score_bruteforce(t.ctypes.data, len(t), q.ctypes.data, len(q), C.byref(o))


class Index:
    """aligner.with_index(fasta of many sequences)"""

    def __init__(self, mm, seqs, opts=None):
        self.mm = mm
        self.o = opts or mm.opts()
        enc = [mm.codes(s) for s in seqs]
        self.off = np.zeros(len(enc) + 1, np.int64)
        self.off[1:] = np.cumsum([len(e) for e in enc])
        self.codes = np.concatenate(enc) if enc else np.zeros(0, np.uint8)
        self.h = mm.L.omm_index_build(self.codes.ctypes.data, self.off.ctypes.data, len(enc), C.byref(self.o))

    def map(self, query, opts=None, max_hits=16, want_cigar=False):
        o = opts or self.o
        q = self.mm.codes(query)
        hits = np.zeros(max_hits, HIT_DTYPE)
        pool = np.zeros(8 * len(q) * (max_hits if want_cigar else 0) + 16, np.uint32)
        n = self.mm.L.omm_map(self.h, q.ctypes.data, len(q), C.byref(o), hits.ctypes.data, max_hits,
                              pool.ctypes.data if want_cigar else None, len(pool) if want_cigar else 0)
        return self.mm._hits(n, hits, pool)

    def map_raw(self, q, opts=None, max_hits=16):
        o = opts or self.o
        hits = np.zeros(max_hits, HIT_DTYPE)
        n = self.mm.L.omm_map(self.h, q.ctypes.data, len(q), C.byref(o), hits.ctypes.data, max_hits, None, 0)
        return hits[:n]

    def chain_stage(self, query, opts=None, cap=65536):
        """seeding + chaining + selection without the base-level alignment -> (regs int32[n][10] in chain_anchors' order, stats int64[8])"""
        o = opts or self.o
        q = self.mm.codes(query)
        regs, st = np.zeros((cap, 10), np.int32), np.zeros(8, np.int64)
        n = self.mm.L.omm_chain_stage(self.h, q.ctypes.data, len(q), C.byref(o), regs.ctypes.data, cap, st.ctypes.data)
        return regs[:min(n, cap)].copy(), st

    def anchors(self, query, opts=None, cap=1 << 21):
        o = opts or self.o
        q = self.mm.codes(query)
        x, y = np.zeros(cap, np.uint64), np.zeros(cap, np.uint64)
        n = self.mm.L.omm_anchors(self.h, q.ctypes.data, len(q), C.byref(o), x.ctypes.data, y.ctypes.data, cap)
        return x[:min(n, cap)].copy(), y[:min(n, cap)].copy()

    def k1_seeded(self, query, opts=None):
        """realign_record's seeded map as the library runs it (omm_hla_k1_seeded) -> (index of the accepted hit or -1, hits in output order, number of chains)"""
        o = opts or self.o
        q = self.mm.codes(query)
        hits = np.zeros(SEED_SEL, SEED_HIT_DTYPE)
        nh, nc = C.c_int32(0), C.c_int32(0)
        pick = self.mm.L.omm_hla_k1_seeded(self.h, q.ctypes.data, len(q), C.byref(o), hits.ctypes.data, C.byref(nh), C.byref(nc))
        return pick, hits[:nh.value].copy(), nc.value

    @property
    def mid_occ(self):
        return self.mm.L.omm_index_mid_occ(self.h)

    def close(self):
        if self.h:
            self.mm.L.omm_index_free(self.h)
            self.h = None

    def __del__(self):
        self.close()
