"""The reference's CPU path for one CYP2D6 sample in its own CALL PATTERN (test infrastructure: tests/, tests/golden/make_concordance.py and bench.py's
cpu_baseline leg).  The counterpart of tests/cpu_port_seeded.py for `diplotype_cyp2d6` (src/cyp2d6/caller.rs:39-741).

What the reference does per sample and what stands in for it here:
  * per read: the read is indexed, the 39 templates are mapped onto it, every mapping minimap2 returns is a candidate; filter / sort / collapse
    (find_base_type_in_sequence, src/cyp2d6/haplotyper.rs:142-315): oracle/cyp_mm2.c on oracle/mm2.c (minimap2's published algorithm restated; the
    binary is not on disk)
  * multi-way consensus of the region sequences (PriorityConsensusDWFA, caller.rs:145-306; waffle_con): oracle/consensus.c
  * typing of every consensus: find_full_type_in_sequence with the placement on the backbone taken from the restatement's mapping (the longest block,
    haplotyper.rs:391-412), the variant graph on that stretch (oracle/cyp.c), the allele scores (:470-524)
  * per region of every read: weight_sequence (src/cyp2d6/chaining.rs:28-103): the segment indexed, every allowed consensus mapped onto it
  * chains, the best chain pair, the strings: the oracle's routines (pinned on the reference's own vectors, tests/test_oracle_cyp.py)
The driver is tests/cyp_pipeline.py's -- the same one the library's own contract is checked with -- with these three call sites swapped in.
A scalar port: a reported baseline and a concordance anchor, not minimap2's SSE build."""
import ctypes as C
import multiprocessing as mp
import os
import sys
import time

import numpy as np

import cyp_pipeline as cp
import oracle_ffi as of

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = {}


def tables(cfg, gene_def, locus):
    """the typing database of a sample from the independent Python statement of the tables (oracle/cyp_db.py) -> (cyp_pipeline.Db, the chain configuration)"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cyp_db
    hyb = cyp_db.generate_cyp_hybrids(locus.slice, cfg)
    order = cyp_db.template_order(hyb)
    lv = cyp_db.load_variant_database(gene_def)
    names, rows = cyp_db.haplotype_lookup(gene_def, lv)
    bb = cfg["cyp_coordinates"]["CYP2D6_wfa_backbone"]
    T = of.REGION_TYPES
    db = cp.Db([cyp_db.full_allele(*k) for k in order], [T[k[0]] for k in order], [k[1] for k in order], [hyb[k] for k in order],
               [k in cyp_db.MAPPED_HYBRIDS for k in order], locus.slice(bb["start"], bb["end"]),
               [(p - bb["start"], r, a) for p, r, a in lv["variants"]], lv["vi"], names, rows, var_labels=lv["labels"])
    chain_cfg = dict(translate=sorted(cfg.get("cyp_translate", {}).items()), connections=sorted(tuple(x) for x in cfg.get("inferred_connections", [])),
                     singletons=sorted(cfg.get("unexpected_singletons", [])))
    return db, chain_cfg


class Mm2Aligner:
    """the three alignment call sites of the CYP2D6 caller on the minimap2 restatement"""

    def __init__(self, oracle, opts=None):
        import mm2_ffi
        self.mm = mm2_ffi.Mm2(oracle)
        self.o = opts or self.mm.opts()
        L = oracle.L
        vp, i32 = C.c_void_p, C.c_int32
        L.omm_cyp_find_base_type.restype = i32
        L.omm_cyp_find_base_type.argtypes = [vp, i32, i32, vp, vp, vp, C.c_double, vp, vp, i32]
        L.omm_cyp_weight_sequence.restype = i32
        L.omm_cyp_weight_sequence.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp]
        L.omm_cyp_place.restype = i32
        L.omm_cyp_place.argtypes = [vp, i32, vp, i32, vp, vp]
        L.osp_cyp_variant_states_at.restype = i32
        self._tm = {}

    def _templates(self, oracle, db):
        key = id(db)
        if key not in self._tm:
            enc = [oracle.encode(t) for t in db.seqs]
            self._tm[key] = (enc, (C.c_void_p * len(enc))(*[e.ctypes.data for e in enc]), np.array([len(e) for e in enc], np.int32),
                             np.ascontiguousarray(db.types, np.int32))
        return self._tm[key]

    def find_base_type(self, oracle, seq, db, max_missing):
        enc, ptrs, lens, tt = self._templates(oracle, db)
        s = oracle.encode(seq)
        out = np.zeros(64, of.REGION_HIT_DTYPE)
        n = oracle.L.omm_cyp_find_base_type(s.ctypes.data, len(s), len(enc), ptrs, lens.ctypes.data, tt.ctypes.data, float(max_missing), C.byref(self.o),
                                            out.ctypes.data, len(out))
        return out[:n]

    def weight_sequence(self, oracle, seq, consensus, allowed):
        enc = [oracle.encode(t) for t in consensus]
        ptrs = (C.c_void_p * max(1, len(enc)))(*[e.ctypes.data for e in enc])
        lens = np.array([len(e) for e in enc], np.int32)
        al = np.ascontiguousarray(allowed, np.uint8)
        s = oracle.encode(seq)
        ed, ov = np.zeros(len(enc), np.uint64), np.zeros(len(enc), np.float64)
        kept = oracle.L.omm_cyp_weight_sequence(s.ctypes.data, len(s), len(enc), ptrs, lens.ctypes.data, al.ctypes.data, C.byref(self.o), ed.ctypes.data, ov.ctypes.data)
        return ed, ov, kept

    def variant_states(self, oracle, seq, db):
        nv = len(db.variants)
        se, be = oracle.encode(seq), oracle.encode(db.backbone)
        states = np.full(nv, 3, np.uint8)
        place = np.zeros(6, np.int32)
        if not oracle.L.omm_cyp_place(se.ctypes.data, len(se), be.ctypes.data, len(be), C.byref(self.o), place.ctypes.data):
            return states
        pos = np.ascontiguousarray([v[0] for v in db.variants], np.int32)
        refs = (C.c_char_p * max(1, nv))(*[v[1].encode() for v in db.variants])
        alts = (C.c_char_p * max(1, nv))(*[v[2].encode() for v in db.variants])
        oracle.L.osp_cyp_variant_states_at(se.ctypes.data_as(C.c_void_p), len(se), be.ctypes.data_as(C.c_void_p), len(be), nv, pos.ctypes.data_as(C.c_void_p), refs, alts,
                                           int(place[0]), int(place[1]), int(place[2]), int(place[3]), int(place[5]), states.ctypes.data_as(C.c_void_p))
        return states


def _regions_worker(args):
    lo, hi = args
    t0 = time.perf_counter()
    out = [G["al"].find_base_type(G["o"], G["reads"][r], G["db"], 0.5) for r in range(lo, hi)]
    return lo, out, time.perf_counter() - t0


def _weights_worker(args):
    lo, hi = args
    t0 = time.perf_counter()
    out = [G["al"].weight_sequence(G["o"], G["segs"][s], G["final"], G["allowed"]) for s in range(lo, hi)]
    return lo, out, time.perf_counter() - t0


def _spread(fn, n, cores, chunk):
    jobs = [(i, min(n, i + chunk)) for i in range(0, n, chunk)]
    if cores <= 1 or len(jobs) <= 1:
        parts = [fn(j) for j in jobs]
    else:
        with mp.get_context("fork").Pool(min(cores, len(jobs))) as pool:
            parts = pool.map(fn, jobs)
    out, cpu = [], 0.0
    for _lo, rows, dt in sorted(parts, key=lambda p: p[0]):
        out += rows; cpu += dt
    return out, cpu


def run(oracle, db, chain_cfg, reads, cores=None, stages=None):
    """-> (the call as tests/cyp_pipeline.diplotype reports it, timings).  The per-read region search and the per-segment weights are spread over `cores` forked
    workers (the reference is single-threaded: timings carry the CPU seconds of each stage, so that one-thread and all-core figures can both be stated)"""
    cores = cores or max(1, len(os.sched_getaffinity(0)))
    al = Mm2Aligner(oracle)
    G.update(o=oracle, al=al, db=db, reads=reads)
    tm = {}
    t0 = time.perf_counter()
    regions, tm["regions_cpu_s"] = _spread(_regions_worker, len(reads), cores, max(1, min(16, len(reads) // (4 * cores) + 1)))
    tm["regions_wall_s"] = time.perf_counter() - t0

    def weigh(segs, final, allowed):
        G.update(segs=segs, final=final, allowed=allowed)
        t1 = time.perf_counter()
        rows, tm["weights_cpu_s"] = _spread(_weights_worker, len(segs), cores, max(1, min(32, len(segs) // (4 * cores) + 1)))
        tm["weights_wall_s"] = time.perf_counter() - t1
        return rows

    st = stages if stages is not None else {}
    t1 = time.perf_counter()
    res = cp.diplotype(oracle, db, reads, cfg=chain_cfg, aligner=al, regions=regions, weigh=weigh, stages=st)
    tm["rest_wall_s"] = time.perf_counter() - t1 - tm.get("weights_wall_s", 0.0)        # consensus, typing, chains, chain pair: sequential, as in the reference
    tm["wall_s"] = time.perf_counter() - t0
    tm["one_thread_s"] = tm["regions_cpu_s"] + tm.get("weights_cpu_s", 0.0) + tm["rest_wall_s"]
    tm["cores"] = cores
    return res, tm
