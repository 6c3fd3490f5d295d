"""Generates tests/golden/fullsize_oracle.json.gz: what the CPU ORACLE of the library's own contract (oracle/align.c, consensus.c, cyp.c -- the statement the HIP kernels are
bit-exact against) computes at BASELINE's full sizes, where it is too slow for the GPU box's test run:

  k1     configs[1]: the oracle's whole-read search (osp_hla_k1_read: anchors, every allele cell, acceptance) for 1,000 of the 10,000 reads
  cyp    configs[2]: the oracle-assembled pipeline (tests/cyp_pipeline.py) at 2,000 reads for `*1/*2` and the branching `*4+*68/*1`: consensus strings, labels, chains,
         the f64 score, the haplotype strings

Data only; inputs are regenerated from seeds on both sides.  Usage (CPU, ~25 min on 8 cores): python tests/golden/make_fullsize.py [k1] [cyp]"""
import ctypes as C
import gzip
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402

OUT = os.path.join(HERE, "fullsize_oracle.json.gz")
G = {}


def _k1_worker(chunk):
    o, tb, fx, reads = G["o"], G["tb"], G["fx"], G["reads"]
    L = o.L
    L.osp_hla_k1_read.restype = C.c_int32
    refs = tb.refs
    n_all = len(fx.ids)
    enc = [e if e is not None else np.zeros(0, np.uint8) for e in tb.fwd_e]
    off = np.array([(-2 ** 31 if x is None else x) for x in tb.off], np.int32)
    ref_ptr = (C.c_void_p * len(refs))(*[r.ctypes.data for r in refs]); ref_len = np.array([len(r) for r in refs], np.int32)
    al_ptr = (C.c_void_p * n_all)(*[(e.ctypes.data if len(e) else None) for e in enc]); al_len = np.array([len(e) for e in enc], np.int32)
    gene_of = fx.gene_of.astype(np.int32)
    out = []
    for r in chunk:
        re = o.encode(reads[r])
        ncell = C.c_int64(0)
        b = L.osp_hla_k1_read(re.ctypes.data_as(C.c_void_p), len(re), len(refs), ref_ptr, ref_len.ctypes.data_as(C.c_void_p), n_all, al_ptr,
                              al_len.ctypes.data_as(C.c_void_p), gene_of.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), None, C.byref(ncell))
        out.append((int(r), int(b)))
    return out


def k1_section(o, synth):
    import hla_expected as hx
    fx = synth.HlaFixture()
    wl = synth.Config2Workload(fx, n_reads=10000, seed=1000)
    pick = np.sort(np.random.default_rng(1).choice(len(wl.reads), 1000, replace=False)).tolist()      # the reads tests/test_gpu_panel.py compares pruned with exhaustive
    tb = hx.K1Tables(o, fx)
    for a in range(len(fx.ids)):
        tb.am(a) if False else None
    G.update(o=o, tb=tb, fx=fx, reads=wl.reads)
    t0 = time.time()
    cores = len(os.sched_getaffinity(0))
    with mp.get_context("fork").Pool(cores) as pool:
        parts = pool.map(_k1_worker, [pick[i::cores * 4] for i in range(cores * 4)])
    best = dict(x for p in parts for x in p)
    return {"workload": "synth.Config2Workload(HlaFixture(), n_reads=10000, seed=1000); reads = sorted(default_rng(1).choice(10000, 1000, replace=False))",
            "reads": pick, "best_allele": [best[r] for r in pick], "seconds": time.time() - t0}


def cyp_section(o, synth):
    import cpu_port_cyp as cpc
    import cyp_cases_real as cr
    import cyp_pipeline as cp
    cfg, gene_def = cr.load_db()
    locus = synth.Chr22Locus(cfg, gene_def, seed=3)
    db, ccfg = cpc.tables(cfg, gene_def, locus)
    out = {"workload": "Chr22Locus(seed=3).sample(default_rng(7), haps, 2000); tests/cyp_pipeline.diplotype on the contract's aligner", "scenarios": {}}
    sc = {n: (h, e) for n, h, e in cr.scenarios(locus)}
    for name in ("*1/*2", "*4+*68/*1"):
        reads = locus.sample(np.random.default_rng(7), sc[name][0], 2000)
        t0 = time.time()
        G.update(o=o, db=db, reads=reads)
        cores = len(os.sched_getaffinity(0))
        with mp.get_context("fork").Pool(cores) as pool:
            parts = pool.map(_regions_worker, [(i, min(len(reads), i + 16)) for i in range(0, len(reads), 16)])
        regions = [h for _lo, rows in sorted(parts, key=lambda p: p[0]) for h in rows]

        def weigh(segs, final, allowed):
            G.update(segs=segs, final=final, allowed=allowed)
            with mp.get_context("fork").Pool(cores) as pool:
                parts = pool.map(_weights_worker, [(i, min(len(segs), i + 32)) for i in range(0, len(segs), 32)])
            return [w for _lo, rows in sorted(parts, key=lambda p: p[0]) for w in rows]
        res = cp.diplotype(o, db, reads, cfg=ccfg, regions=regions, weigh=weigh)
        out["scenarios"][name] = {"status": int(res["status"]), "consensus": res["consensus"], "labels": [[int(t), s] for t, s in res["labels"]],
                                  "chain1": [int(x) for x in res.get("chain1", [])], "chain2": [int(x) for x in res.get("chain2", [])], "score": res.get("score"),
                                  "hap": [res.get("hap1", ""), res.get("hap2", "")], "core": [res.get("core1", ""), res.get("core2", "")],
                                  "deep": [res.get("deep1", ""), res.get("deep2", "")], "seconds": time.time() - t0}
        print(name, res["status"], res.get("hap1"), "/", res.get("hap2"), round(time.time() - t0, 1), "s", flush=True)
    return out


def _regions_worker(args):
    import cyp_pipeline as cp
    lo, hi = args
    return lo, [cp.CONTRACT.find_base_type(G["o"], G["reads"][r], G["db"], 0.5) for r in range(lo, hi)]


def _weights_worker(args):
    import cyp_pipeline as cp
    lo, hi = args
    return lo, [cp.CONTRACT.weight_sequence(G["o"], G["segs"][s], G["final"], G["allowed"]) for s in range(lo, hi)]


def main():
    ge.build()
    ge.load_package()
    from pb_starphase_amd import synth
    import oracle_ffi
    o = oracle_ffi.load()
    what = sys.argv[1:] or ["k1", "cyp"]
    doc = json.load(gzip.open(OUT, "rt")) if os.path.exists(OUT) else {}
    doc["generator"] = "tests/golden/make_fullsize.py"
    for key, fn in (("k1", k1_section), ("cyp", cyp_section)):
        if key in what:
            doc[key] = fn(o, synth)
            with gzip.open(OUT, "wt", compresslevel=9) as f:
                json.dump(doc, f, sort_keys=True, separators=(",", ":"))
            print(key, "done", flush=True)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
