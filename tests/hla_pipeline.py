"""The gene loop of diplotype_hla_batch (src/hla/caller.rs:642-1040) assembled from the CPU oracle's pieces -- the expected
result for sp_hla_diplotype_gene (test infrastructure)."""
import ctypes as C

import numpy as np

import hla_expected as hx
import oracle_ffi as of


def is_passing_dual(oracle, c1, c2, min_fraction=0.10, expected_maf=0.45, min_cdf=0.001):
    maf, cdf = C.c_double(0), C.c_double(0)
    oracle.L.osp_is_passing_dual.restype = C.c_int
    ok = oracle.L.osp_is_passing_dual(C.c_uint64(c1), C.c_uint64(c2), C.c_double(min_fraction), C.c_double(expected_maf), C.c_double(min_cdf),
                                      C.byref(maf), C.byref(cdf))
    return bool(ok), maf.value, cdf.value


def type_consensus(oracle, fx, g, cons, synth):
    """score_consensus + splice_read (src/hla/caller.rs:1258-1319,1518-1576) -> best allele index or -1"""
    if not cons:
        return -1
    ref = fx.gene_ref[g][fx.buffer:len(fx.gene_ref[g]) - fx.buffer]
    d, v = oracle.anchor(fx.gene_ref[g], cons)
    if v < 2:
        return -1
    al, ev = oracle.wfa(cons, ref, -d - fx.buffer, retry=2)
    if not al.ok or hx.score_value(al.a_len, al.nm, al.a_len - (al.a_end - al.a_start)) >= 1.0:
        return -1
    cigar = oracle.cigar(al, ev)
    bam = [(l, {7: 0, 8: 0, 1: 1, 2: 2}[op]) for l, op in cigar]
    if al.a_start:
        bam.insert(0, (al.a_start, 4))
    exons = [(e0 - fx.buffer, e1 - fx.buffer) for e0, e1 in fx.exons[g]]
    segs, _ = oracle.splice_read(al.b_start, bam, exons)
    spliced = "".join(cons[x:y] for x, y in segs)
    fwd = bool(fx.gene_fwd[g])
    e_dna = cons if fwd else synth.revcomp(cons)
    e_cdna = spliced if fwd else synth.revcomp(spliced)
    best, _stats = hx.k2_expected(oracle, fx, g, e_dna, e_cdna)
    return best


def is_hemizygous_better(oracle, dual, n, delta, normalized_coverage):
    """is_hemizygous_better (src/hla/caller.rs:1583-1653) on a DualConsensus-shaped result dict"""
    s1 = np.array([int(x) for x in dual["score1"]], np.int64)
    s2 = np.array([int(x) for x in dual["score2"]], np.int64)
    is1 = np.array(dual["is_cons1"], np.uint8)
    h, d = C.c_double(), C.c_double()
    oracle.L.osp_is_hemizygous_better.restype = C.c_int32
    r = oracle.L.osp_is_hemizygous_better(s1.ctypes.data_as(C.c_void_p), s2.ctypes.data_as(C.c_void_p), is1.ctypes.data_as(C.c_void_p), n,
                                          1 if dual["is_dual"] else 0, delta, 1 if normalized_coverage is not None else 0,
                                          float(normalized_coverage if normalized_coverage is not None else 0.0), C.byref(h), C.byref(d))
    return bool(r)


def diplotype_gene(oracle, fx, g, reads, k1, synth, min_count=3, min_fraction=0.10, delta=100, absent_capable=False, normalized_coverage=None, type_fn=None):
    """k1 = hx.k1_expected(...)[0] for `reads`.  Returns a dict with the fields of sp_hla_call + consensuses + is_cons1."""
    sel = [r for r, e in enumerate(k1) if e["status"] == 0 and e["gene"] == g]
    out = dict(status=0, n_reads=len(sel), allele1=-1, allele2=-1, typed1=-1, typed2=-1, cons1="", cons2="")
    if not sel:
        out["status"] = 1
        return out
    segs = [reads[r][k1[r]["seg_start"]:k1[r]["seg_end"]] for r in sel]
    hpcs = [oracle.hpc(s) for s in segs]
    run = lambda rd, offs, cfg: of.oracle_consensus(oracle, rd, offs, cfg)
    cfg = of.cons_config(min_count=min_count, min_af=min_fraction, dual_max_ed_delta=delta, early_termination=True, dual=True)

    def offsets(key, members=None):
        vals = [k1[r][key] for r in sel]
        mn = min(v for i, v in enumerate(vals) if members is None or members[i])
        return [None if v == mn else v - mn + 200 for v in vals]

    dual = of.dual_consensus_two_pass(run, hpcs, offsets("hpc_offset"), cfg)
    c1 = int(dual["is_cons1"].sum()); c2 = len(sel) - c1
    ok, maf, cdf = is_passing_dual(oracle, c1, c2, min_fraction) if dual["is_dual"] else (False, 0.0, 0.0)
    used_dna = False
    if not ok:
        dual = of.dual_consensus_two_pass(run, segs, offsets("dna_offset"), cfg)
        c1 = int(dual["is_cons1"].sum()); c2 = len(sel) - c1
        ok, maf, cdf = is_passing_dual(oracle, c1, c2, min_fraction) if dual["is_dual"] else (False, 0.0, 0.0)
        used_dna = True
    hemi = absent_capable and is_hemizygous_better(oracle, dual, len(sel), delta, normalized_coverage)
    if hemi:                                                                 # boiler-plate non-dual consensus (caller.rs:687-701)
        dual = dict(dual, is_dual=False, is_cons1=np.ones(len(sel), bool))
    single = of.cons_config(min_count=min_count, min_af=min_fraction, dual_max_ed_delta=delta, early_termination=True, dual=False)
    cons = []
    for which in ((True, False) if dual["is_dual"] else (True,)):
        members = [bool(x) == which for x in dual["is_cons1"]]
        if not any(members):
            cons.append("")
            continue
        offs = offsets("dna_offset", members)
        grp = [i for i, m in enumerate(members) if m]
        cons.append(of.oracle_consensus(oracle, [segs[i] for i in grp], [offs[i] for i in grp], single)["cons"][0])
    type_consensus_ = type_fn or type_consensus                  # (bench.py's CPU leg types with the minimap2 restatement: tests/cpu_port_seeded.py)
    both = None
    if dual["is_dual"] and getattr(type_consensus_, "threads", False):      # (the CPU port types the two consensuses side by side: one long C call each, the GIL is released)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(2) as ex:
            both = [f.result() for f in [ex.submit(type_consensus_, oracle, fx, g, c, synth) for c in cons[:2]]]
    t1 = both[0] if both else type_consensus_(oracle, fx, g, cons[0], synth)
    out.update(cons1=cons[0], typed1=t1, is_dual=int(dual["is_dual"]), dual_passed=0, counts1=c1, counts2=c2, maf=maf, cdf=cdf,
               used_dna_dual=int(used_dna), is_cons1=dual["is_cons1"])
    if dual["is_dual"]:
        t2 = both[1] if both else type_consensus_(oracle, fx, g, cons[1], synth)
        out.update(cons2=cons[1], typed2=t2, dual_passed=int(ok))
        if ok:
            out.update(allele1=t1, allele2=t2)
        elif c1 > c2:
            out.update(allele1=t1, allele2=t1)
        else:
            out.update(allele1=t2, allele2=t2)
    else:
        out.update(allele1=-2 if hemi else t1, allele2=t1)
    out["is_hemizygous"] = int(hemi)
    return out
