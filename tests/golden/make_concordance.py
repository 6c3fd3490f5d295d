"""Generates tests/golden/concordance.json.gz: what the REFERENCE-CALL-PATTERN CPU port (tests/cpu_port_seeded.py for HLA, tests/cpu_port_cyp.py for CYP2D6:
the reference's own sequence of alignment calls on the minimap2 restatement oracle/mm2.c, consensus = oracle/consensus.c) computes on the BASELINE workloads,
so that tests/test_gpu_concordance.py can hold the HIP path to it on the GPU box without spending its CPU minutes:

  hla      configs[1], 10,000 reads (synth.Config2Workload seed 1000): per read the seeded winner (allele, nm, span), per gene the two consensus strings and
           the diplotype (database ids)
  cyp      the six configs[2] scenarios at 2,000 reads (rng 7): per read the region hits, the final consensuses + labels, the minimum-edit consensus sets of
           every region segment, the call strings
  k2       score_read's per-allele numbers (src/hla/caller.rs:1411-1510: what the reference prints into hla_debug.json) of the port for the four consensuses of the hla
           section: for every allele of the gene (nm, unmapped) at the cDNA and the DNA level, and the winner
  cohort   the configs[4] samples named in tests/golden/cohort_samples.json (those whose library call differs from the simulated truth, plus controls): the
           port's HLA diplotype per (sample, gene)

Data only: inputs are regenerated from seeds on both sides.  Usage (CPU, ~15 min on 8 cores):  python tests/golden/make_concordance.py [hla] [cyp] [cohort]"""
import gzip
import json
import os
import sys
import time
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402

OUT = os.path.join(HERE, "concordance.json.gz")


def crc(s):
    return zlib.crc32(s.encode()) & 0xFFFFFFFF


def hla_section(o, synth):
    import cpu_port_seeded as cps
    fx = synth.HlaFixture()
    wl = synth.Config2Workload(fx, n_reads=10000, seed=1000)
    t0 = time.time()
    res, best, calls, cons, done = cps.run(o, fx, wl.reads, n_sample=len(wl.reads), budget_s=1e9)
    assert done == list(range(len(wl.reads)))
    win = [int(best[r][0]) for r in done]
    nm = [int(best[r][1][0]) if best[r][1] else -1 for r in done]
    span = [int(best[r][1][2] - best[r][1][1]) if best[r][1] else 0 for r in done]
    # realign_record's whole result per read, second stage in the reference's call pattern (cpu_port_seeded.record_mm2: gene_aligner.map + select_best_mapping,
    # src/hla/realigner.rs:231-317): [status, seg_start, seg_end, dna_offset, hpc_offset] (-1 where the record has none)
    recs = cps.G["records"]
    records = [[int(recs[r].get(k, -1)) for k in ("status", "seg_start", "seg_end", "dna_offset", "hpc_offset")] for r in done]
    return {"workload": "synth.Config2Workload(HlaFixture(), n_reads=10000, seed=1000)", "seconds": time.time() - t0,
            "winner": win, "nm": nm, "span": span, "records": records,
            "calls": {fx.genes[g]: sorted(int(x) for x in calls[g]) for g in calls}, "call_ids": {fx.genes[g]: sorted(fx.ids[int(x)] for x in calls[g] if x >= 0) for g in calls},
            "consensus": {fx.genes[g]: list(cons[g]) for g in cons},
            "truth": {fx.genes[g]: sorted(int(a) for (gg, _c, _d, a) in wl.consensus if gg == g) for g in range(len(fx.genes))},
            "cpu": {k: res[k] for k in ("k1_seeded_ms_per_read_one_thread", "k1_all_cores_s", "genes_wall_s", "cpu_s", "cores")}}


def cyp_section(o, synth):
    import cpu_port_cyp as cpc
    import cyp_cases_real as cr
    cfg, gene_def = cr.load_db()
    locus = synth.Chr22Locus(cfg, gene_def, seed=3)
    db, ccfg = cpc.tables(cfg, gene_def, locus)
    out = {"workload": "synth.Chr22Locus(cfg, gene_def, seed=3).sample(default_rng(7), haps, 2000) for the six scenarios of tests/cyp_cases_real.py", "scenarios": {}}
    for name, haps, expected in cr.scenarios(locus):
        reads = locus.sample(np.random.default_rng(7), haps, 2000)
        st = {}
        res, tm = cpc.run(o, db, ccfg, reads, stages=st)
        regions = [[[int(h[k]) for k in ("template_idx", "start", "end", "nm", "unmapped")] for h in hits] for hits in st["regions"]]
        ed = np.asarray(st["ed"])
        allowed = st["allowed"]                 # what weight_sequence saw (is_allowed_label before the chains mark consensuses without unique support, caller.rs:574-583)
        min_sets = []
        for s in range(len(ed)):
            row = [int(x) for x in ed[s]]
            m = min(row) if row else 0
            min_sets.append([int(m), [c for c in range(len(row)) if row[c] == m], int(st["kept"][s])])
        out["scenarios"][name] = {
            "expected_truth": expected, "n_reads": len(reads), "status": int(res["status"]),
            "hap": [res.get("hap1", ""), res.get("hap2", "")], "core": [res.get("core1", ""), res.get("core2", "")], "score": res.get("score"),
            "consensus": res.get("consensus", []), "labels": [[int(t), s] for t, s in res.get("labels", [])], "allowed": allowed,
            "regions": regions, "min_ed_sets": min_sets, "n_consensus_inputs": int(st.get("n_inputs", 0)),
            "cpu": {k: (round(v, 3) if isinstance(v, float) else v) for k, v in tm.items()}}
        print(name, res["status"], res.get("hap1"), "/", res.get("hap2"), "truth", expected, {k: round(v, 1) for k, v in tm.items()}, flush=True)
    return out


def k2_section(o, synth, hla):
    """the port's score_read on the hla section's own consensuses: per consensus the winner and, allele by allele in database order, (cDNA nm, cDNA unmapped, DNA nm,
    DNA unmapped) -- -1 where the level has no mapping (the lengths are the alleles' own)"""
    import cpu_port_seeded as cps
    from concurrent.futures import ThreadPoolExecutor
    fx = synth.HlaFixture()
    jobs = [(g, name, c) for g, name in enumerate(fx.genes) for c in hla["consensus"][name] if c]

    def one(job):
        g, name, c = job
        st = []
        best = cps.type_consensus_mm2(o, fx, g, c, synth, stats_out=st)
        idx, arr = st[0]
        return {"gene": name, "consensus_crc": crc(c), "winner": int(best), "first_allele": int(idx[0]), "n_alleles": len(idx),
                "alleles_are_consecutive": bool(idx == list(range(idx[0], idx[0] + len(idx)))),
                "nm_unmapped": [[int(arr[k][0][1]), int(arr[k][0][2]), int(arr[k][1][1]), int(arr[k][1][2])] for k in range(len(idx))]}
    t0 = time.time()
    with ThreadPoolExecutor(max_workers=len(jobs)) as ex:               # (the C routine releases the GIL: ctypes)
        rows = list(ex.map(one, jobs))
    return {"workload": "the consensus strings of the hla section, typed by tests/cpu_port_seeded.py type_consensus_mm2 (omm_hla_score_read, a = 5)", "seconds": time.time() - t0, "consensuses": rows}


def cohort_section(o, synth):
    """the HLA half of the configs[4] samples named in cohort_samples.json through the port: ~44 reads per gene, every allele of the gene typed against the
    consensuses (the expensive part: ~12 CPU-s per consensus)"""
    import cpu_port_seeded as cps
    fx = synth.HlaFixture()
    want = json.load(open(os.path.join(HERE, "cohort_samples.json")))
    out = {"workload": "bench.CohortShare sample s: default_rng(10000 + s); per gene two full-length alleles, 22 reads each (mean 7,000, sd 1,500, min overlap 2,500)",
           "samples": {}}
    for s in want["samples"]:
        rng = np.random.default_rng(10_000 + s)
        reads, truth = [], {}
        for g in range(len(fx.genes)):
            pick = sorted(rng.choice(fx.full_length_alleles(g), 2, replace=False).tolist())
            truth[g] = pick
            for a in pick:
                hap, st = fx.haplotype(g, a)
                reads += synth.simulate_reads(rng, hap, st, len(fx.dna[a]), 22, mean_len=7000, sd_len=1500, min_overlap=2500)
        res, best, calls, cons, done = cps.run(o, fx, reads, n_sample=len(reads), budget_s=1e9)
        out["samples"][str(s)] = {"calls": {fx.genes[g]: sorted(int(x) for x in calls[g]) for g in calls},
                                  "call_ids": {fx.genes[g]: sorted(fx.ids[int(x)] for x in calls[g] if x >= 0) for g in calls},
                                  "truth": {fx.genes[g]: [int(x) for x in truth[g]] for g in truth}, "n_reads": len(reads),
                                  "winner": [int(best[r][0]) for r in done]}
        print("cohort sample", s, out["samples"][str(s)]["call_ids"], "truth", {fx.genes[g]: [fx.ids[x] for x in truth[g]] for g in truth}, flush=True)
    return out


def main():
    ge.build()
    ge.load_package()
    from pb_starphase_amd import synth
    import oracle_ffi
    o = oracle_ffi.load()
    what = sys.argv[1:] or ["hla", "cyp", "k2", "cohort"]
    doc = json.load(gzip.open(OUT, "rt")) if os.path.exists(OUT) else {}
    doc["generator"] = "tests/golden/make_concordance.py"
    for key, fn in (("hla", hla_section), ("cyp", cyp_section), ("k2", lambda o_, s_: k2_section(o_, s_, doc["hla"])), ("cohort", cohort_section)):
        if key in what:
            doc[key] = fn(o, synth)
            with gzip.open(OUT, "wt", compresslevel=9) as f:
                json.dump(doc, f, sort_keys=True, separators=(",", ":"))
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
