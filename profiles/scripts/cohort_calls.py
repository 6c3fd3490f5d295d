"""GPU half of the cohort concordance: the HLA / CYP2D6 calls the library makes for every sample of BASELINE configs[4]
(bench.py's cohort, 256 samples, all on this one GPU), with the truth beside them -> gpurun_out/cohort_calls.json.  tests/golden/make_concordance.py
runs the samples whose call differs from the truth (and a few that do not) through the reference-call-pattern CPU port."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402


def main():
    pkg = ge.load_package()
    from pb_starphase_amd import synth, shard
    import cyp_cases_real as cr
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    ctx = pkg.Context(0)
    fx = synth.HlaFixture()
    db = fx.make_db(pkg, ctx)
    cfg, gene_def = cr.load_db()
    locus = synth.Chr22Locus(cfg, gene_def, seed=3)
    cdb = pkg.ffi.CypDb(ctx, cfg, gene_def, locus.sequence, locus.start)
    scen = cr.scenarios(locus)
    panel = bench.VariantPanel(pkg)
    sh = bench.CohortShare(pkg, fx, locus, scen, panel, list(range(n)))
    table, good = sh.step(pkg, ctx, db, cdb, shard, None, 0)
    genes = len(fx.genes)
    out = {"samples": n, "hla_equal_truth": int(good[0]), "cyp_equal_truth": int(good[1]), "hla": [], "cyp": []}
    by = {(int(r["sample"]), int(r["gene"])): (int(r["allele1"]), int(r["allele2"])) for r in table}
    for s in range(n):
        for g in range(genes):
            a = sorted(by[(s, g)])
            t = sh.hla_truth[(s, g)]
            ok = all(bench.same_allele(fx, x, y) for x, y in zip(a, t))
            out["hla"].append({"sample": s, "gene": g, "call": a, "truth": [int(x) for x in t], "ok": bool(ok), "call_ids": [fx.ids[x] if x >= 0 else None for x in a],
                               "truth_ids": [fx.ids[x] for x in t]})
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "cohort_calls.json"), "w"))
    print("HLA", good[0], "/", n * genes, "CYP2D6", good[1], "/", n, "misses:", [(h["sample"], h["gene"], h["call_ids"], h["truth_ids"]) for h in out["hla"] if not h["ok"]])


if __name__ == "__main__":
    main()
