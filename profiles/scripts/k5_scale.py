"""K5 at scale: sp_cyp_best_chain_pair with P = 10^3 .. 10^4 enumerated chains x R reads (synthetic chain problem: several CYP2D6 / hybrid
consensuses that the reads chain in every order, so the enumeration explodes as it does on duplication-rich samples).
Run on the GPU box:  python profiles/scripts/k5_scale.py [n_d6 ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from pb_starphase_amd import synth


def main():
    ctx = pkg.Context(0)
    out = []
    for n_d6, n_reads in [(int(a), 1000) for a in (sys.argv[1:] or ["4", "6", "8"])]:
        prob = synth.chain_pair_problem(n_d6, n_reads, np.random.default_rng(7))
        best = None
        for rep in range(2):
            ctx.profile_reset(); ctx.synchronize()
            t0 = time.perf_counter()
            rc, res = ctx.cyp_best_chain_pair(**prob)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        P = res.n_possible
        pairs = P * (P + 1) // 2
        ms_pairs, ms_tab = ctx.profile_get("k5_pairs")[0], ctx.profile_get("k5_chain_reads")[0]
        rec = {"d6_alleles": n_d6, "reads": n_reads, "chains_P": P, "pairs": pairs, "pairs_scored": int(res.n_pairs_scored), "status": rc,
               "wall_ms": 1e3 * best, "k5_pairs_ms": ms_pairs, "k5_chain_reads_ms": ms_tab,
               "pairs_per_s": pairs / (ms_pairs * 1e-3) if ms_pairs else None,
               "algorithmic_bytes": pairs * n_reads * 96, "GBps_streaming_model": pairs * n_reads * 96 / (ms_pairs * 1e-3) / 1e9 if ms_pairs else None,
               "score": res.score}
        print(json.dumps(rec), flush=True)
        out.append(rec)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "k5_scale.json"), "w"), indent=1)


main()
