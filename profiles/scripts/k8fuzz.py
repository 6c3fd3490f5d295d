"""longer hunt for K8 != oracle: prints the first differing random problem in full (run on the GPU box).  Seeds >= 1000: low-complexity haplotypes and noisier reads; seeds >= 2000: haplotypes 1-5 % apart, recombinant reads."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from pb_starphase_amd import synth
import oracle_ffi as of
from test_oracle_consensus import run_case
from test_gpu_consensus import gpu_cfg
import consensus_fuzz
oracle = of.load()
ctx = pkg.Context(0)
seeds = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else list(range(1, 41))
bad = 0
for seed in seeds:
    rng = np.random.default_rng(seed)
    for it in range(25):
        L, reads, offs, kw, two_pass = consensus_fuzz.problem(rng, seed, synth)
        exp = run_case(oracle, reads, offs, kw, two_pass)
        got = ctx.consensus(ctx.upload(reads), gpu_cfg(pkg, **kw), offsets=offs, two_pass=two_pass)
        keys = ["cons", "is_dual", "split_at", "nodes_expanded"]
        diff = [k for k in keys if got[k] != exp[k]] + [k for k in ("is_cons1", "score1", "score2") if got[k].tolist() != exp[k].tolist()]
        if diff:
            bad += 1
            print(f"seed {seed} it {it}: differs in {diff}; n_reads {len(reads)} L {L} kw {kw} offs {offs}")
            for k in keys:
                print("   ", k, "gpu", got[k] if k != "cons" else [len(c) if c else None for c in got[k]], "oracle", exp[k] if k != "cons" else [len(c) if c else None for c in exp[k]])
            print("    score1 gpu", got["score1"].tolist(), "oracle", exp["score1"].tolist())
            print("    score2 gpu", got["score2"].tolist(), "oracle", exp["score2"].tolist())
            if got["cons"][0] != exp["cons"][0]:
                a, b = got["cons"][0], exp["cons"][0]
                p = next((i for i in range(min(len(a), len(b))) if a[i] != b[i]), min(len(a), len(b)))
                print("    cons1 first difference at", p, a[max(0, p - 10):p + 10], b[max(0, p - 10):p + 10])
            if bad >= 3:
                sys.exit(1)
print("checked", len(seeds) * 25, "problems;", bad, "differ")
