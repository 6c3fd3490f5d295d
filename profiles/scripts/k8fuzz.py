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
oracle = of.load()
ctx = pkg.Context(0)
seeds = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else list(range(1, 41))
bad = 0
for seed in seeds:
    rng = np.random.default_rng(seed)
    for it in range(25):
        L = int(rng.integers(150, 700))
        h1 = "".join(rng.choice(list("ACGT"), L))
        perr = 0.004
        if seed >= 1000:
            # low-complexity haplotypes (homopolymers, tandem repeats) and noisier reads: wavefronts keep several tips for many columns
            parts, tot = [], 0
            while tot < L:
                if rng.random() < 0.5:
                    motif = "".join(rng.choice(list("ACGT"), int(rng.integers(1, 5)))); seg = (motif * 40)[:int(rng.integers(8, 60))]
                else:
                    seg = "".join(rng.choice(list("ACGT"), int(rng.integers(5, 40))))
                parts.append(seg); tot += len(seg)
            h1 = "".join(parts)[:L]
            perr = float(rng.choice([0.004, 0.01, 0.03]))
        h2 = synth.mutate(rng, h1, int(rng.integers(1, 4)), int(rng.integers(0, 2)), int(rng.integers(0, 2))) if L > 200 else h1
        if seed >= 2000:
            # two haplotypes a few percent apart (as two gene copies are), longer, and reads that switch from one to the other: the worse state
            # of a read falls behind by dozens of edits, is dropped at dual_max_ed_delta, or draws level again behind a switch
            L = int(rng.integers(400, 1500)); h1 = "".join(rng.choice(list("ACGT"), L))
            k = max(3, int(L * float(rng.choice([0.01, 0.03, 0.05])) / 1.0)); k = min(k, (L - 40) // 12 - 1)
            h2 = synth.mutate(rng, h1, k - k // 8 - k // 8, k // 8, k // 8)
        reads, offs = [], []
        for _ in range(int(rng.integers(1, 14))):
            hap = h1 if rng.random() < 0.5 else h2
            if seed >= 2000 and rng.random() < 0.3:
                x = int(rng.integers(L // 4, 3 * L // 4)); other = h2 if hap is h1 else h1
                hap = hap[:x] + other[min(x, len(other)):]
            a = int(rng.integers(0, L // 3)) if rng.random() < 0.5 else 0
            b = int(rng.integers(2 * L // 3, len(hap) + 1))
            reads.append(synth.hifi_errors(rng, hap[a:b], p_sub=perr, p_ins=perr, p_del=perr))
            offs.append(None if a == 0 else a + int(rng.integers(0, 40)))
        if all(o is not None for o in offs):
            offs[0] = None
        kw = dict(early_termination=bool(rng.integers(0, 2)), dual=True, min_count=int(rng.integers(1, 4)), min_af=float(rng.choice([0.1, 0.25])),
                  dual_max_ed_delta=int(rng.choice([2, 20, 100])), offset_window=int(rng.choice([60, 120, 400] if seed < 2000 else [60, 120, 400, 700])), offset_compare_length=int(rng.choice([20, 50, 64])))
        two_pass = bool(rng.integers(0, 2))
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
