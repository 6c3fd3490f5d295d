"""longer hunt for K6 != oracle (run on the GPU box): random diplotype problems of many shapes, small variant spaces included (alternatives of an
OR-group observed together, phase sets, SV labels); prints the first differing problems.  usage: k6fuzz.py <n> [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
import oracle_ffi as of
import variant_glue as vg
from test_gpu_variant import RandomProblem, gpu_struct
oracle = of.load()
ctx = pkg.Context(0)
n = int(sys.argv[1]); rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for k in range(n):
    n_vars = int(rng.choice([6, 8, 12, 24, 48]))
    p = RandomProblem(rng, n_vars=n_vars, n_haps=int(rng.integers(1, 70)), n_obs=int(rng.integers(0, min(n_vars, 12) + 1)))
    exp = vg.oracle_solve(oracle, p)
    try:
        got = ctx.variant_solve(gpu_struct(pkg, p))
    except pkg.StarphaseError as e:
        got = ("error", str(e))
    if got != exp:
        bad += 1
        print(f"problem {k}: n_vars {n_vars} haps {len(p.haps)} obs {p.obs_var.tolist()} gt {p.obs_gt.tolist()} ps {p.obs_ps.tolist()} sv {p.obs_sv.tolist()}")
        print("   gpu   ", got if got[0] == "error" else (got[0], got[1][:6]))
        print("   oracle", (exp[0], exp[1][:6]))
        if bad >= 5:
            break
print("checked", k + 1, "problems;", bad, "differ")
