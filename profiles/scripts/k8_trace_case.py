"""one case of tests/consensus_cases.py through the library (a trace build prints its expansions)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
pkg = ge.load_package()
from pb_starphase_amd import synth
import oracle_ffi, consensus_cases
from test_gpu_consensus import gpu_cfg, run_case
oracle = oracle_ffi.load()
ctx = pkg.Context(0)
fx = synth.HlaFixture()
cs, _ = consensus_cases.cases(fx, synth, oracle)
want = sys.argv[1]
for name, reads, offs, kw, two_pass in cs:
    if name != want: continue
    exp = run_case(oracle, reads, offs, kw, two_pass)
    got = ctx.consensus(ctx.upload(reads), gpu_cfg(pkg, **kw), offsets=offs, two_pass=two_pass)
    print(name, kw, "n_reads", len(reads), "cons equal", got["cons"] == exp["cons"], "score1 gpu", got["score1"].tolist()[:12], "oracle", exp["score1"].tolist()[:12], "nodes", got.get("nodes_expanded"), exp.get("nodes_expanded"))
