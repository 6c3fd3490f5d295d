"""where the 32-sample cohort call (BASELINE configs[4] per GPU) spends its time: HIP-event and host-clock totals per name"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")          # as bench.py: the streams of a call each get a hardware queue
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from pb_starphase_amd import synth
import bench
ctx = pkg.Context(0)
if len(sys.argv) > 1:
    ctx.set_option("hla_split_streams", int(sys.argv[1]))
fx = synth.HlaFixture()
db = fx.make_db(pkg, ctx)
names = ["anchor_k1", "anchor_k2", "anchor_type", "k1_cells", "k1_cells_deep", "k1_reduce", "k1_finalize", "cons_steps", "type_consensus_ref", "k2_cells_cdna", "k2_cells_dna",
         "k2_scan"] + ["host:" + k for k in ("hla_select", "hla_segments", "hla_setup", "hla_dual_hpc", "hla_dual_dna", "hla_groups", "hla_typing", "k8_prologue", "k8_loop",
                                              "k8_result_wait", "k8_epilogue", "k1_total", "k2_setup", "k2_launch", "k2_wait")]
bench.cohort_leg(pkg, ctx, fx, db, reps=0)
ctx.profile_reset()
out = bench.cohort_leg(pkg, ctx, fx, db, reps=0)          # one warm call + the timed one inside
print({k: out[k] for k in ("ms", "samples_per_s", "calls_equal_truth")})
for n in names:
    ms, launches, cells = ctx.profile_get(n)
    print(f"{n:24s} {ms:9.3f} ms  launches {launches:8.1f}")
for n in ("cons_windows", "cons_cut_windows", "cons_expansions"):
    print(n, ctx.profile_get(n)[2])
