"""Event counters of k1_cells at BASELINE config-2 scale, from a profiling build of the library (-DSP_K1_STATS on sp_hla.hip only,
see k1_stats.sh).  Prints cells executed / skipped / finished, DP steps, cooperative-extension iterations and the cap histogram."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge

pkg = ge.load_package()
from pb_starphase_amd import synth

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
fx = synth.HlaFixture()
wl = synth.Config2Workload(fx, n_reads=n_reads, seed=1000)
ctx = pkg.Context(0)
db = fx.make_db(pkg, ctx)
reads = ctx.upload(wl.reads)
L = pkg.ffi.lib()
L.sp_debug_wfa_stats.argtypes = [C.c_void_p, C.c_int32]
db.realign_reads(reads)                       # warm-up (bounds are per call, nothing carries over)
L.sp_debug_wfa_stats(None, 1)
db.realign_reads(reads)
ctx.synchronize()
st = np.zeros(64, np.uint64)
L.sp_debug_wfa_stats(st.ctypes.data, 0)
st = [int(x) for x in st]
out = {"executed": st[0], "skipped": st[1], "finished": st[2], "steps": st[3], "steps_in_finished": st[4], "long_lanes": st[5],
       "long_iterations": st[6], "cap_sum": st[7], "resumed": st[8], "rounds_saved": st[9], "long_stretches_from_kept_scan": st[13], "after_chain": st[10], "no_state": st[11], "state_past_cap": st[12], "failed_by_steps": st[16:32], "finished_by_steps": st[32:48], "executed_by_cap": st[48:64]}
print(json.dumps(out))
