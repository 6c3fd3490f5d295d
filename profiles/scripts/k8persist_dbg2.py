import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from pb_starphase_amd import synth
ctx = pkg.Context(0)
rng = np.random.default_rng(1)
cfgc = pkg.ffi.sp_cons_config(3, 100, 1, 1, 400, 50, 0.10, 20, 10, 1000, 0)
bases = ["".join(rng.choice(list("ACGT"), 600)) for _ in range(40)]
# k problems of 64 reads (8 blocks each): total blocks 8k + controllers k
for k in [int(x) for x in os.environ.get('KS', '8,16,24,28,30,32,40').split(',')]:
    reads = [synth.hifi_errors(rng, bases[j]) for j in range(k) for _ in range(64)]
    S = ctx.upload(reads)
    probs = [dict(reads=S, read_idx=np.arange(j * 64, (j + 1) * 64, dtype=np.uint32), cfg=cfgc) for j in range(k)]
    ctx.profile_reset(); t0 = time.time()
    try:
        outs = ctx.consensus_batch(probs)
        print(k, "problems", 8 * k, "blocks ok; persistent", ctx.profile_get("cons_persistent_batches")[2], round(time.time() - t0, 3), flush=True)
    except Exception as e:
        print(k, "problems", 8 * k, "blocks FAILED", str(e)[:80], round(time.time() - t0, 3), flush=True)
