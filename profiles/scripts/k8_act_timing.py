"""timing build with -DSP_K8_DBG_EDITS: where the placements of late reads spend their time (first batch of the second call)"""
import numpy as np
R, L = 4096, 1024
raw = np.fromfile('gpurun_out/k8_dump.bin', dtype=np.uint64)
rec = 1 + R * L
n_chunks = len(raw) // rec
k = n_chunks // 2
total = int(raw[k * rec]); m = raw[k * rec + 1:(k + 1) * rec].reshape(L, R)[:, :min(total, R)]
v = m[m != 0]
placed = ((v >> np.uint64(42)) & np.uint64(1)) != 0
dt = (v & np.uint64(0xFFFFFF)).astype(float) / 100
stage = ((v >> np.uint64(24)) & np.uint64(511)).astype(float) * 0.32
find = ((v >> np.uint64(33)) & np.uint64(511)).astype(float) * 0.32
catch = ((v >> np.uint64(54)) & np.uint64(1023)).astype(float) * 0.32
print(f"records {len(v)}, with a placement {placed.sum()}")
p = placed
print(f"placing waves: total {dt[p].mean():.1f} us mean | staging+packing {stage[p].mean():.1f} | start search {find[p].mean():.1f} | catch-up {catch[p].mean():.1f} (both states of the read summed)")
print(f"others: total {dt[~p].mean():.1f} us mean")
