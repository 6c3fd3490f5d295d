import numpy as np
R, L = 1024, 4096
raw = np.fromfile('gpurun_out/k8_dump.bin', dtype=np.uint64)
rec = 1 + R * L
n_chunks = len(raw) // rec
k = n_chunks // 2
total = int(raw[k * rec]); m = raw[k * rec + 1:(k + 1) * rec].reshape(L, R)[:, :min(total, R)]
print("reads", total)
for i in (2455, 2461, 161, 66, 1000, 3000):
    v = m[i]
    dt = (v & np.uint64(0xFFFFFF)).astype(float) / 100
    o = np.argsort(-dt)[:5]
    print("launch", i, "n", int((v[o[0]] >> np.uint64(45)) & np.uint64(511)), [(int(g), round(float(dt[g]), 1), "slow", int((v[g] >> np.uint64(24)) & np.uint64(511)), "e0", int((v[g] >> np.uint64(33)) & np.uint64(511)), "e1", int((v[g] >> np.uint64(54)) & np.uint64(1023))) for g in o])
# distribution of e0 over reads at launch 2455
v = m[2455]; e0 = ((v >> np.uint64(33)) & np.uint64(511)).astype(int); e1 = ((v >> np.uint64(54)) & np.uint64(1023)).astype(int)
print("e0 percentiles", np.percentile(e0[v != 0], [50, 90, 99, 100]), "e1", np.percentile(e1[v != 0], [50, 90, 99, 100]))
