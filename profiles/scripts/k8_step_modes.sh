# per-launch records of the consensus step kernel of one CYP2D6 call (timing build: build/variants/lib_timing.so, -DSP_K8_TIMING -DSP_K8_DBG_READS=4096), launch pairs:
# the slowest and the median wave of every step launch by mode (window / expansion) and size
SC=${1:-3}
rm -f gpurun_out/k8_dump.bin
SP_K8_PERSISTENT=0 SP_K8_DUMP=$PWD/gpurun_out/k8_dump.bin SP_LIB_PATH=$PWD/build/variants/lib_timing.so python profiles/scripts/cyp_kernels.py $SC 2>&1 | grep -E "total ms|cons_steps|cons_"
python - <<'PY'
import numpy as np
R, L = 4096, 1024
raw = np.fromfile('gpurun_out/k8_dump.bin', dtype=np.uint64)
rec = 1 + R * L
n_chunks = len(raw) // rec
rows = []
for k in range(n_chunks // 2, n_chunks):                      # the second (timed) call
    total = int(raw[k * rec]); m = raw[k * rec + 1:(k + 1) * rec].reshape(L, R)[:, :min(total, R)]
    for i in range(L):
        v = m[i][m[i] != 0]
        if not len(v): continue
        dt = (v & np.uint64(0xFFFFFF)).astype(float) / 100
        w = int(np.argmax(dt)); x = int(v[w])
        rows.append((k, i, dt[w], float(np.median(dt)), float(np.percentile(dt, 90)), (x >> 24) & 511, (x >> 33) & 511, (x >> 42) & 1, (x >> 43) & 3, (x >> 45) & 511, len(v),
                     int((((v >> np.uint64(42)) & np.uint64(1)) != 0).sum()), float(dt.sum())))
a = np.array(rows, float)
print(f"step launches recorded: {len(a)} in {n_chunks - n_chunks // 2} batches; slowest wave summed {a[:,2].sum()/1e3:.1f} ms, median wave {a[:,3].sum()/1e3:.1f} ms, p90 wave {a[:,4].sum()/1e3:.1f} ms")
names = {1: "init", 2: "window", 3: "expand"}
for md in (1, 2, 3):
    b = a[a[:, 8] == md]
    if not len(b): continue
    print(f"  {names[md]:7s} launches {len(b):5d}: slowest wave mean {b[:,2].mean():6.1f} us (sum {b[:,2].sum()/1e3:6.2f} ms), median wave {b[:,3].mean():6.1f}, p90 {b[:,4].mean():6.1f}; waves {b[:,10].mean():7.1f}; slowest placed a read in {int(b[:,7].sum())} launches; n (window / pre) mean {b[:,9].mean():5.1f}")
    for lo, hi in ((0, 1), (1, 9), (9, 33), (33, 129), (129, 512)):
        c = b[(b[:, 9] >= lo) & (b[:, 9] < hi)]
        if len(c): print(f"      n in [{lo},{hi}): {len(c):5d} launches, slowest {c[:,2].mean():6.1f} us, median {c[:,3].mean():6.1f} us, slowest placed {int(c[:,7].sum())}, slow cols of the slowest {c[:,5].mean():5.1f}")
    c = b[b[:, 7] == 0]
    if len(c): print(f"      launches whose slowest wave placed no read: {len(c)}, slowest {c[:,2].mean():6.1f} us median {c[:,3].mean():6.1f}")
    c = b[b[:, 11] == 0]
    if len(c): print(f"      launches in which no wave placed a read: {len(c)}, slowest {c[:,2].mean():6.1f} us median {c[:,3].mean():6.1f}")
PY
rm -f gpurun_out/k8_dump.bin
