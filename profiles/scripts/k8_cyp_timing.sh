# per-read records of the consensus window launches of one CYP2D6 call (timing build: build/variants/lib_timing.so, -DSP_K8_TIMING -DSP_K8_DBG_READS=4096):
# what the slowest wave of every window launch was doing
SC=${1:-3}
rm -f gpurun_out/k8_dump.bin
SP_K8_PERSISTENT=0 SP_K8_DUMP=$PWD/gpurun_out/k8_dump.bin SP_LIB_PATH=$PWD/build/variants/lib_timing.so python profiles/scripts/cyp_kernels.py $SC 2>&1 | grep -E "total ms|cons_steps"
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
        rows.append((k, i, dt[w], float(np.median(dt)), (x >> 24) & 511, (x >> 33) & 511, (x >> 42) & 1, (x >> 43) & 3, (x >> 45) & 511, ((x >> 54) & 1023) * 1024 / 2400.0, len(v),
                     int((((v >> np.uint64(42)) & np.uint64(1)) != 0).sum())))
a = np.array(rows, float)
print(f"window/init launches recorded: {len(a)}; sum of the slowest wave per launch {a[:,2].sum()/1e3:.1f} ms, of the median wave {a[:,3].sum()/1e3:.1f} ms")
for name, sel in (("slowest wave placed a late read", a[:, 6] == 1), ("slowest wave placed none, >= 32 slow columns", (a[:, 6] == 0) & (a[:, 4] >= 32)),
                  ("slowest wave placed none, 1..31 slow columns", (a[:, 6] == 0) & (a[:, 4] > 0) & (a[:, 4] < 32)), ("no slow column", (a[:, 6] == 0) & (a[:, 4] == 0))):
    b = a[sel]
    if len(b): print(f"  {name:48s} launches {len(b):5d}  slowest-wave time sum {b[:,2].sum()/1e3:7.2f} ms mean {b[:,2].mean():7.1f} us | mean slow cols {b[:,4].mean():6.1f} multi-tip {b[:,5].mean():6.1f} window {b[:,8].mean():6.1f} column-push us {b[:,9].mean():6.1f}")
sel = (a[:, 6] == 0) & (a[:, 4] > 0)
if sel.sum() > 3:
    p = np.polyfit(a[sel][:, 4], a[sel][:, 2], 1); print(f"  slowest wave without placement: {p[0]:.2f} us per slow column + {p[1]:.1f} us")
o = np.argsort(-a[:, 2])[:12]
print("  slowest launches: (chunk, launch, us, median wave us, slow cols, multi-tip, placed, mode, window, column-push us, waves, waves that placed)")
for i in o: print("   ", [round(float(x), 1) for x in a[i]])
PY
rm -f gpurun_out/k8_dump.bin
