# where the slowest wave of every consensus step launch spends its time (parts build: build/variants/lib_parts.so, -DSP_K8_TIMING -DSP_K8_DBG_PARTS -DSP_K8_DBG_READS=4096), one
# CYP2D6 call, launch pairs.   bash profiles/scripts/k8_step_parts.sh <scenario>
SC=${1:-3}
rm -f gpurun_out/k8_dump.bin
SP_K8_PERSISTENT=0 SP_K8_DUMP=$PWD/gpurun_out/k8_dump.bin SP_LIB_PATH=$PWD/build/variants/lib_parts.so python profiles/scripts/cyp_kernels.py $SC 2>&1 | grep -E "total ms|cons_steps|cons_"
python - <<'PY'
import numpy as np
R, L = 4096, 512
raw = np.fromfile('gpurun_out/k8_dump.bin', dtype=np.uint64)
rec = 1 + R * L * 2
n_chunks = len(raw) // rec
rows = []
for k in range(n_chunks // 2, n_chunks):                      # the second (timed) call
    total = int(raw[k * rec]); m = raw[k * rec + 1:(k + 1) * rec].reshape(L, R, 2)[:, :min(total, R)]
    for i in range(L):
        ok = m[i, :, 0] != 0
        v = m[i, ok, 0]; p = m[i, ok, 1]
        if not len(v): continue
        dt = (v & np.uint64(0xFFFFFF)).astype(float) / 100
        w = int(np.argmax(dt)); x = int(v[w]); y = int(p[w])
        parts = [(y & 1023) * 0.02, ((y >> 10) & 4095) * 0.02, ((y >> 22) & 4095) * 0.02, ((y >> 34) & 4095) * 0.02, ((y >> 46) & 4095) * 0.02, (y >> 58) & 63]
        rows.append([k, i, dt[w], float(np.median(dt)), (x >> 24) & 511, (x >> 33) & 511, (x >> 42) & 1, (x >> 43) & 3, (x >> 45) & 511, len(v)] + parts)
a = np.array(rows, float)
names = {1: "init", 2: "window", 3: "expand"}
print(f"step launches recorded: {len(a)}; slowest wave summed {a[:,2].sum()/1e3:.1f} ms, median wave {a[:,3].sum()/1e3:.1f} ms")
for md in (1, 2, 3):
    b = a[a[:, 7] == md]
    if not len(b): continue
    print(f"  {names[md]:7s} launches {len(b):5d}: slowest wave mean {b[:,2].mean():6.1f} us (sum {b[:,2].sum()/1e3:6.2f} ms), median wave {b[:,3].mean():6.1f} us; waves {b[:,9].mean():7.1f}")
    if md == 3:
        for nk in sorted(set(b[:, 4].astype(int))):
            c = b[b[:, 4] == nk]
            print(f"      {nk:2d} children: {len(c):5d} launches, slowest {c[:,2].mean():6.1f} us, median {c[:,3].mean():6.1f} us, verified bases replayed in front (slowest wave) {c[:,8].mean():5.1f}")
    if md == 2:
        other = b[:, 2] - b[:, 10:15].sum(1)
        print(f"      the slowest wave, mean us: loads in front {b[:,10].mean():5.1f} | clean runs {b[:,11].mean():5.1f} ({b[:,15].mean():4.1f} iterations) | column pushes {b[:,12].mean():5.1f} ({b[:,4].mean():4.1f} slow columns, {b[:,5].mean():4.1f} multi-tip) | votes {b[:,13].mean():5.1f} | worse state ahead {b[:,14].mean():5.1f} | rest (lookahead, final cost, store) {other.mean():5.1f}; placed a read: {int(b[:,6].sum())}")
        for nm, sel in (("placed a read", b[:, 6] == 1), ("placed none", b[:, 6] == 0)):
            c = b[sel]
            if len(c):
                o = c[:, 2] - c[:, 10:15].sum(1)
                print(f"      {nm:14s} {len(c):5d}: total {c[:,2].mean():6.1f} | front {c[:,10].mean():5.1f} | clean {c[:,11].mean():5.1f} ({c[:,15].mean():4.1f} it) | push {c[:,12].mean():5.1f} ({c[:,4].mean():4.1f} cols) | votes {c[:,13].mean():5.1f} | ahead {c[:,14].mean():5.1f} | rest {o.mean():5.1f}")
PY
rm -f gpurun_out/k8_dump.bin
