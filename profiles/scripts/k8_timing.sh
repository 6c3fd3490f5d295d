rm -f gpurun_out/k8_dump.bin
SP_K8_DUMP=$PWD/gpurun_out/k8_dump.bin SP_LIB_PATH=$PWD/build/variants/lib_timing.so python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra-legs > gpurun_out/bt.json 2> gpurun_out/bt.err
tail -2 gpurun_out/bt.err
python - <<'PY'
import numpy as np
raw = np.fromfile('gpurun_out/k8_dump.bin', dtype=np.uint64)
rec = 1 + 256 * 16384
for k in range(len(raw) // rec):
    total = int(raw[k * rec]); m = raw[k * rec + 1:(k + 1) * rec].reshape(256, 16384)[:, :total]
    used = [i for i in range(256) if m[i].any()]
    if not used: continue
    mx, mean, sl_max, sl_mean, mt, nb, top = [], [], [], [], [], [], []
    for i in used:
        row = m[i][m[i] != 0]
        t = (row & np.uint64(0xFFFFFFFF)).astype(float) / 100
        sc = ((row >> np.uint64(32)) & np.uint64(255)).astype(float); mtv = ((row >> np.uint64(40)) & np.uint64(255)).astype(float)
        mx.append(t.max()); mean.append(t.mean()); sl_max.append(sc.max()); sl_mean.append(sc.mean()); mt.append(mtv.mean()); nb.append(int(row[0] >> np.uint64(52)))
        w = np.argmax(t); top.append((t[w], sc[w], mtv[w], int((row[w] >> np.uint64(48)) & np.uint64(1))))
    print(f"batch {k}: reads {total} launches {len(used)} window {np.mean(nb):.0f} | slowest wave us mean {np.mean(mx):.1f} p90 {np.percentile(mx,90):.1f} | mean wave {np.mean(mean):.1f} | slow cols max {np.mean(sl_max):.1f} mean {np.mean(sl_mean):.2f} multi-tip {np.mean(mt):.2f}")
    print("   slowest waves (us, slow cols, multi-tip, placed):", [tuple(round(float(x),1) for x in t) for t in top[:12]])
    # correlation of wave time with slow columns over all waves of launch used[len//2]
    i = used[len(used)//2]; row = m[i][m[i] != 0]; t = (row & np.uint64(0xFFFFFFFF)).astype(float)/100; sc = ((row >> np.uint64(32)) & np.uint64(255)).astype(float)
    for lo, hi in ((0,0),(1,2),(3,6),(7,15),(16,64)):
        sel = (sc >= lo) & (sc <= hi)
        if sel.any(): print(f"   launch {i}: waves with {lo}-{hi} slow columns: {sel.sum()} mean {t[sel].mean():.1f} us max {t[sel].max():.1f}")
PY
rm -f gpurun_out/k8_dump.bin
