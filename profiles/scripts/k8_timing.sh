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
    rows = np.concatenate([m[i][m[i] != 0] for i in used])
    t = (rows & np.uint64(0xFFFFFFFF)).astype(float) / 100
    sc = ((rows >> np.uint64(32)) & np.uint64(255)).astype(float)
    f = ((rows >> np.uint64(40)) & np.uint64(255)).astype(float) * 256; c = ((rows >> np.uint64(48)) & np.uint64(255)).astype(float) * 256; v = ((rows >> np.uint64(56)) & np.uint64(255)).astype(float) * 256
    print(f"batch {k}: reads {total} launches {len(used)} | mean wave {t.mean():.1f} us, slow columns {sc.mean():.2f} | shader clocks per wave: fast-path part {f.mean():.0f}, column pushes {c.mean():.0f}, votes {v.mean():.0f}")
    sel = sc > 0
    print(f"   per slow column (waves with any): column push {c[sel].sum()/sc[sel].sum():.0f} clocks, votes {v[sel].sum()/sc[sel].sum():.0f} clocks; fast-path part per wave {f[sel].mean():.0f} clocks")
PY
rm -f gpurun_out/k8_dump.bin
