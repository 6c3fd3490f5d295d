# what the step workgroups of a CYP2D6 call hold their CUs for: per launch the sum over workgroups (8 consecutive reads, one per wave) of the slowest wave's time, as the reads are
# dealt to the workgroups today and as they would be if the reads were dealt in the order of their times (parts build; launch pairs).   bash profiles/scripts/k8_wg_occupancy.sh <scenario>
SC=${1:-3}
rm -f gpurun_out/k8_dump.bin
SP_K8_PERSISTENT=0 SP_K8_DUMP=$PWD/gpurun_out/k8_dump.bin SP_LIB_PATH=$PWD/build/variants/lib_parts.so python profiles/scripts/cyp_kernels.py $SC 2>&1 | grep -E "total ms"
python - <<'PY'
import numpy as np
R, L = 4096, 512
raw = np.fromfile('gpurun_out/k8_dump.bin', dtype=np.uint64)
rec = 1 + R * L * 2
n_chunks = len(raw) // rec
tot_now = tot_sorted = tot_prev = tot_mean = 0.0; n_l = 0
for k in range(n_chunks // 2, n_chunks):
    total = int(raw[k * rec]); m = raw[k * rec + 1:(k + 1) * rec].reshape(L, R, 2)[:, :min(total, R), 0]
    prev = None
    for i in range(L):
        v = m[i]
        if not (v != 0).any(): continue
        dt = (v & np.uint64(0xFFFFFF)).astype(float) / 100          # us per read slot (0: the read took no part in this launch)
        n = len(dt); pad = (-n) % 8
        d8 = np.concatenate([dt, np.zeros(pad)]).reshape(-1, 8)
        tot_now += d8.max(1).sum()
        s8 = np.sort(np.concatenate([dt, np.zeros(pad)]))[::-1].reshape(-1, 8)
        tot_sorted += s8.max(1).sum()
        if prev is not None and len(prev) == n:                      # dealt by the times of the launch BEFORE (what a re-sort every step could know)
            o = np.argsort(-prev, kind="stable")
            p8 = np.concatenate([dt[o], np.zeros(pad)]).reshape(-1, 8)
            tot_prev += p8.max(1).sum()
        else:
            tot_prev += d8.max(1).sum()
        tot_mean += dt.sum() / 8
        prev = dt; n_l += 1
print(f"launches {n_l}: CU time held by the step workgroups, summed: as dealt today {tot_now/1e3:.1f} ms | dealt by this launch's own times {tot_sorted/1e3:.1f} ms | dealt by the times of the launch before {tot_prev/1e3:.1f} ms | sum of wave times / 8 {tot_mean/1e3:.1f} ms")
PY
rm -f gpurun_out/k8_dump.bin
