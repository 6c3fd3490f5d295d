# kernel trace of one bench step: durations of the three consensus kernels and the idle gaps between consecutive kernels of the loop
export TMPDIR=/tmp
OUT=gpurun_out/prof_gaps
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $OUT/trace.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_gaps/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    for k in ('cons_step', 'cons_reduce', 'cons_control', 'cons_finalize', 'cons_setup'):
        if k in n: return k
    return None
prev = None
dur, gap = {}, {}
for r in rows:
    k = short(r['Kernel_Name'])
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if k:
        dur.setdefault(k, []).append(e - s)
        if prev and prev[0]:
            gap.setdefault(prev[0] + '->' + k, []).append(s - prev[1])
    prev = (k, e)
import statistics as st
for k, v in dur.items(): print(f"{k:14s} n={len(v):5d} mean {st.mean(v)/1000:7.2f} us  median {st.median(v)/1000:7.2f}  total {sum(v)/1e6:7.2f} ms")
for k, v in gap.items(): print(f"gap {k:28s} n={len(v):5d} mean {st.mean(v)/1000:6.2f} us median {st.median(v)/1000:6.2f}  total {sum(v)/1e6:6.2f} ms")
PY
rm -rf $OUT/trace
