# kernel trace of one CYP2D6 call (scenario $1 of tests/cyp_cases_real.py, default 3 = *4+*68/*1): durations of the three consensus kernels
# by percentile and the idle gaps between consecutive kernels of the loop
export TMPDIR=/tmp
SC=${1:-3}
OUT=gpurun_out/prof_cyp_trace
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 profiles/scripts/cyp_kernels.py $SC > $OUT/trace.log 2>&1
grep -E "total ms|cons_" $OUT/trace.log
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_cyp_trace/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    for k in ('cons_step', 'cons_reduce', 'cons_control', 'cons_finalize', 'cons_setup'):
        if k in n: return k
    return None
prev = None
dur, gap = {}, {}
for r in rows[len(rows) // 2:]:                      # the second (timed) call
    k = short(r['Kernel_Name'])
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if k:
        key = k + (' grid %s' % r.get('Grid_Size', '?') if k == 'cons_step' else '')
        dur.setdefault(key, []).append(e - s)
        if prev and prev[0]:
            gap.setdefault(prev[0] + '->' + k, []).append(s - prev[1])
    prev = (k, e)
def pct(v, p): v = sorted(v); return v[min(len(v) - 1, int(p * len(v)))] / 1000
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print(f"{k:32s} n={len(v):5d} mean {sum(v)/len(v)/1000:7.2f} us  p10 {pct(v,.1):6.2f} p50 {pct(v,.5):6.2f} p90 {pct(v,.9):6.2f} max {max(v)/1000:7.2f}  total {sum(v)/1e6:7.2f} ms")
for k, v in gap.items():
    print(f"gap {k:28s} n={len(v):5d} mean {sum(v)/len(v)/1000:6.2f} us p50 {pct(v,.5):6.2f} p90 {pct(v,.9):6.2f}  total {sum(v)/1e6:6.2f} ms")
PY
rm -rf $OUT/trace
