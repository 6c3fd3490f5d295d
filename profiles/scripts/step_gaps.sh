# kernel trace of bench steps: where the GPU idles inside a step (largest gaps between consecutive kernels, by neighbour names)
export TMPDIR=/tmp
OUT=gpurun_out/prof_stepgaps
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra-legs > $OUT/trace.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/prof_stepgaps/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return n.split('(')[0].split('<')[0][:28]
# the last three steps = from the third-last 'sp_anchor_kernel' that follows a k2_scan... simpler: take the last 3/5 of the trace by time
t0, t1 = int(rows[0]['Start_Timestamp']), int(rows[-1]['End_Timestamp'])
# steps begin at an anchor kernel that follows a k2_scan_kernel (or the start)
starts = [i for i, r in enumerate(rows) if 'k1_init_kernel' in r['Kernel_Name']]
print('step starts found:', len(starts))
if len(starts) >= 4:
    a, b = starts[-3] - 1, len(rows)
    seg = rows[a:b]
    # cut the tail (microbench kernels) at the last k2_scan
    last = max(i for i, r in enumerate(seg) if 'k2_scan' in r['Kernel_Name'])
    seg = seg[:last + 1]
    span = int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg)
    print(f"last steps: span {span/1e6:.2f} ms, kernels busy {busy/1e6:.2f} ms, idle {(span-busy)/1e6:.2f} ms, kernels {len(seg)}")
    gaps = collections.defaultdict(lambda: [0, 0])
    for x, y in zip(seg, seg[1:]):
        g = int(y['Start_Timestamp']) - int(x['End_Timestamp'])
        if g > 0:
            k = short(x['Kernel_Name']) + ' -> ' + short(y['Kernel_Name'])
            gaps[k][0] += g; gaps[k][1] += 1
    for k, (g, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:12]:
        print(f"  {g/1e6:7.3f} ms  n={n:4d}  {k}")
    # the last step alone, in time order: every gap above 40 us with what ran before and after it
    a = starts[-1] - 1
    seg1 = [r for r in rows[a:] ]
    last = max(i for i, r in enumerate(seg1) if 'k2_scan' in r['Kernel_Name'])
    seg1 = seg1[:last + 1]
    t0 = int(seg1[0]['Start_Timestamp'])
    print('last step, gaps > 40 us:')
    for x, y in zip(seg1, seg1[1:]):
        g = int(y['Start_Timestamp']) - int(x['End_Timestamp'])
        if g > 40000:
            print(f"  at {(int(x['End_Timestamp'])-t0)/1e6:7.3f} ms  gap {g/1e3:7.1f} us   {short(x['Kernel_Name'])} ({(int(x['End_Timestamp'])-int(x['Start_Timestamp']))/1e3:.1f} us) -> {short(y['Kernel_Name'])}")
PY
rm -rf $OUT/trace
