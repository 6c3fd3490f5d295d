#!/bin/bash
# kernel trace of two bench steps, printed as a timeline with the gaps between dispatches (what the host does between launches)
set -u
OUT=gpurun_out/timeline
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > $OUT/trace.log 2>&1
python3 - <<'P'
import csv,glob
f=glob.glob('gpurun_out/timeline/trace/*/*_kernel_trace.csv')[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k1_cells_kernel<false, false>' in r['Kernel_Name']]
j=idx[-1]
while j>0 and 'sp_anchor' not in rows[j]['Kernel_Name']: j-=1
t0=int(rows[j]['Start_Timestamp']); prev=t0
for r in rows[j:j+34]:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    print('%9.1f us  gap %7.1f  dur %8.1f  %s' % ((s-t0)/1e3,(s-prev)/1e3,(e-s)/1e3,r['Kernel_Name'][:56]))
    prev=e
P
