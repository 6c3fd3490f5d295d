#!/bin/bash
# kernel trace of one end-to-end step: duration of the consensus step launches and the idle gaps between them
set -u
OUT=gpurun_out/cons_gaps
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --e2e-steps 1 > $OUT/trace.log 2>&1
python3 - <<'P'
import csv,glob
f=glob.glob('gpurun_out/cons_gaps/trace/*/*_kernel_trace.csv')[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
prev=None; durs=[]; gaps=[]
for r in rows:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    if 'cons_step_kernel' in r['Kernel_Name']:
        durs.append(e-s)
        if prev is not None and prev[1]=='step': gaps.append(s-prev[0])
        prev=(e,'step')
    else:
        prev=(e,'other')
import statistics as st
print('steps', len(durs), 'mean dur us', st.mean(durs)/1e3, 'median', st.median(durs)/1e3)
print('gaps', len(gaps), 'mean gap us', st.mean(gaps)/1e3, 'median', st.median(gaps)/1e3, 'p90', sorted(gaps)[int(.9*len(gaps))]/1e3)
P
