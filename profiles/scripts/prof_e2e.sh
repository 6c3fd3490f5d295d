#!/bin/bash
# kernel trace + stats of the reads -> diplotype step (bench.py's headline).  Run on the GPU box: bash profiles/scripts/prof_e2e.sh <tag>
set -u
TAG=${1:-run}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $OUT/trace.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    print(f)
    for r in rows[:18]:
        print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):7d} total {float(r["TotalDurationNs"])/1e6:9.3f} ms avg {float(r["AverageNs"])/1e3:9.2f} us  {float(r["Percentage"]):5.1f}%')
PY
