"""rocprofv3 kernel trace of the bench (profiles/run_rocprof.sh, pass 1): the gaps between the dependent kernels of every consensus chain -- step kernel end -> control kernel start and
control kernel end -> next step kernel start on the same stream -- as a distribution, and which kernels of OTHER streams were running while a long gap lasted.
usage: trace_gaps.py <kernel_trace.csv>"""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Stream_Id"]), int(r["Queue_Id"]), r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) * max(1, int(r["Grid_Size_Y"]))))
rows.sort(key=lambda x: x[3])
t_lo = rows[len(rows) // 3][3]                       # (skip the set-up and warm-up third)
by_stream = collections.defaultdict(list)
for x in rows:
    by_stream[x[0]].append(x)
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
gaps = {"step->control": [], "control->step": []}
for s, ks in by_stream.items():
    for a, b in zip(ks, ks[1:]):
        if a[3] < t_lo:
            continue
        an, bn = short(a[2]), short(b[2])
        if an.startswith("cons_step") and bn.startswith("cons_control_kernel"):
            gaps["step->control"].append((b[3] - a[4], a, b))
        elif an.startswith("cons_control_kernel") and bn.startswith("cons_step"):
            gaps["control->step"].append((b[3] - a[4], a, b))
print("streams", len(by_stream), "queues", len({x[1] for x in rows}), "kernels", len(rows))
for name, g in gaps.items():
    v = sorted(x[0] for x in g)
    if not v:
        continue
    q = lambda f: v[min(len(v) - 1, int(f * len(v)))] / 1e3
    print("%s: n %d, mean %.1f us, median %.1f, p75 %.1f, p90 %.1f, p99 %.1f, max %.1f; share of the gap time in gaps > 100 us: %.2f" % (
        name, len(v), sum(v) / len(v) / 1e3, q(0.5), q(0.75), q(0.9), q(0.99), v[-1] / 1e3, sum(x for x in v if x > 100e3) / max(1, sum(v))))
# which kernels of other streams overlap the long gaps (> 100 us), by overlapped time
for name, g in gaps.items():
    over = collections.Counter(); tot = 0
    for gap, a, b in g:
        if gap <= 100e3:
            continue
        tot += gap
        for x in rows:
            if x[0] == a[0] or x[4] <= a[4] or x[3] >= b[3]:
                continue
            over[short(x[2])] += min(x[4], b[3]) - max(x[3], a[4])
    print(name, "long gaps: total %.1f ms; kernels of other streams running meanwhile (ms of overlap):" % (tot / 1e6), [(k, round(v / 1e6, 1)) for k, v in over.most_common(8)])
# the same for short gaps, as a control
for name, g in gaps.items():
    over = collections.Counter(); tot = 0
    for gap, a, b in g:
        if gap > 30e3:
            continue
        tot += gap
        for x in rows:
            if x[0] == a[0] or x[4] <= a[4] or x[3] >= b[3]:
                continue
            over[short(x[2])] += min(x[4], b[3]) - max(x[3], a[4])
    print(name, "short gaps (< 30 us): total %.1f ms; meanwhile:" % (tot / 1e6), [(k, round(v / 1e6, 1)) for k, v in over.most_common(6)])
