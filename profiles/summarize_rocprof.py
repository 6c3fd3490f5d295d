#!/usr/bin/env python3
"""Summarise a profiles/run_rocprof.sh output directory: per-kernel time and PMC sums per kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats)")
for f in find("trace/**/*kernel_stats.csv"):
    for row in csv.DictReader(open(f)):
        print("  %-60s calls=%s total_ns=%s avg_ns=%s pct=%s" % (row.get("Name", "")[:60], row.get("Calls"), row.get("TotalDurationNs"),
                                                                 row.get("AverageNs"), row.get("Percentage")))
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_lds"):
    files = find(sub + "/**/*counter_collection.csv")
    if not files:
        print("== %s: no counter file" % sub)
        continue
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    for f in files:
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")[:48]
            acc[k][row.get("Counter_Name")] += float(row.get("Counter_Value", 0) or 0)
            cnt[(k, row.get("Counter_Name"))] += 1
    print("== %s (sum over dispatches; n = dispatches)" % sub)
    for k in sorted(acc):
        for c in sorted(acc[k]):
            print("  %-48s %-24s sum=%.6g n=%d per_dispatch=%.6g" % (k, c, acc[k][c], cnt[(k, c)], acc[k][c] / cnt[(k, c)]))

# HBM traffic of the dominant kernel per main launch, for bench.py's roofline.traffic (MI355X_MICROARCH.md, HBM: FETCH_SIZE and
# WRITE_SIZE are KB; gfx950 tallies 128-byte read requests at 64 bytes, so FETCH_SIZE is doubled).  The deep passes of
# k1_cells_kernel are small launches of the same kernel: only dispatches within 10x of the largest one count as main launches.
import json
import hashlib


def k1_source_sha16():
    """the kernel the counters belong to: bench.py marks the committed numbers stale when these sources have changed since"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in ("pb-starphase_amd/csrc/sp_hla.hip", "pb-starphase_amd/csrc/sp_wfa.hip.h"):
        h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()[:16]


vals = {}
for sub, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    v = []
    for f in find(sub + "/**/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            if "k1_cells_kernel" in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:   # both instantiations; the 10x rule below drops the deep passes
                v.append(float(row.get("Counter_Value", 0) or 0))
    # one row per (dispatch, XCD/instance) may exist: group by Dispatch_Id when available
    vals[counter] = v
if vals.get("FETCH_SIZE") and vals.get("WRITE_SIZE"):
    def per_main(v):
        big = [x for x in v if x * 10 >= max(v)]
        return sum(big) / len(big), len(big)
    f, nf = per_main(vals["FETCH_SIZE"]); w, nw = per_main(vals["WRITE_SIZE"])
    rec = {"kernel": "k1_cells_kernel", "fetch_kb_per_launch": f, "write_kb_per_launch": w, "main_launches": nf,
           "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0, "k1_source_sha16": k1_source_sha16(),
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (profiles/run_rocprof.sh); KB -> bytes; FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B)"}
    json.dump(rec, open(os.path.join(out, "traffic_k1_cells.json"), "w"), indent=1)
    print("== traffic", rec)

# VALU wave-instructions of the dominant kernel per main launch, for bench.py's roofline_valu (SQ_INSTS_VALU of the pmc_sq pass)
v, sa = [], []
for f in find("pmc_sq/**/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "k1_cells_kernel" in row.get("Kernel_Name", "") and row.get("Counter_Name") == "SQ_INSTS_VALU":
            v.append(float(row.get("Counter_Value", 0) or 0))
        if "k1_cells_kernel" in row.get("Kernel_Name", "") and row.get("Counter_Name") == "SQ_INSTS_SALU":
            sa.append(float(row.get("Counter_Value", 0) or 0))
if v:
    big = [x for x in v if x * 10 >= max(v)]
    bigs = [x for x in sa if x * 10 >= max(sa)] if sa else []
    rec = {"kernel": "k1_cells_kernel", "sq_insts_valu_per_launch": sum(big) / len(big), "sq_insts_salu_per_launch": (sum(bigs) / len(bigs)) if bigs else None,
           "main_launches": len(big), "k1_source_sha16": k1_source_sha16(),
           "method": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU (pass pmc_sq of profiles/run_rocprof.sh), mean over the main launches"}
    json.dump(rec, open(os.path.join(out, "valu_k1_cells.json"), "w"), indent=1)
    print("== valu", rec)


# The consensus step kernel (K8, the dominant kernel of the headline since round 5) per bench step, for bench.py's roofline: the counter passes run the CYP2D6 context with a
# launch pair per step (cons_step_kernel<8>: the persistent pair cannot run one kernel at a time); sums over every dispatch of the pass divided by the pass's steps (warm-up included)
def cons_source_sha16():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return hashlib.sha256(open(os.path.join(root, "pb-starphase_amd/csrc/sp_consensus.hip"), "rb").read()).hexdigest()[:16]


steps_in_pass = int(os.environ.get("SP_PROF_STEPS", "7"))
tot = defaultdict(float); ndisp = defaultdict(int)
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_lds"):
    for f in find(sub + "/**/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            if "cons_step_kernel<8>" in row.get("Kernel_Name", "") or "cons_step_wide_kernel<8>" in row.get("Kernel_Name", ""):
                tot[row.get("Counter_Name")] += float(row.get("Counter_Value", 0) or 0); ndisp[row.get("Counter_Name")] += 1
if tot.get("SQ_INSTS_VALU"):
    rec = {"kernel": "cons_step_wide_kernel<8> + cons_step_kernel<8> (the CYP2D6 contexts' consensus steps as launch pairs: one body, two register budgets)", "steps_in_pass": steps_in_pass,
           "dispatches_per_bench_step": ndisp["SQ_INSTS_VALU"] / steps_in_pass,
           "sq_insts_valu_per_bench_step": tot["SQ_INSTS_VALU"] / steps_in_pass, "sq_insts_salu_per_bench_step": tot.get("SQ_INSTS_SALU", 0.0) / steps_in_pass,
           "sq_insts_lds_per_bench_step": tot.get("SQ_INSTS_LDS", 0.0) / steps_in_pass,
           "fetch_kb_per_bench_step": tot.get("FETCH_SIZE", 0.0) / steps_in_pass, "write_kb_per_bench_step": tot.get("WRITE_SIZE", 0.0) / steps_in_pass,
           "hbm_bytes_per_bench_step": (2.0 * tot.get("FETCH_SIZE", 0.0) + tot.get("WRITE_SIZE", 0.0)) * 1024.0 / steps_in_pass,
           "cons_source_sha16": cons_source_sha16(),
           "method": "rocprofv3 --pmc passes of profiles/run_rocprof.sh (SQ_INSTS_VALU / SALU / LDS, FETCH_SIZE, WRITE_SIZE in separate passes; KB -> bytes, FETCH_SIZE doubled: "
                     "gfx950 tallies 128-B requests at 64 B), summed over every cons_step_wide_kernel<8> / cons_step_kernel<8> dispatch of the pass and divided by the pass's %d bench steps (the six configs[2] scenarios + the warm-up step)" % steps_in_pass}
    # the same kernel in the kernel trace of the unprofiled command (bench.py's roofline block quotes it beside its own device-clock figure)
    for f in find("trace/**/*kernel_stats.csv"):
        for row in csv.DictReader(open(f)):
            if "cons_step_wide_kernel<8>" in row.get("Name", ""):
                rec["rocprof_avg_launch_us"] = round(float(row.get("AverageNs", 0) or 0) / 1e3, 2); rec["rocprof_calls_in_trace"] = int(row.get("Calls", 0) or 0)
    # how much of the device a step launch holds (the counter passes run one kernel at a time): waves launched, the cycles they were resident, the cycles the
    # shader engines were busy at all and the cycles the GPU was active, all summed over the dispatches
    for name in ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY"):
        if name in tot:
            rec[name.lower() + "_per_bench_step"] = tot[name] / steps_in_pass
    if tot.get("SQ_WAVE_CYCLES") and tot.get("GRBM_GUI_ACTIVE") and tot.get("SQ_WAVES"):
        # MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_ACTIVE_* count quad-cycles, GRBM_GUI_ACTIVE is the sum over the 8 XCDs
        active_cycles = tot["GRBM_GUI_ACTIVE"] / 8.0
        rec["occupancy"] = {"waves_per_dispatch": tot["SQ_WAVES"] / ndisp["SQ_WAVES"], "active_cycles_per_dispatch": active_cycles / ndisp["GRBM_GUI_ACTIVE"],
                            "mean_resident_waves": 4.0 * tot["SQ_WAVE_CYCLES"] / active_cycles,
                            "wave_slots_at_2_per_simd": 256 * 4 * 2, "mean_resident_over_slots": 4.0 * tot["SQ_WAVE_CYCLES"] / active_cycles / (256 * 4 * 2),
                            "issuing_share_of_resident_wave_cycles": (tot.get("SQ_ACTIVE_INST_ANY", 0.0) / tot["SQ_WAVE_CYCLES"]) if tot.get("SQ_ACTIVE_INST_ANY") else None,
                            "note": "SQ_WAVE_CYCLES x 4 / (GRBM_GUI_ACTIVE / 8): the waves resident on average while a step launch is active (one kernel at a time under the counter pass); "
                                    "the launch's length is its slowest wave's, most waves are done in a sixth of it"}
    json.dump(rec, open(os.path.join(out, "counters_cons_step.json"), "w"), indent=1)
    print("== cons_step", rec)
