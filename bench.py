#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on BASELINE.json configs[1] (HLA-A/-B, 10k synthetic HiFi reads, 1 x MI355X).

A step = one pass of the HLA hot path over one batch that is already resident in HBM:
   K1  sp_hla_realign_reads   10,000 reads x every DNA allele of the bundled IMGT/HLA DB (anchor, cells, reduce, finalize)
   K2  sp_hla_score_consensus 4 consensuses (2 genes x 2 haplotypes) x every allele of the gene (cDNA + DNA)
value = reads diplotyped per second, whole job (all ranks).  N > 1: one process per GPU, each rank owns one
synthetic sample (weak scaling, no data-path collective); the per-gene calls are gathered with one RCCL all_gather.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes(fx, wl, votes_gene_of_read):
    """bytes_per_cell = ceil(Lq/4) + ceil(Lt/4) + 32 with Lq = allele length and Lt = the read window
    a cell can touch (allele length + 64-diagonal band, clipped to the read).  Summed over the cells one K1 launch
    executes: every DNA allele of the gene(s) the read anchors in."""
    alen = np.array([len(s) for s in fx.dna], np.int64)
    per_gene = []
    for g in range(len(fx.genes)):
        m = (fx.gene_of == g) & (alen > 0)
        per_gene.append(alen[m])
    total, cells = 0, 0
    for r, read in enumerate(wl.reads):
        for g in votes_gene_of_read[r]:
            la = per_gene[g]
            lt = np.minimum(len(read), la + 64)
            total += int(((la + 3) // 4 + (lt + 3) // 4 + 32).sum())
            cells += len(la)
    return total, cells


_CB = {}


def _cpu_worker(args):
    """one worker process: whole-read K1 searches on its slice of reads until the time budget is spent"""
    import ctypes as C
    lo, hi, budget_s = args
    o, L = _CB["o"], _CB["L"]
    done, best, t0 = 0, [], time.perf_counter()
    for r in range(lo, hi):
        re = o.encode(_CB["reads"][r])
        ncell = C.c_int64(0)
        b = L.osp_hla_k1_read(re.ctypes.data_as(C.c_void_p), len(re), len(_CB["refs"]), _CB["ref_ptr"], _CB["ref_len"].ctypes.data_as(C.c_void_p),
                              _CB["n_all"], _CB["al_ptr"], _CB["al_len"].ctypes.data_as(C.c_void_p), _CB["gene_of"].ctypes.data_as(C.c_void_p),
                              _CB["off"].ctypes.data_as(C.c_void_p), None, C.byref(ncell))
        best.append((r, b))
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    return done, time.perf_counter() - t0, best


def cpu_baseline(fx, wl, budget_s=12.0):
    """The oracle ("port": same algorithm, scalar C) timed on a bounded sample of the same reads, BEFORE the GPU is touched
    (worker processes are forked).  Reported with every host core in use -- the reference itself is single-threaded
    (src/cli/diplotype.rs:185-191), so the one-thread rate is given as well."""
    import ctypes as C
    import multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi
    o = oracle_ffi.load()
    L = o.L
    L.osp_hla_k1_read.restype = C.c_int32
    refs = [o.encode(s) for s in fx.gene_ref]
    n_all = len(fx.ids)
    enc = [o.encode(fx.dna_fwd(a)) if fx.dna[a] else np.zeros(0, np.uint8) for a in range(n_all)]
    off = np.full(n_all, -2 ** 31, np.int32)
    for a in range(n_all):                       # untimed set-up (the reference builds its index once too)
        if len(enc[a]):
            d, v = o.anchor(refs[int(fx.gene_of[a])], enc[a])
            if v >= 16:
                off[a] = d
    _CB.update(o=o, L=L, reads=wl.reads, refs=refs, n_all=n_all, enc=enc, off=off,
               ref_ptr=(C.c_void_p * len(refs))(*[r.ctypes.data for r in refs]), ref_len=np.array([len(r) for r in refs], np.int32),
               al_ptr=(C.c_void_p * n_all)(*[(e.ctypes.data if len(e) else None) for e in enc]),
               al_len=np.array([len(e) for e in enc], np.int32), gene_of=fx.gene_of.astype(np.int32))
    cores = max(1, min(len(os.sched_getaffinity(0)), 64, len(wl.reads) // 8))
    per = len(wl.reads) // cores                  # reads reserved per worker (not exhausted inside the budget at bench sizes)
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(w * per, min(len(wl.reads), (w + 1) * per), budget_s) for w in range(cores)])
    wall = time.perf_counter() - t0
    done = sum(r[0] for r in res)
    single = res[0][0] / res[0][1] if res[0][1] > 0 else 0.0
    best = dict(b for r in res for b in r[2])
    return {"value": done / max(r[1] for r in res), "unit": "reads/s", "cores": cores, "kind": "port", "single_thread_value": single,
            "sample": f"{done} reads of the same batch over {cores} forked workers, K1 search only (anchor + every allele cell + acceptance), {wall:.1f} s wall"}, best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=10000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--e2e-steps", type=int, default=3)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    pkg = ge.load_package()
    from pb_starphase_amd import synth, shard
    fx = synth.HlaFixture()
    wl = synth.Config2Workload(fx, n_reads=args.reads, seed=1000 + rank)
    cb, cpu_best = (None, None)
    if not args.no_cpu_baseline and world == 1:
        cb, cpu_best = cpu_baseline(fx, wl)          # forks workers: must happen before anything touches the GPU
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a gfx950 GPU: the HIP extension is the product, there is no CPU fallback")
    # one process per GPU; SP_BENCH_BACKEND=gloo lets the N > 1 code path be exercised on a single-GPU box
    # (ranks then share device 0 -- RCCL itself refuses two ranks on one device)
    backend = os.environ.get("SP_BENCH_BACKEND", "nccl")
    device_index = local_rank if backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(device_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    ctx = pkg.Context(device_index)
    db = fx.make_db(pkg, ctx)
    t_up = time.perf_counter()
    reads = ctx.upload(wl.reads)
    t_up = time.perf_counter() - t_up

    def step():
        out = db.realign_reads(reads)
        calls = [b for b, _n in db.score_consensus_batch([(g, cons_dna, cons_cdna) for (g, cons_dna, cons_cdna, _a) in wl.consensus])]
        if world > 1:
            # RCCL: the only exchange step of the path -- one gather of the per-(sample, gene) call records
            rec = np.zeros(len(fx.genes), shard.CALL_DTYPE)
            for g in range(len(fx.genes)):
                pair = [b for (gg, _c, _d, _a), b in zip(wl.consensus, calls) if gg == g]
                rec[g] = (rank, g, pair[0], pair[1] if len(pair) > 1 else pair[0])
            shard.gather_calls(rec, device="cuda" if backend == "nccl" else "cpu", same_count=True)
        return out, calls

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    for _ in range(args.warmup):
        out, calls = step()
    ctx.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, calls = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # concordance with the synthetic truth (calls) -- informational
    k1_gene_ok = float(np.mean([out[r]["gene"] == wl.read_truth[r][0] for r in range(len(wl.reads))]))
    k1_realigned = float(np.mean(out["status"] == 0))
    k2_ok = sum(1 for (g, _c, _d, a), b in zip(wl.consensus, calls) if b == a or (b >= 0 and fx.cdna[b] == fx.cdna[a] and fx.dna[b] == fx.dna[a]))

    kernel_ms = {k: ctx.profile_get(k)[0] / max(1, args.steps) for k in
                 ("anchor", "k1_cells", "k1_cells_deep", "k1_reduce", "k1_finalize", "k2_cells_cdna", "k2_cells_dna", "k2_scan")}
    ms_cells, launches, cells = ctx.profile_get("k1_cells")
    genes_of_read = [[int(out[r]["gene"])] if out[r]["gene"] >= 0 else [] for r in range(len(wl.reads))]
    alg_bytes, alg_cells = algorithmic_bytes(fx, wl, genes_of_read)
    avg_ms = ms_cells / max(1, launches)
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0

    # ---- the same batch from reads to diplotype: K1, then the gene driver (segments + HPC on the device, K8 consensus, K2 typing).
    # Reported beside `value`; `value` stays on the path north_star names (read -> allele scoring + consensus -> allele scoring).
    e2e = None
    if not args.no_end_to_end:
        genes = list(range(len(fx.genes)))

        def full():
            o = db.realign_reads(reads)
            return db.diplotype_genes(genes, reads, o)[0]

        full()
        ctx.profile_reset()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.e2e_steps):
            gene_calls = full()
        barrier()
        dt2 = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([dt2], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt2 = float(t.item())
        truth = {g: sorted(a for (gg, _c, _d, a) in wl.consensus if gg == g) for g in genes}
        same = lambda a, b: a == b or (a >= 0 and b >= 0 and fx.cdna[a] == fx.cdna[b] and fx.dna[a] == fx.dna[b])
        ok = 0
        for g, (c, _c1, _c2) in enumerate(gene_calls):
            got, want = sorted([c.allele1, c.allele2]), (truth[g] * 2)[:2]
            ok += all(same(a, b) for a, b in zip(got, sorted(want)))
        e2e = {"value": args.reads * world * args.e2e_steps / dt2, "unit": "reads/s", "ms_per_step": 1e3 * dt2 / args.e2e_steps, "steps": args.e2e_steps,
               "workload": "the same reads -> K1 realignment -> per-gene dual consensus (HPC, DNA fallback) + per-group consensus (K8) -> typing (K2) -> diplotype",
               "kernel_ms": {k: ctx.profile_get(k)[0] / args.e2e_steps for k in ("anchor", "k1_cells", "hla_segments", "cons_steps", "k2_cells_cdna", "k2_cells_dna", "k2_scan")},
               "diplotypes_equal_truth": f"{ok}/{len(genes)} genes"}

    traffic, traffic_note = None, None
    tfile = os.path.join(ROOT, "profiles", "r01", "traffic_k1_cells.json")
    if os.path.exists(tfile) and args.reads == 10000:
        rec = json.load(open(tfile))
        traffic, traffic_note = rec["hbm_bytes_per_launch"], "from the committed rocprofv3 PMC passes of this workload (" + rec["method"] + ")"

    if rank == 0:
        total_reads = args.reads * world * args.steps
        line = {
            "metric": "HiFi reads/sec diplotyped (HLA-A + HLA-B hot path: read->allele realignment + consensus->allele scoring)",
            "value": total_reads / dt, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8 (2-bit packed bases, int32 wavefront DP, f64 score ratios)", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: HLA-A/-B, %d synthetic HiFi reads/GPU vs bundled IMGT/HLA DB v0.14.1 "
                                   "(18,461 alleles, 11,199 with DNA), 4 consensuses" % args.reads,
                       "reads_per_gpu": args.reads, "alleles": len(fx.ids), "parallelism": "one sample per GPU, RCCL all_gather of calls"},
            "roofline": {"bound": "hbm", "kernel": "k1_cells_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                         "algorithmic_bytes_per_launch": alg_bytes, "cells_per_launch": alg_cells, "avg_launch_ms": avg_ms,
                         "note": "integer-DP kernel bound by VALU issue (rocprofv3 SQ_ACTIVE_INST_VALU ~ 90 % of the issue slots, profiles/r01), DB served from "
                                 "L2/MALL.  'achieved' is the streaming model of SURVEY 8(d) over EVERY (read, allele) cell the launch settles; "
                                 "exact prefix sharing settles ~2/3 of them without running them and each XCD keeps its eighth of the database in L2, so the figure can exceed the HBM peak -- "
                                 "'traffic' is what actually crossed HBM"},
            "kernel_ms": kernel_ms,
            "concordance": {"k1_gene_correct": k1_gene_ok, "k1_realigned": k1_realigned, "k2_truth_calls": f"{k2_ok}/{len(calls)}"},
            "pcie_inclusive_upload_s": t_up,
            "end_to_end": e2e,
        }
        if cb is not None:
            agree = sum(1 for i, b in cpu_best.items() if b == int(out[i]["best_allele"]))
            cb["calls_identical_to_gpu"] = f"{agree}/{len(cpu_best)}"
            line["cpu_baseline"] = cb
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
