#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric "HiFi reads/sec diplotyped (HLA+CYP2D6)" on one node of MI355X.

Headline (`value`): BASELINE configs[1] from READS TO DIPLOTYPE -- HLA-A/-B, 10,000 synthetic HiFi reads already resident in HBM
    K1  sp_hla_realign_reads      every read x every DNA allele of the bundled IMGT/HLA DB (anchor, cells, reduce, finalize)
    --  sp_hla_diplotype_genes    segments + homopolymer compression on the device, K8 dual consensus (HPC, DNA fallback) and
                                  per-group consensus, K2 typing of the consensuses against every allele, het / hom decision
Beside it in the same JSON line:
    scoring_only   K1 + K2 on truth consensuses (round 1's headline; the read -> allele and consensus -> allele scoring without K8)
    roofline       k1_cells_kernel over the cells it actually EXECUTED; roofline_valu: its real limiter
    cyp2d6         BASELINE configs[2]: sp_cyp_diplotype on 2,000 targeted reads, real 39 templates / variant table
    cohort         BASELINE configs[4] shape per GPU: 32 WGS-style samples through sp_hla_diplotype_cohort
    cpu_baseline   the oracle (scalar C port of the same contract) on a bounded sample of the same reads: K1 + consensus + K2
`--gpus N` with N > 1: one process per GPU (spawned here when no launcher set WORLD_SIZE), each rank owns one synthetic sample (weak
scaling, no data-path collective); the per-gene calls are gathered with one RCCL all_gather -- the only exchange of the path.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

# HIP maps streams onto a few hardware queues (4 by default): the streams of the samples-in-flight leg should each get their own
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_WAVE_INSTR = 256 * 4 * 2.4e9 / 2    # 256 CU x 4 SIMD-32 x one wave-instruction per 2 cycles at 2.4 GHz (MI355X_MICROARCH.md); sp_microbench measures it


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks as child processes BEFORE this process touches the GPU (never
    re-exec a process that has), wait for them, fail if any of them fails.  Rank 0's stdout is the JSON line."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        code = p.wait()
        if code != 0 and rc == 0:
            rc = code
            for q in procs:                  # our own children, by handle
                if q.poll() is None:
                    q.kill()
    return rc


# ---------------------------------------------------------------------------------------------------------------- CPU baseline
_CB = {}


def _cpu_k1_worker(args):
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


def _cpu_gene_worker(args):
    """one worker process: the gene loop of one gene on the sample (dual consensus, group consensus, typing = K2) from oracle pieces"""
    g, sample = args
    import hla_expected as hx
    import hla_pipeline as hp
    from pb_starphase_amd import synth
    o, fx = _CB["o"], _CB["fx"]
    reads = [_CB["reads"][r] for r in sample]
    t0 = time.perf_counter()
    k1 = hx.k1_records_for(o, fx, reads, [_CB["best"][r] for r in sample], hx.K1Tables(o, fx, _CB["off"]))   # segments and offsets of the reads K1 placed
    t1 = time.perf_counter()
    res = hp.diplotype_gene(o, fx, g, reads, k1, synth)
    return g, t1 - t0, time.perf_counter() - t1, (res["allele1"], res["allele2"])


def native_oracle():
    """the oracle rebuilt on THIS host with -O3 -march=native (the CPU leg is compiled for the machine it runs on); the
    shipped liboracle.so (-O3, generic x86-64) is the fallback when no compiler is at hand"""
    src = os.path.join(ROOT, "oracle")
    out = os.path.join(src, "liboracle_native.so")
    flags = "-O3 -march=native -std=c11 -fPIC -ffp-contract=off -fno-fast-math"
    try:
        files = sorted(f for f in os.listdir(src) if f.endswith(".c"))
        subprocess.check_call(["gcc"] + flags.split() + ["-shared", "-o", out] + [os.path.join(src, f) for f in files] + ["-lm"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return out, flags
    except Exception:
        return None, "-O3 (shipped build)"


def cpu_baseline(fx, wl, budget_s=10.0):
    """The oracle ("port": the same contract in scalar C, one alignment per (read, allele) cell -- an EXHAUSTIVE search, not
    minimap2's seed-chain-extend with best_n = 5, which does orders of magnitude less base-level work per read and is not on disk)
    on a bounded sample of the same batch, BEFORE the GPU is touched (workers are forked): K1 over every host core, then per gene
    the dual + group consensus and the typing of the consensuses against every allele (K2).  The reference itself is
    single-threaded (src/cli/diplotype.rs:185-191): the one-thread K1 rate is given as well."""
    import ctypes as C
    import multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi
    lib, flags = native_oracle()
    o = oracle_ffi.load(lib) if lib else oracle_ffi.load()
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
    _CB.update(o=o, L=L, fx=fx, reads=wl.reads, refs=refs, n_all=n_all, enc=enc, off=off,
               ref_ptr=(C.c_void_p * len(refs))(*[r.ctypes.data for r in refs]), ref_len=np.array([len(r) for r in refs], np.int32),
               al_ptr=(C.c_void_p * n_all)(*[(e.ctypes.data if len(e) else None) for e in enc]),
               al_len=np.array([len(e) for e in enc], np.int32), gene_of=fx.gene_of.astype(np.int32))
    cores = max(1, min(len(os.sched_getaffinity(0)), 64, len(wl.reads) // 8))
    per = len(wl.reads) // cores                  # reads reserved per worker (not exhausted inside the budget at bench sizes)
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.map(_cpu_k1_worker, [(w * per, min(len(wl.reads), (w + 1) * per), budget_s) for w in range(cores)])
    t_k1 = max(r[1] for r in res)
    done = sum(r[0] for r in res)
    single = res[0][0] / res[0][1] if res[0][1] > 0 else 0.0
    best = dict(b for r in res for b in r[2])
    # the rest of the path on the reads K1 just placed: per gene, consensus + typing (two workers, one per gene)
    _CB["best"] = best
    sample = sorted(best)
    t1 = time.perf_counter()
    with mp.get_context("fork").Pool(len(fx.genes)) as pool:
        gres = pool.map(_cpu_gene_worker, [(g, sample) for g in range(len(fx.genes))])
    t_rest = time.perf_counter() - t1
    calls = {g: c for g, _a, _b, c in gres}
    wall = time.perf_counter() - t0
    return {"value": done / (t_k1 + t_rest), "unit": "reads/s", "cores": cores, "kind": "port", "compiler_flags": flags,
            "single_thread_k1_value": single, "k1_s": t_k1, "consensus_and_k2_s": t_rest,
            "sample": f"{done} reads of the same batch: K1 (anchor + every allele cell + acceptance) over {cores} forked workers in {t_k1:.1f} s, then per gene "
                      f"(2 workers) dual + group consensus and typing against every allele (K2) in {t_rest:.1f} s; {wall:.1f} s wall",
            "note": "exhaustive scalar port of the library's alignment contract, NOT minimap2 (absent): a reported baseline, not the >= 20x target"}, best, calls


# ---------------------------------------------------------------------------------------------------------------- legs
def cyp_leg(pkg, ctx, n_reads=2000, reps=2):
    """BASELINE configs[2]: sp_cyp_diplotype on the synthetic chr22 locus (database coordinates, 39 templates, real variant table)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cyp_cases_real as cr
    from pb_starphase_amd import synth
    cfg, gene_def = cr.load_db()
    locus = synth.Chr22Locus(cfg, gene_def, seed=3)
    db = pkg.ffi.CypDb(ctx, cfg, gene_def, locus.sequence, locus.start)
    out, sets = {}, []
    total_reads, total_s, ok = 0, 0.0, 0
    for name, haps, expected in cr.scenarios(locus):
        reads = locus.sample(np.random.default_rng(7), haps, n_reads)
        R = ctx.upload(reads)
        best = None
        for _ in range(reps):
            ctx.profile_reset()
            ctx.synchronize()
            t0 = time.perf_counter()
            call, _cons, _labels = db.diplotype(R)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        good = sorted([call.hap1.decode(), call.hap2.decode()]) == sorted(expected)
        ok += good
        out[name] = {"ms": 1e3 * best, "reads": len(reads), "call_equals_truth": bool(good), "cons_ms": ctx.profile_get("cons_steps")[0],
                     "launch_pairs": ctx.profile_get("cons_windows")[2], "cut_windows": ctx.profile_get("cons_cut_windows")[2],
                     "expansions": ctx.profile_get("cons_expansions")[2], "nodes_expanded": ctx.profile_get("cons_columns")[2],
                     "host_wall_ms": {k: round(ctx.profile_get("host:cyp_" + k)[0], 2) for k in ("regions", "segments", "consensus", "merge", "typing", "weights", "chains", "chain_pair")}}
        total_reads += len(reads); total_s += best
        sets.append(R)
    # the same six samples as one GPU's share of a cohort: one sp_cyp_diplotype_cohort call, samples spread over the context's streams
    cohort_best = None
    for _ in range(reps):
        ctx.synchronize()
        t0 = time.perf_counter()
        cohort = db.diplotype_cohort(sets)
        dt = time.perf_counter() - t0
        cohort_best = dt if cohort_best is None or dt < cohort_best else cohort_best
    cohort_ok = sum(sorted([c.hap1.decode(), c.hap2.decode()]) == sorted(exp) for (c, _cons, _rc), (_n, _h, exp) in zip(cohort, cr.scenarios(locus)))
    return {"value": total_reads / total_s, "unit": "reads/s",
            "cohort_call": {"value": total_reads / cohort_best, "unit": "reads/s", "samples_per_s": len(sets) / cohort_best, "ms": 1e3 * cohort_best,
                            "calls_equal_truth": f"{cohort_ok}/{len(sets)}", "workload": "the six samples in one sp_cyp_diplotype_cohort call"},
            "workload": f"BASELINE configs[2]: six scenarios x {n_reads} targeted-style reads (3-8 kb) on the synthetic chr22 "
            "locus, 39 templates, 393 variants / 520 star alleles of the bundled DB; sp_cyp_diplotype (K3 -> K8 -> K9/K7 -> K4 -> chains -> K5)",
            "calls_equal_truth": f"{ok}/{len(out)}", "scenarios": out}


def chain_pair_leg(pkg, ctx, n_d6=4, n_reads=1000, reps=2):
    """K5 at the scale of a duplication-rich sample: ~1.3k enumerated chains x 1,000 reads -> ~0.9 M chain pairs, each the f64
    likelihood of all reads under the pair (src/cyp2d6/chaining.rs:421-566).  Larger problems: profiles/r02/k5_scale.json."""
    from pb_starphase_amd import synth
    prob = synth.chain_pair_problem(n_d6, n_reads, np.random.default_rng(7))
    best = None
    for _ in range(reps):
        ctx.profile_reset(); ctx.synchronize()
        t0 = time.perf_counter()
        rc, res = ctx.cyp_best_chain_pair(**prob)
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    P = res.n_possible
    pairs = P * (P + 1) // 2
    ms_pairs, ms_tab = ctx.profile_get("k5_pairs")[0], ctx.profile_get("k5_chain_reads")[0]
    # per pair and read: two table entries (the read's best window total under either chain, u64, + the mask of the starts reaching it, u64)
    table_bytes = 2 * 16
    return {"value": pairs / (ms_pairs * 1e-3) if ms_pairs else None, "unit": "chain pairs/s", "status": rc, "chains": P, "reads": n_reads, "pairs": pairs,
            "pairs_scored": int(res.n_pairs_scored), "k5_pairs_ms": ms_pairs, "k5_chain_reads_ms": ms_tab, "wall_ms": 1e3 * best,
            "pair_read_terms_per_s": pairs * n_reads / (ms_pairs * 1e-3) if ms_pairs else None,
            "table_bytes_read_per_s": pairs * n_reads * table_bytes / (ms_pairs * 1e-3) if ms_pairs else None,
            "table_footprint_bytes": P * n_reads * 16,
            "workload": f"sp_cyp_best_chain_pair: {n_d6} CYP2D6 consensuses, {n_reads} reads, {P} enumerated chains (the per-(chain, read) tables sit in L2; "
                        "the kernel is f64-add / compare bound, not HBM bound)"}


def inflight_leg(pkg, fx, n_streams=3, steps=10, n_reads=10000, device=0):
    """Several samples in flight on one GPU: n_streams host threads, each with its own context (= HIP stream), database handle and
    resident 10,000-read sample, run the same reads -> diplotype step as the headline.  The VALU-bound K1 of one sample overlaps the
    latency-bound consensus launches of the others (ctypes releases the GIL inside the library)."""
    import threading
    from pb_starphase_amd import synth
    workers = []
    for t in range(n_streams):
        wl = synth.Config2Workload(fx, n_reads=n_reads, seed=2000 + t)
        c = pkg.Context(device)
        c.set_option("hla_split_genes", 0)          # the other samples' streams fill the gaps: one stream per sample
        d = fx.make_db(pkg, c)
        workers.append((c, d, c.upload(wl.reads), wl))
    genes = list(range(len(fx.genes)))
    same = lambda a, b: a == b or (a >= 0 and b >= 0 and fx.cdna[a] == fx.cdna[b] and fx.dna[a] == fx.dna[b])

    def run(w, k, out):
        c, d, reads, _wl = w
        for _ in range(k):
            o = d.realign_reads(reads)
            out[:] = [d.diplotype_genes(genes, reads, o)[0]]

    for w in workers:
        run(w, 1, [])
    outs = [[] for _ in workers]
    th = [threading.Thread(target=run, args=(workers[i], steps, outs[i])) for i in range(n_streams)]
    for c, _d, _r, _w in workers:
        c.synchronize()
    t0 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    for c, _d, _r, _w in workers:
        c.synchronize()
    dt = time.perf_counter() - t0
    ok = 0
    for (c, d, reads, wl), out in zip(workers, outs):
        truth = {g: sorted(a for (gg, _c, _d2, a) in wl.consensus if gg == g) for g in genes}
        for g, (call, _c1, _c2) in enumerate(out[0]):
            ok += all(same(a, b) for a, b in zip(sorted([call.allele1, call.allele2]), sorted((truth[g] * 2)[:2])))
    return {"value": n_reads * n_streams * steps / dt, "unit": "reads/s", "streams": n_streams, "samples": n_streams * steps, "ms_per_sample": 1e3 * dt / (n_streams * steps),
            "diplotypes_equal_truth": f"{ok}/{n_streams * len(genes)}",
            "workload": f"{n_streams} samples of {n_reads} reads in flight on {n_streams} HIP streams of one GPU, the headline's step each (one process, one host thread per stream)"}


def cohort_leg(pkg, ctx, fx, db, n_samples=32, reps=2, seed=5):
    """BASELINE configs[4] per GPU: n_samples WGS-style samples (~45 reads per gene) through one K1 call + sp_hla_diplotype_cohort"""
    from pb_starphase_amd import synth
    rng = np.random.default_rng(seed)
    reads, sample_of, truth = [], [], []
    for s in range(n_samples):
        t = {}
        for g in range(len(fx.genes)):
            pick = rng.choice(fx.full_length_alleles(g), 2, replace=False).tolist()
            t[g] = sorted(pick)
            for a in pick:
                hap, st = fx.haplotype(g, a)
                rs = synth.simulate_reads(rng, hap, st, len(fx.dna[a]), 22, mean_len=7000, sd_len=1500, min_overlap=2500)
                reads += rs; sample_of += [s] * len(rs)
        truth.append(t)
    R = ctx.upload(reads)
    genes = list(range(len(fx.genes)))
    same = lambda a, b: a == b or (a >= 0 and b >= 0 and fx.cdna[a] == fx.cdna[b] and fx.dna[a] == fx.dna[b])
    best = None
    for _ in range(reps + 1):
        ctx.synchronize()
        t0 = time.perf_counter()
        k1 = db.realign_reads(R)
        cohort, _ = db.diplotype_cohort(n_samples, sample_of, genes, R, k1)
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    ok = sum(all(same(x, y) for x, y in zip(sorted([cohort[s][g][0].allele1, cohort[s][g][0].allele2]), truth[s][g])) for s in range(n_samples) for g in genes)
    return {"value": len(reads) / best, "unit": "reads/s", "samples_per_s": n_samples / best, "ms": 1e3 * best, "samples": n_samples, "reads": len(reads),
            "workload": f"BASELINE configs[4] per GPU: {n_samples} WGS-style samples x HLA-A/-B (~44 reads per gene), one K1 call + sp_hla_diplotype_cohort",
            "calls_equal_truth": f"{ok}/{n_samples * len(genes)}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=10000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the scoring-only, CYP2D6 and cohort legs")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    pkg = ge.load_package()
    from pb_starphase_amd import synth, shard
    fx = synth.HlaFixture()
    wl = synth.Config2Workload(fx, n_reads=args.reads, seed=1000 + rank)
    cb, cpu_best, cpu_calls = (None, None, None)
    if not args.no_cpu_baseline and world == 1:
        cb, cpu_best, cpu_calls = cpu_baseline(fx, wl)          # forks workers: must happen before anything touches the GPU
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
    coll_dev = "cuda" if backend == "nccl" else "cpu"

    ctx = pkg.Context(device_index)
    db = fx.make_db(pkg, ctx)
    t_up = time.perf_counter()
    reads = ctx.upload(wl.reads)
    t_up = time.perf_counter() - t_up
    genes = list(range(len(fx.genes)))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    def max_over_ranks(dt):
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return dt

    # ---- headline: reads -> diplotype
    def step():
        o = db.realign_reads(reads)
        gene_calls = db.diplotype_genes(genes, reads, o)[0]
        if world > 1:
            # RCCL: the only exchange step of the path -- one gather of the per-(sample, gene) call records
            rec = np.zeros(len(genes), shard.CALL_DTYPE)
            for g, (c, _c1, _c2) in enumerate(gene_calls):
                rec[g] = (rank, g, c.allele1, c.allele2)
            shard.gather_calls(rec, device=coll_dev, same_count=True)
        return o, gene_calls

    for _ in range(args.warmup):
        out, gene_calls = step()
    ctx.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, gene_calls = step()
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)

    e2e_names = ("anchor", "anchor_k1", "anchor_k2", "anchor_type", "k1_cells", "k1_cells_deep", "k1_reduce", "k1_finalize", "hla_segments", "cons_steps", "type_consensus_ref",
                 "k2_cells_cdna", "k2_cells_dna", "k2_scan")
    kernel_ms = {k: ctx.profile_get(k)[0] / max(1, args.steps) for k in e2e_names}
    ms_cells, launches, cells_all = ctx.profile_get("k1_cells")
    executed, resumed, active = ctx.counter("k1_cells_executed"), ctx.counter("k1_cells_resumed"), ctx.counter("k1_cells_active")
    exec_bytes = ctx.counter("k1_cells_bytes")
    cons_windows = ctx.profile_get("cons_windows")[2]
    cons_cut = ctx.profile_get("cons_cut_windows")[2]
    cons_cols = ctx.profile_get("cons_columns")[2]
    cons_exp = ctx.profile_get("cons_expansions")[2]
    host_ms = {k: ctx.profile_get("host:" + k)[0] / max(1, args.steps) for k in ("hla_select", "hla_segments", "hla_setup", "hla_dual_hpc", "hla_dual_dna", "hla_groups", "hla_typing",
                                                                                  "k8_prologue", "k8_loop", "k8_result_wait", "k8_epilogue", "k1_total", "k1_result",
                                                                                  "hla_genes_total", "hla_split_spawn", "hla_split_own", "hla_split_join")}
    cons_ticks = {k: ctx.profile_get("cons_ticks_" + k)[2] / 100.0 / max(1, args.steps) for k in ("reduce", "result", "search", "tail")}   # 100 MHz -> us
    avg_ms = ms_cells / max(1, launches)
    per_launch = lambda v: v / max(1, launches)
    achieved = per_launch(exec_bytes) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0

    truth = {g: sorted(a for (gg, _c, _d, a) in wl.consensus if gg == g) for g in genes}
    same = lambda a, b: a == b or (a >= 0 and b >= 0 and fx.cdna[a] == fx.cdna[b] and fx.dna[a] == fx.dna[b])
    ok = 0
    for g, (c, _c1, _c2) in enumerate(gene_calls):
        got, want = sorted([c.allele1, c.allele2]), (truth[g] * 2)[:2]
        ok += all(same(a, b) for a, b in zip(got, sorted(want)))
    k1_gene_ok = float(np.mean([out[r]["gene"] == wl.read_truth[r][0] for r in range(len(wl.reads))]))
    k1_realigned = float(np.mean(out["status"] == 0))

    # ---- scoring only (round 1's step): K1 + K2 on the truth consensuses
    scoring = None
    if not args.no_extra_legs:
        def score_step():
            o = db.realign_reads(reads)
            return o, [b for b, _n in db.score_consensus_batch([(g, cons_dna, cons_cdna) for (g, cons_dna, cons_cdna, _a) in wl.consensus])]
        score_step()
        barrier()
        t1 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            _o, calls = score_step()
        barrier()
        dts = max_over_ranks(time.perf_counter() - t1)
        k2_ok = sum(1 for (g, _c, _d, a), b in zip(wl.consensus, calls) if same(a, b))
        scoring = {"value": args.reads * world * reps / dts, "unit": "reads/s", "ms_per_step": 1e3 * dts / reps,
                   "workload": "K1 on the same reads + K2 on 4 truth consensuses (no consensus step): the read -> allele and consensus -> allele scoring alone",
                   "k2_truth_calls": f"{k2_ok}/{len(calls)}"}

    peaks, cyp, cohort, k5, inflight = None, None, None, None, None
    if rank == 0:
        peaks = {"valu_int_wave_instr_per_s": ctx.microbench("valu_int"), "match16_valu_wave_instr_per_s": ctx.microbench("match16"),
                 "hbm_copy_bytes_per_s": ctx.microbench("hbm_copy")}
    if rank == 0 and world == 1 and not args.no_extra_legs:
        cyp = cyp_leg(pkg, ctx)
        cohort = cohort_leg(pkg, ctx, fx, db)
        k5 = chain_pair_leg(pkg, ctx)
        inflight = inflight_leg(pkg, fx, device=device_index)

    if rank == 0:
        # VALU wave-instructions of one k1_cells launch: rocprofv3 --pmc SQ_INSTS_VALU of this workload (profiles/r02/valu_k1_cells.json);
        # the peak is measured in this run (sp_microbench: eight independent v_add_u32 chains per lane on every SIMD)
        import hashlib
        k1_sha = hashlib.sha256(b"".join(open(os.path.join(ROOT, "pb-starphase_amd", "csrc", f), "rb").read() for f in ("sp_hla.hip", "sp_wfa.cuh"))).hexdigest()[:16]
        stale = lambda rec: "" if rec.get("k1_source_sha16") == k1_sha else "STALE: the kernel sources changed after this counter pass; "
        valu = None
        vfile = os.path.join(ROOT, "profiles", "r02", "valu_k1_cells.json")
        if os.path.exists(vfile) and args.reads == 10000 and avg_ms > 0:
            rec = json.load(open(vfile))
            rate = rec["sq_insts_valu_per_launch"] / (avg_ms * 1e-3)
            valu = {"bound": "valu", "kernel": "k1_cells_kernel", "achieved": rate, "peak": peaks["valu_int_wave_instr_per_s"], "unit": "wave-instr/s",
                    "frac": rate / peaks["valu_int_wave_instr_per_s"], "nominal_peak": VALU_PEAK_WAVE_INSTR,
                    "frac_of_match16_mix_peak": rate / peaks["match16_valu_wave_instr_per_s"],
                    "note": stale(rec) + "instruction count from the committed PMC pass (" + rec["method"] + "), launch time and peak measured in this run"}
        traffic, traffic_note = None, None
        tfile = os.path.join(ROOT, "profiles", "r02", "traffic_k1_cells.json")
        if os.path.exists(tfile) and args.reads == 10000:
            rec = json.load(open(tfile))
            traffic, traffic_note = rec["hbm_bytes_per_launch"], stale(rec) + "from the committed rocprofv3 PMC passes of this workload (" + rec["method"] + ")"
        total_reads = args.reads * world * args.steps
        line = {
            "metric": "HiFi reads/sec diplotyped (HLA-A + HLA-B, reads -> diplotype: realignment, dual + group consensus, typing, het/hom call)",
            "value": total_reads / dt, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8 (2-bit packed bases, int32 wavefront DP, f64 score ratios)", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: HLA-A/-B, %d synthetic HiFi reads/GPU vs bundled IMGT/HLA DB v0.14.1 "
                                   "(18,461 alleles, 11,199 with DNA), reads -> diplotype" % args.reads,
                       "reads_per_gpu": args.reads, "alleles": len(fx.ids), "parallelism": "one sample per GPU, RCCL all_gather of calls"},
            "roofline": {"bound": "hbm", "kernel": "k1_cells_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                         "algorithmic_bytes_per_launch": per_launch(exec_bytes), "cells_executed_per_launch": per_launch(executed),
                         "cells_resumed_per_launch": per_launch(resumed), "cells_active_per_launch": per_launch(active),
                         "cells_settled_without_running_per_launch": per_launch(active - executed), "avg_launch_ms": avg_ms,
                         "measured_hbm_copy_GBs": peaks["hbm_copy_bytes_per_s"] / 1e9,
                         "note": "'achieved' = algorithmic bytes (SURVEY 8(d): ceil(Lq/4) + ceil(Lt/4) + 32 per cell) of the cells the launch EXECUTED, counted on the "
                                 "device, / launch time (HIP events).  The kernel is an integer-DP kernel whose database sits in L2: its limiter is VALU issue "
                                 "(roofline_valu), 'traffic' is what actually crossed HBM"},
            "roofline_valu": valu,
            "kernel_ms": kernel_ms, "host_wall_ms": host_ms,
            **({"dbg_counters": [ctx.counter(f"dbg{i}") for i in range(8)]} if os.environ.get("SP_BENCH_DBG") else {}),
            "anchor_pairs_per_step": {k: ctx.profile_get(k)[2] / max(1, args.steps) for k in ("anchor_k1", "anchor_k2", "anchor_type")},
            "anchor_launches_per_step": {k: ctx.profile_get(k)[1] / max(1, args.steps) for k in ("anchor_k1", "anchor_k2", "anchor_type")},
            "consensus": {"windows_per_step": cons_windows / max(1, args.steps), "launches_per_step": 3 * cons_windows / max(1, args.steps),
                          "cut_windows_per_step": cons_cut / max(1, args.steps), "expansions_per_step": cons_exp / max(1, args.steps),
                          "nodes_expanded_per_step": cons_cols / max(1, args.steps),
                          "control_kernel_us_per_step": cons_ticks},
            "concordance": {"k1_gene_correct": k1_gene_ok, "k1_realigned": k1_realigned, "diplotypes_equal_truth": f"{ok}/{len(genes)} genes"},
            "pcie_inclusive_upload_s": t_up,
            "scoring_only": scoring, "samples_in_flight": inflight, "cyp2d6": cyp, "cohort": cohort, "k5_chain_pairs": k5,
        }
        if cb is not None:
            agree = sum(1 for i, b in cpu_best.items() if b == int(out[i]["best_allele"]))
            cb["k1_calls_identical_to_gpu"] = f"{agree}/{len(cpu_best)}"
            cb["diplotypes_of_the_sample"] = {fx.genes[g]: [int(x) for x in c] for g, c in cpu_calls.items()}
            line["cpu_baseline"] = cb
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
