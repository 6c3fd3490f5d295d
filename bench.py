#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric "HiFi reads/sec diplotyped (HLA+CYP2D6)" on one node of MI355X.

N = 1, headline (`value`): ONE synthetic sample carrying both loci of the metric -- BASELINE configs[1] (HLA-A / -B, 10,000 HiFi reads against the
bundled IMGT/HLA database) and configs[2] (CYP2D6, 2,000 targeted reads) -- from the reads' bytes in host memory to the diplotypes, a NEW sample
every step:
    upload   sp_seqset_upload_async: BAM's 4-bit SEQ bytes through the pinned staging ring, the bytes of sample i + 1 under the kernels of sample i
    HLA      sp_hla_realign_reads (K1: anchors, every read x every DNA allele) -> sp_hla_diplotype_genes (segments, HPC dual + group consensus = K8,
             typing of the consensuses against every allele = K2, het / hom call)
    CYP2D6   sp_cyp_diplotype (K3 regions -> multi-way consensus K8 -> K9 / K7 typing -> K4 weights -> chains -> K5 chain pair) on a context of its own,
             beside the HLA half on the same GPU (two host threads, two HIP streams + their helpers)
Beside it in the same JSON line: `roofline` (k1_cells_kernel), `cpu_baseline` (the reference's call pattern on the minimap2
restatement of the oracle, see tests/cpu_port_seeded.py), `legs` (HLA alone with resident reads = round 2's headline, the six CYP2D6 scenarios, the
256-sample cohort on this one GPU, K5 at scale).

N > 1 (`--gpus N`, or under torch.distributed.run), one process per GPU: every rank runs the same stream of samples (its own samples) -- per-GPU work
fixed, nothing exchanged, `value` = all ranks' reads / the slowest rank's time between the barriers (weak scaling by independent samples).
`--workload cohort` asks the other question at any N: BASELINE configs[4], the 256-sample cohort sharded by sample over the ranks -- every rank uploads
its samples' reads, runs sp_hla_diplotype_cohort + sp_cyp_diplotype_cohort + sp_variant_solve_batch and the per-(sample, gene) call records are gathered
with ONE sp_gather_results (ncclAllGather over RCCL / xGMI) per step, the only exchange of the path; the total work is fixed (strong scaling), and the
N = 1 line carries the same cohort as `legs.cohort`.
"""
import argparse
import gzip
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

# HIP maps streams onto a few hardware queues (4 by default), in the order the streams are made: the two halves of a sample, their helper and copy streams should
# each get their own (with 8, one more idle helper stream moved the CYP2D6 chain of a later context onto the queue of another sample's K1: 300k -> 230k reads/s
# with three samples in flight; 16 and 24 measure the same)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_WAVE_INSTR = 256 * 4 * 2.4e9 / 2    # 256 CU x 4 SIMD-32 x one wave-instruction per 2 cycles at 2.4 GHz (MI355X_MICROARCH.md); sp_microbench measures it
COUNTER_DIR = os.path.join(ROOT, "profiles", "r06")


def cyp_persistent():
    """what the legs' contexts that run CYP2D6 consensus chains set "k8_persistent" to: 2 = the library's own choice (its default: persistent kernels for a single sample's
    batches when the process's streams have hardware queues of their own, a launch pair per step for the wide batches of a cohort call); SP_BENCH_CYP_PERSISTENT=0 / 1 forces
    launch pairs / persistent kernels"""
    return int(os.environ.get("SP_BENCH_CYP_PERSISTENT", "2"))


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


def one_socket_cpus():
    """the logical CPUs of ONE socket (north_star: "single-socket CPU"): those of this process's affinity set that sit in the package of the first of them"""
    mine = sorted(os.sched_getaffinity(0))
    pkg_of = {}
    for c in mine:
        try:
            pkg_of[c] = int(open(f"/sys/devices/system/cpu/cpu{c}/topology/physical_package_id").read())
        except Exception:
            pkg_of[c] = 0
    first = pkg_of[mine[0]]
    return [c for c in mine if pkg_of[c] == first], len(set(pkg_of.values()))


def host_description():
    """CPU model / sockets / cores of the box the CPU leg runs on"""
    out = {"logical_cpus": len(os.sched_getaffinity(0))}
    try:
        for line in subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout.splitlines():
            k, _, v = line.partition(":")
            if k.strip() in ("Model name", "Socket(s)", "Core(s) per socket", "Thread(s) per core", "CPU max MHz"):
                out[k.strip()] = v.strip()
    except Exception:
        pass
    return out



# ---------------------------------------------------------------------------------------------------------------- the printed line
LINE_LIMIT = 4000          # bytes of the ONE JSON line on stdout (BENCH_r05.json: a 22 KB line came back unparsed); everything else goes to the side file


def _r(x, digits=4):
    """a float to `digits` significant digits (None stays None)"""
    if isinstance(x, float):
        return float(("%." + str(digits) + "g") % x)
    return x


def emit(full, compact, path):
    """write the whole record to `path` (bench_full.json: legs, critical path, per-scenario CPU blocks) and print the compact line -- the contract's keys, the roofline and
    cpu_baseline blocks, a few summary numbers -- as the only line on stdout.  Optional blocks are dropped, last first, should the line ever pass LINE_LIMIT"""
    try:
        with open(path, "w") as f:
            json.dump(full, f)
        compact["full"] = os.path.relpath(path, ROOT) if path.startswith(ROOT) else path
    except OSError as e:
        compact["full"] = "not written: %s" % e
    optional = ["summary", "concordance", "cohort"]
    text = json.dumps(compact)
    while len(text) >= LINE_LIMIT and optional:
        compact.pop(optional.pop(0), None)
        text = json.dumps(compact)
    if len(text) >= LINE_LIMIT:
        raise SystemExit("bench.py: the line's required blocks alone are %d bytes" % len(text))
    if os.environ.get("SP_BENCH_FULL_STDERR") == "1":
        sys.stderr.write(json.dumps(full) + "\n")
    print(text, flush=True)



# ---------------------------------------------------------------------------------------------------------------- the ranks' group
ABANDONED = []             # what a group that was given up leaves behind: kept alive and never touched again (no destroy call that could wait for the same peers)


def bring_up_group(pkg, shard, backend, coll_dev, device_index, rank, world):
    """sp_group: RCCL through the library (sp_gather_results).  The communicator is made AND tried -- one small gather whose answer is known -- before anything is timed, on a
    helper thread with a time limit and on a CONTEXT OF ITS OWN (its own stream: a collective that never completes sits there, not on a stream the run needs; the helper never
    touches the run's context, which is not thread safe).  The id travels through the rendezvous store, not through a collective of the process group, so a rank that hangs or
    fails anywhere on the way leaves nothing pending on that group; the ranks then agree with ONE all-reduce -- the first collective every rank issues -- and fall back to the
    torch process group together.  What was given up is never touched again (ABANDONED); a run that gave up a thread still inside the library leaves through os._exit.
    -> (group, which path gathers, why the library's group was given up or None)"""
    import torch
    import torch.distributed as dist
    box = {}
    inject = os.environ.get("SP_BENCH_INJECT_GROUP_HANG", "")     # tests: "1" = every rank hangs; "<rank>:before" / "<rank>:after" / "<rank>:error" = that rank only, before / after its group is made

    def injected(where):
        if inject == "1":
            return where == "before"
        if ":" in inject:
            r, w = inject.split(":")
            return int(r) == rank and w == where
        return False

    def bring_up():
        try:
            torch.cuda.set_device(device_index)           # (the current device is a per-thread setting)
            if injected("before"):
                time.sleep(3600)
            if injected("error"):
                raise RuntimeError("injected error")
            gctx = pkg.Context(device_index) if backend == "nccl" else None
            box["gctx"] = gctx
            g = shard.make_group(gctx, pkg.ffi, backend=backend, device=coll_dev)
            box["made"] = g
            if injected("after"):
                time.sleep(3600)
            if not isinstance(g, shard.TorchGroup):       # (the torch group is the process group itself: nothing to try, and no collective of it belongs on a helper thread)
                probe = np.zeros(2, shard.CALL_DTYPE); probe["sample"] = rank; probe["gene"] = [0, 1]; probe["allele1"] = 1000 + rank
                got = g.gather(probe)
                want = np.stack([np.array([(r, 0, 1000 + r, 0), (r, 1, 1000 + r, 0)], shard.CALL_DTYPE) for r in range(world)])
                if got.shape != want.shape or not (got == want).all():
                    raise RuntimeError("the first gather did not return every rank's records")
            box["group"] = g
        except Exception as e:
            box["error"] = e
    th = threading.Thread(target=bring_up, daemon=True)
    th.start()
    th.join(timeout=float(os.environ.get("SP_BENCH_GROUP_TIMEOUT_S", "180")))
    timed_out = th.is_alive()
    err = TimeoutError("the communicator did not come up within the time limit") if timed_out else box.get("error")
    ok_here = 0 if err is not None else 1
    if err is not None:
        print(f"rank {rank}: the library's group failed ({err}); gathering through torch.distributed instead", file=sys.stderr, flush=True)
    flag = torch.tensor([ok_here], dtype=torch.int32, device=coll_dev if backend == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 1:
        g = box["group"]
        return g, ("torch.distributed" if isinstance(g, shard.TorchGroup) else "sp_gather_results (librccl)"), None
    # given up on every rank together.  This rank's own group may be fine, half made, or still inside the library on the helper thread: it is left alone either way
    ABANDONED.append((box, th))
    if timed_out:
        ABANDONED.append("exit without teardown")           # (main() leaves through os._exit: a thread inside ncclCommInitRank / a gather on a dead peer would hold the interpreter's exit)
    return shard.TorchGroup(coll_dev), "torch.distributed", str(err if err is not None else "another rank's group failed")


# ---------------------------------------------------------------------------------------------------------------- CPU baseline
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


def cpu_baseline(fx, hla_reads, cyp_setup, cyp_sets, n_hla=None, n_cyp=None):
    """The reference's CPU path in its own call pattern for BOTH loci of the headline's mix (tests/cpu_port_seeded.py, tests/cpu_port_cyp.py): every alignment is the minimap2
    restatement's (oracle/mm2.c: seeded maps, `best_n 5`, two-piece affine gaps), the consensus is oracle/consensus.c, typing and chains are the oracle's routines --
    on a bounded share of the reads `value` is measured on (the first n_hla HLA reads of sample 0, the first n_cyp reads of EVERY CYP2D6 scenario of the mix), on the
    logical CPUs of ONE socket (north_star: single-socket CPU), BEFORE the GPU is touched (the workers are forked).
    -> (the cpu_baseline block, what the GPU has to reproduce: reads used, calls)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi
    import cpu_port_seeded
    import cpu_port_cyp
    all_cpus = os.sched_getaffinity(0)
    socket_cpus, n_sockets = one_socket_cpus()
    os.sched_setaffinity(0, socket_cpus)
    try:
        lib, flags = native_oracle()
        o = oracle_ffi.load(lib) if lib else oracle_ffi.load()
        n_hla = len(hla_reads) if not n_hla else min(n_hla, len(hla_reads))
        hres, best, calls, _cons, done = cpu_port_seeded.run(o, fx, hla_reads[:n_hla], n_sample=n_hla, budget_s=1e9, cores=len(socket_cpus))
        cfg, gene_def, locus = cyp_setup
        cdb, ccfg = cpu_port_cyp.tables(cfg, gene_def, locus)
        cyp_blocks, cyp_refs = {}, []
        cyp_wall = cyp_one = 0.0
        n_cyp_total = 0
        for name, reads in cyp_sets:
            creads = reads if not n_cyp else reads[:n_cyp]
            cres, ctm = cpu_port_cyp.run(o, cdb, ccfg, creads)
            cyp_wall += ctm["wall_s"]; cyp_one += ctm["one_thread_s"]; n_cyp_total += len(creads)
            cyp_blocks[name] = {"reads": len(creads), "wall_s": ctm["wall_s"], "one_thread_s": ctm["one_thread_s"], "cores": ctm["cores"],
                                "cpu_s": {"regions": ctm["regions_cpu_s"], "weights": ctm.get("weights_cpu_s", 0.0), "consensus_typing_chains": ctm["rest_wall_s"]},
                                "call": [cres.get("hap1", ""), cres.get("hap2", "")], "status": int(cres["status"])}
            cyp_refs.append((name, creads, cres))
    finally:
        os.sched_setaffinity(0, all_cpus)
    n = len(done) + n_cyp_total
    wall = hres["wall_s"] + cyp_wall
    one = hres["one_thread_s"] + cyp_one
    phys = set()
    for c in socket_cpus:
        try:
            phys.add(open(f"/sys/devices/system/cpu/cpu{c}/topology/core_id").read().strip())
        except Exception:
            phys.add(str(c))
    block = {"value": n / wall, "unit": "reads/s", "cores": len(socket_cpus), "physical_cores": len(phys), "kind": "port",
             "sample_short": f"a SAMPLE of the mix `value` runs on: the first {len(done)} of sample 0's {len(hla_reads)} HLA reads + the first {n_cyp or 'all'} reads of each of the {len(cyp_sets)} "
                             f"CYP2D6 scenarios ({n_cyp_total}); {len(socket_cpus)} logical CPUs ({len(phys)} cores) of one socket of {n_sockets}; the reference's call pattern on oracle/mm2.c + oracle/consensus.c",
             "sample": f"a bounded share of the mix `value` is measured on: the first {len(done)} of sample 0's {len(hla_reads)} HLA reads and the first {n_cyp or 'all'} reads of each of the "
                       f"{len(cyp_sets)} CYP2D6 scenarios the steps cycle through ({n_cyp_total} reads), one locus and one scenario after the other, each with its parallel stages over the "
                       f"{len(socket_cpus)} logical CPUs of one socket (of {n_sockets}) and its sequential stages (consensus, chains) on one",
             "single_thread_value": n / one, "wall_s": wall, "one_thread_s": one,
             "hla": hres, "cyp2d6": {"reads": n_cyp_total, "value": n_cyp_total / cyp_wall, "unit": "reads/s", "wall_s": cyp_wall, "one_thread_s": cyp_one, "scenarios": cyp_blocks,
                                     "workers_per_stage": {"find_base_type_in_sequence (39 template maps per read)": len(socket_cpus), "weight_sequence (per region segment)": len(socket_cpus),
                                                           "multi-way consensus, typing, chains, chain pair": 1}},
             "compiler_flags": flags, "host": host_description(), "affinity": f"{len(socket_cpus)} logical CPUs of socket 0 (sched_setaffinity), {n_sockets} socket(s) on the box",
             "note": "kind 'port': the reference's call pattern on minimap2's published algorithm restated in scalar C (oracle/mm2.c; minimap2 itself and its SSE kernels are "
                     "not on disk), waffle_con = oracle/consensus.c.  `cores` = the logical CPUs of the one socket the leg is pinned to; the reference itself is single-threaded "
                     "(src/cli/diplotype.rs:185-191): single_thread_value is what one thread needs for the same reads"}
    return block, {"hla_best": best, "hla_calls": calls, "hla_done": done, "hla_records": dict(cpu_port_seeded.G.get("records", {})), "cyp": cyp_refs}


# ---------------------------------------------------------------------------------------------------------------- workloads
class HlaSample:
    """BASELINE configs[1]: the reads of one sample as a BAM reader hands them over (sp_bam_last_seq4: 4 bits per base)"""

    def __init__(self, pkg, fx, n_reads, seed):
        from pb_starphase_amd import synth
        self.wl = synth.Config2Workload(fx, n_reads=n_reads, seed=seed)
        self.n = len(self.wl.reads)
        self.payload = pkg.ffi.encode_bam4(self.wl.reads)
        self.ascii_bytes = sum(len(r) for r in self.wl.reads)
        genes = range(len(fx.genes))
        self.truth = {g: sorted(a for (gg, _c, _d, a) in self.wl.consensus if gg == g) for g in genes}


class CypSample:
    """BASELINE configs[2]: 2,000 targeted-style reads of one scenario on the synthetic chr22 locus"""

    def __init__(self, pkg, locus, scenario, n_reads, seed):
        self.name, haps, self.expected = scenario
        self.reads = locus.sample(np.random.default_rng(seed), haps, n_reads)
        self.n = len(self.reads)
        self.payload = pkg.ffi.encode_bam4(self.reads)
        self.ascii_bytes = sum(len(r) for r in self.reads)


def same_allele(fx, a, b):
    return a == b or (a >= 0 and b >= 0 and fx.cdna[a] == fx.cdna[b] and fx.dna[a] == fx.dna[b])


class Lane(threading.Thread):
    """one locus of the stream of samples on a context of its own: wait for this sample's bytes, start the next sample's upload, compute"""

    def __init__(self, pkg, ctx, payloads, work, steps, fresh_upload=True, tickets=None):
        super().__init__()
        self.pkg, self.ctx, self.payloads, self.work, self.steps, self.fresh = pkg, ctx, payloads, work, steps, fresh_upload
        self.result, self.error = None, None
        # tickets: several lanes of one locus share the stream of samples -- a lane takes the next sample's number when it starts that sample's upload
        self.tickets = tickets if tickets is not None else Tickets(steps)
        self.pending = self.start_upload(self.tickets.take()) if fresh_upload else None
        self.resident = None if fresh_upload else [ctx.upload_format(pkg.ffi.SP_SEQ_BAM4, *p) for p in payloads]

    def start_upload(self, i):
        if i is None:
            return None
        blob, offs, lens = self.payloads[i % len(self.payloads)]
        return (i, self.ctx.upload_format(self.pkg.ffi.SP_SEQ_BAM4, blob, offs, lens, wait=False))

    def wait_first(self):
        if self.pending is not None:
            self.pending[1].wait()

    def run(self):
        try:
            while True:
                if self.fresh:
                    if self.pending is None:
                        break
                    t_u = time.perf_counter()
                    i, up = self.pending
                    cur = up.wait()
                    t_v = time.perf_counter()
                    self.pending = self.start_upload(self.tickets.take())    # the next sample's bytes travel under this sample's kernels
                    self.t_wait = getattr(self, "t_wait", 0.0) + (t_v - t_u); self.t_start = getattr(self, "t_start", 0.0) + (time.perf_counter() - t_v)
                else:
                    i = self.tickets.take()
                    if i is None:
                        break
                    cur = self.resident[i % len(self.resident)]
                t_w = time.perf_counter()
                self.result = self.work(cur, i)
                t_c = time.perf_counter()
                if self.fresh:
                    cur.close()
                self.t_work = getattr(self, "t_work", 0.0) + (t_c - t_w); self.t_close = getattr(self, "t_close", 0.0) + (time.perf_counter() - t_c)
        except Exception as e:                                      # surfaces in the main thread
            self.error = e


class Tickets:
    """the numbers 0 .. steps - 1, handed out once each to the lanes that share them"""

    def __init__(self, steps):
        self.n, self.next, self.lock = steps, 0, threading.Lock()

    def take(self):
        with self.lock:
            if self.next >= self.n:
                return None
            self.next += 1
            return self.next - 1


class ContextGroup:
    """the contexts of the lanes of one locus, read as one: option settings go to all of them, profile records add up"""

    def __init__(self, ctxs):
        self.ctxs = list(ctxs)

    def set_option(self, name, value):
        for c in self.ctxs:
            c.set_option(name, value)

    def profile_reset(self):
        for c in self.ctxs:
            c.profile_reset()

    def synchronize(self):
        for c in self.ctxs:
            c.synchronize()

    def profile_get(self, name):
        rows = [c.profile_get(name) for c in self.ctxs]
        return tuple(sum(r[k] for r in rows) for k in range(3))


def run_lanes(lanes):
    for x in lanes:
        x.start()
    for x in lanes:
        x.join()
    for x in lanes:
        if x.error is not None:
            raise x.error


# ---------------------------------------------------------------------------------------------------------------- legs
def cyp_leg(pkg, ctx, cdb, locus, n_reads=2000, reps=2):
    """BASELINE configs[2]: sp_cyp_diplotype on the six scenarios of the survey (database coordinates, 39 templates, real variant table)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cyp_cases_real as cr
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
            call, _cons, _labels = cdb.diplotype(R)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        good = sorted([call.hap1.decode(), call.hap2.decode()]) == sorted(expected)
        ok += good
        out[name] = {"ms": 1e3 * best, "reads": len(reads), "call_equals_truth": bool(good), "cons_ms": ctx.profile_get("cons_steps")[0],
                     "launch_triples": ctx.profile_get("cons_windows")[2], "cut_windows": ctx.profile_get("cons_cut_windows")[2],
                     "expansions": ctx.profile_get("cons_expansions")[2], "nodes_expanded": ctx.profile_get("cons_columns")[2],
                     "host_wall_ms": {k: round(ctx.profile_get("host:cyp_" + k)[0], 2) for k in ("regions", "segments", "consensus", "merge", "typing", "weights", "chains", "chain_pair")}}
        total_reads += len(reads); total_s += best
        sets.append(R)
    return {"value": total_reads / total_s, "unit": "reads/s",
            "workload": f"BASELINE configs[2]: six scenarios x {n_reads} targeted-style reads (3-8 kb) on the synthetic chr22 "
            "locus, 39 templates, 393 variants / 520 star alleles of the bundled DB; sp_cyp_diplotype one sample at a time, reads resident",
            "calls_equal_truth": f"{ok}/{len(out)}", "scenarios": out}


def chain_pair_leg(pkg, ctx, n_d6=4, n_reads=1000, reps=2):
    """K5 at the scale of a duplication-rich sample: ~1.3k enumerated chains x 1,000 reads -> ~0.9 M chain pairs, each the f64
    likelihood of all reads under the pair (src/cyp2d6/chaining.rs:421-566)."""
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
    return {"value": pairs / (ms_pairs * 1e-3) if ms_pairs else None, "unit": "chain pairs/s", "status": rc, "chains": P, "reads": n_reads, "pairs": pairs,
            "pairs_scored": int(res.n_pairs_scored), "k5_pairs_ms": ms_pairs, "k5_chain_reads_ms": ms_tab, "wall_ms": 1e3 * best,
            "workload": f"sp_cyp_best_chain_pair: {n_d6} CYP2D6 consensuses, {n_reads} reads, {P} enumerated chains (the per-(chain, read) tables sit in L2; "
                        "the kernel is f64-add / compare bound, not HBM bound)"}


# ---------------------------------------------------------------------------------------------------------------- the cohort (configs[4])
class VariantPanel:
    """the variant genes of the bundled database as K6 problems (sp_variant_gene_problem builds the haplotype side); the observed side of a
    synthetic sample is written straight into the integer arrays the kernel reads"""
    _shared = None

    @classmethod
    def shared(cls, pkg):
        if cls._shared is None:
            cls._shared = cls(pkg)
        return cls._shared

    def __init__(self, pkg):
        D = pkg.database
        path = os.path.join(GOLDEN, "gene_entries_v0.14.1.json.gz")
        names = sorted(json.load(gzip.open(path))["gene_entries"])
        self.db = D.Database(path)
        self.genes = []
        for name in names:
            gene = self.db.variant_gene(name)
            base = gene.problem()
            arr = D.problem_arrays(base)
            slots = []
            for h in range(arr["n_haps"]):
                sl = []
                for s in range(arr["slot_off"][h], arr["slot_off"][h + 1]):
                    alts = arr["alt_var"][arr["alt_off"][s]:arr["alt_off"][s + 1]]
                    sl.append([int(v) for v in alts])
                slots.append(sl)
            usable = [h for h in range(arr["n_haps"]) if not arr["hap_is_sv"][h] and all(any(v >= 0 for v in s) for s in slots[h])]
            keep = {k: np.ascontiguousarray(arr[k], t) for k, t in (("hap_is_sv", np.uint8), ("hap_is_core", np.uint8), ("slot_off", np.int32), ("alt_off", np.int32),
                                                                     ("alt_var", np.int32), ("var_is_core", np.uint8))}
            self.genes.append(dict(name=name, gene=gene, n_haps=arr["n_haps"], n_vars=arr["n_vars"], slots=slots, usable=usable, arrays=keep))

    def problems(self, pkg, rng, max_hets=8):
        """one synthetic sample: per gene a diplotype of two defined haplotypes and the variants a VCF would show for it"""
        out, keep, truth = [], [], []
        for G in self.genes:
            for _ in range(50):
                h1, h2 = (G["usable"][int(i)] for i in rng.integers(0, len(G["usable"]), 2))
                v1 = {next(v for v in s if v >= 0) for s in G["slots"][h1]}
                v2 = {next(v for v in s if v >= 0) for s in G["slots"][h2]}
                hets = sorted((v1 | v2) - (v1 & v2))
                if len(hets) <= max_hets:
                    break
            obs = {v: (4, -1) for v in v1 & v2}
            phased, ps = rng.random() < 0.6, int(rng.integers(1000, 2000))
            for v in hets:
                obs[v] = ((2 if v in v2 else 3), ps) if (phased and rng.random() < 0.85) else (1, -1)
            order = sorted(obs)
            a = G["arrays"]
            ov, og = np.array(order, np.int32), np.array([obs[v][0] for v in order], np.int32)
            op, ol = np.array([obs[v][1] for v in order], np.int64), np.full(len(order), -1, np.int32)
            p = pkg.ffi.sp_variant_problem()
            p.n_haps, p.n_vars, p.n_obs = G["n_haps"], G["n_vars"], len(order)
            for k in ("hap_is_sv", "hap_is_core", "slot_off", "alt_off", "alt_var", "var_is_core"):
                setattr(p, k, a[k].ctypes.data)
            p.obs_var, p.obs_gt, p.obs_ps, p.obs_sv_label = ov.ctypes.data, og.ctypes.data, op.ctypes.data, ol.ctypes.data
            out.append(p); keep.append((ov, og, op, ol)); truth.append((h1, h2))
        return out, keep, truth


class CohortShare:
    """a rank's share of BASELINE configs[4]: 256 WGS-style samples, all supported genes -- HLA-A / -B (~44 reads per gene), CYP2D6 (~100 reads),
    the 18 variant genes of the bundled database (synthetic VCF observations).  A sample's data depends on its global id only."""

    CACHE = {}                                   # sample id -> its data (a sample's data depends on its id only: the shares of every size are cut from the same samples)

    @classmethod
    def sample_data(cls, pkg, fx, locus, scen, panel, s):
        from pb_starphase_amd import synth
        if s not in cls.CACHE:
            rng = np.random.default_rng(10_000 + s)
            reads, truth = [], {}
            for g in range(len(fx.genes)):
                pick = sorted(rng.choice(fx.full_length_alleles(g), 2, replace=False).tolist())
                truth[g] = pick
                for a in pick:
                    hap, st = fx.haplotype(g, a)
                    reads += synth.simulate_reads(rng, hap, st, len(fx.dna[a]), 22, mean_len=7000, sd_len=1500, min_overlap=2500)
            sc = scen[s % 3]                                                       # *1/*2, *4/*4, *5/*1
            cr_ = locus.sample(np.random.default_rng(20_000 + s), sc[1], 100, lo=8000, hi=16000)
            pr, keep, vtruth = panel.problems(pkg, np.random.default_rng(30_000 + s))
            cls.CACHE[s] = dict(hla_reads=reads, hla_truth=truth, cyp_payload=pkg.ffi.encode_bam4(cr_), cyp_expected=sc[2], cyp_n=len(cr_), var=(pr, keep, vtruth))
        return cls.CACHE[s]

    def __init__(self, pkg, fx, locus, scen, panel, samples):
        self.samples, self.fx = list(samples), fx
        self.hla_reads, self.sample_of, self.hla_truth = [], [], {}
        self.cyp_payloads, self.cyp_expected, self.cyp_reads = [], [], 0
        self.var_problems, self._keep, self.var_truth = [], [], []
        for k, s in enumerate(self.samples):
            d = self.sample_data(pkg, fx, locus, scen, panel, s)
            self.hla_reads += d["hla_reads"]; self.sample_of += [k] * len(d["hla_reads"])
            for g, pick in d["hla_truth"].items():
                self.hla_truth[(k, g)] = pick
            self.cyp_payloads.append(d["cyp_payload"]); self.cyp_expected.append(d["cyp_expected"]); self.cyp_reads += d["cyp_n"]
            pr, keep, truth = d["var"]
            self.var_problems += pr; self._keep.append(keep); self.var_truth += truth
        self.hla_payload = pkg.ffi.encode_bam4(self.hla_reads)
        self.beside = None                       # (context, CypDb) the CYP2D6 half of a pass runs on, beside the HLA half (cohort_line sets it)
        self.n_reads = len(self.hla_reads) + self.cyp_reads
        self.n_genes_panel = len(panel.genes)

    def step(self, pkg, ctx, db, cdb, shard, group, rank, same_count=False):
        """uploads, calls and the gather of one pass over the share; returns (records table, counts of calls equal to the truth).
        same_count: every rank holds the same number of records in this round (the caller knows: equal shares), so the counts are not exchanged"""
        fx = self.fx
        genes = list(range(len(fx.genes)))
        import collections
        tm = self.host_s = collections.defaultdict(float, getattr(self, "host_s", {}))
        t0 = time.perf_counter()
        # the two loci of the share side by side, each on a context of its own (a host thread, a stream and its helpers, a copy stream): the HLA call is K1's big grids and
        # a few hundred consensus problems in lockstep, the CYP2D6 call chains of small launches -- back to back they took 0.27 + 0.45 s for 256 samples
        cyp_ctx, cyp_db = self.beside or (ctx, cdb)
        box = {}

        def cyp_half():
            try:
                ts = time.perf_counter()
                sets = []
                for p in self.cyp_payloads:                                      # (one upload in flight per context: each waits for the one before)
                    sets.append(cyp_ctx.upload_format(pkg.ffi.SP_SEQ_BAM4, *p, wait=False))
                for c in sets:
                    c.wait()
                box["sets"] = sets
                box["calls"] = cyp_db.diplotype_cohort(sets)
                box["seconds"] = time.perf_counter() - ts
            except Exception as e:                                              # surfaces in the main thread
                box["error"] = e
        th = threading.Thread(target=cyp_half) if self.beside else None
        if th:
            th.start()
        R = ctx.upload_format(pkg.ffi.SP_SEQ_BAM4, *self.hla_payload, wait=False).wait()
        t1 = time.perf_counter(); tm["upload"] += t1 - t0
        k1 = db.realign_reads(R)
        cohort, _ = db.diplotype_cohort(len(self.samples), self.sample_of, genes, R, k1)
        t2 = time.perf_counter(); tm["hla"] += t2 - t1
        if th:
            th.join()
        else:
            cyp_half()
        if "error" in box:
            raise box["error"]
        cyp_sets, cyp = box["sets"], box["calls"]
        t3 = time.perf_counter(); tm["cyp2d6"] += box["seconds"]; tm["loci_side_by_side"] += t3 - t1
        var = ctx.variant_solve_batch(self.var_problems)
        t4 = time.perf_counter(); tm["variant_genes"] += t4 - t3
        n_rec = len(self.samples) * (len(genes) + 1 + self.n_genes_panel)
        rec = np.zeros(n_rec, shard.CALL_DTYPE)
        ok_hla = ok_cyp = ok_var = 0
        at = 0
        for k, s in enumerate(self.samples):
            for g in genes:
                c = cohort[k][g][0]
                rec[at] = (s, g, c.allele1, c.allele2); at += 1
                ok_hla += all(same_allele(fx, x, y) for x, y in zip(sorted([c.allele1, c.allele2]), self.hla_truth[(k, g)]))
            call = cyp[k][0]
            ok_cyp += sorted([call.hap1.decode(), call.hap2.decode()]) == sorted(self.cyp_expected[k])
            rec[at] = (s, len(genes), int(call.status), int(call.n_consensus)); at += 1
            for gi in range(self.n_genes_panel):
                score, dips = var[k * self.n_genes_panel + gi]
                h1, h2 = self.var_truth[k * self.n_genes_panel + gi]
                ok_var += any({d[0], d[1]} == {h1, h2} for d in dips)
                d0 = dips[0] if dips else (-1, -1, 0)
                rec[at] = (s, len(genes) + 1 + gi, d0[0], d0[1]); at += 1
        table = shard.gather_calls(rec, group=group, same_count=same_count)
        R.close()
        for c in cyp_sets:
            c.close()
        tm["records_and_gather"] += time.perf_counter() - t4
        return table, (ok_hla, ok_cyp, ok_var)


def cohort_line(pkg, ctx, fx, db, cdb, locus, scen, world, rank, group, args, barrier, max_over_ranks, cdb_source=None, ctx_device=0):
    from pb_starphase_amd import shard
    panel = VariantPanel.shared(pkg)          # (one panel per process: the cached samples' problems point into its arrays)
    mine = shard.partition(args.cohort_samples, world, rank)
    # samples per cohort call: a rank's whole share (the calls keep groups of samples in lockstep, and the larger the groups the fewer launches a sample costs:
    # 256 samples in calls of 32 / 64 / 128 / 256: 170 / 201 / 257 / 280 samples/s on one GPU)
    per_call = max(1, min(len(mine), int(os.environ.get("SP_BENCH_PER_CALL", "256"))))
    chunks = [mine[i:i + per_call] for i in range(0, len(mine), per_call)]
    shares = [CohortShare(pkg, fx, locus, scen, panel, c) for c in chunks]
    cfg2, gd2 = cdb_source
    ctx2 = pkg.Context(ctx_device)
    cdb2 = pkg.ffi.CypDb(ctx2, cfg2, gd2, locus.sequence, locus.start)
    ctx2.set_option("k8_persistent", cyp_persistent())
    for sh in shares:
        sh.beside = (ctx2, cdb2)
    # every rank makes the same number of gathers per pass, whatever its share: shares differ by one sample when the cohort does not divide by the ranks (and a rank
    # may hold none at all), so the rounds a rank has no chunk for are gathers of zero records, and the counts are only taken as known when all shares are equal
    cap_call = int(os.environ.get("SP_BENCH_PER_CALL", "256"))
    sizes = [len(shard.partition(args.cohort_samples, world, r)) for r in range(world)]
    rounds = max(-(-n // max(1, min(n, cap_call))) if n else 0 for n in sizes)
    equal_shares = len(set(sizes)) == 1
    empty = np.zeros(0, shard.CALL_DTYPE)

    def one_pass(only_first=False):
        good_sum, n_tab = np.zeros(3, np.int64), 0
        for q in range(1 if only_first else rounds):
            if q < len(shares):
                table, good = shares[q].step(pkg, ctx, db, cdb, shard, group, rank, same_count=equal_shares)
                good_sum += np.array(good)
            else:
                table = shard.gather_calls(empty, group=group, same_count=False)
            n_tab += len(table)
        return good_sum, n_tab
    for _ in range(args.warmup):
        one_pass(only_first=True)
    for sh in shares:
        sh.host_s = {}
    barrier()
    t0 = time.perf_counter()
    ok = np.zeros(3, np.int64)
    n_table = 0
    for _ in range(args.steps):
        good, n_tab = one_pass()
        ok += good; n_table += n_tab
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    # what a rank's share costs by its size (one GPU): the share a rank holds at N = 8 / 4 / 2 of a 256-sample cohort, each in one pass after a warm-up pass
    by_share = None
    if world == 1 and (getattr(args, "cohort_shares", False) or os.environ.get("SP_BENCH_COHORT_SHARES")) and mine:
        by_share = {}
        for n_sh in (32, 64, 128):
            if n_sh >= len(mine):
                continue
            sh = CohortShare(pkg, fx, locus, scen, panel, mine[:n_sh])
            sh.beside = shares[0].beside
            sh.step(pkg, ctx, db, cdb, shard, None, 0)
            best = None
            for _ in range(2):
                sh.host_s = {}
                barrier(); t1 = time.perf_counter()
                sh.step(pkg, ctx, db, cdb, shard, None, 0)
                barrier(); d1 = time.perf_counter() - t1
                if best is None or d1 < best[0]:
                    best = (d1, dict(sh.host_s))
            by_share[str(n_sh)] = {"seconds": best[0], "samples_per_s": n_sh / best[0], "host_seconds": {k: round(v, 4) for k, v in best[1].items()}}
    my_reads = sum(sh.n_reads for sh in shares)
    totals = np.array([my_reads, len(mine), ok[0], ok[1], ok[2]], np.int64)
    if group is not None:
        totals = group.gather(totals).sum(0)
    reads_all, samples_all = int(totals[0]), int(totals[1])
    n_genes = len(fx.genes)
    return {"value": reads_all * args.steps / dt, "unit": "reads/s", "samples_per_s": samples_all * args.steps / dt, "ms_per_step": 1e3 * dt / args.steps,
            "samples": samples_all, "reads_per_pass": reads_all, "by_share_size": by_share,
            "rank0_host_seconds_per_pass": {k: sum(sh.host_s.get(k, 0.0) for sh in shares) / max(1, args.steps) for k in ("upload", "hla", "cyp2d6", "loci_side_by_side", "variant_genes", "records_and_gather")}, "records_gathered_per_pass": n_table // max(1, args.steps),
            "calls_equal_truth": {"hla": f"{int(totals[2])}/{samples_all * n_genes * args.steps}", "cyp2d6": f"{int(totals[3])}/{samples_all * args.steps}",
                                  "variant_genes_truth_among_reported": f"{int(totals[4])}/{samples_all * len(panel.genes) * args.steps}"},
            "workload": f"BASELINE configs[4]: {args.cohort_samples} synthetic WGS-style samples x (HLA-A / -B ~44 reads per gene, CYP2D6 ~100 reads, {len(panel.genes)} variant genes), "
                        f"sharded by sample over {world} rank(s) in calls of {per_call} samples: upload (BAM 4-bit) -> sp_hla_realign_reads + sp_hla_diplotype_cohort -> "
                        "sp_cyp_diplotype_cohort -> sp_variant_solve_batch -> one gather of the call records"}


def streams_block(pkg, fx, db, ctx, cfg, gene_def, locus, scen, device_index, rank, world, steps, reads, cyp_reads, barrier, max_over_ranks, own_device=True):
    """the second block of the N > 1 line: every rank its own stream of the headline's samples (one sample per step, both loci, a new upload every step, the six CYP2D6 scenarios
    in turn), nothing exchanged -- `value` = all ranks' reads / the slowest rank's time between the barriers (weak scaling by independent samples).  The CYP2D6 consensus runs in the
    library's own choice of mode (k8_persistent auto) when the rank has its device to itself, as launch pairs when ranks share a device (two processes' persistent batches would wait
    for each other's CUs)"""
    samples = [HlaSample(pkg, fx, reads, 1000 + rank + 100 * k) for k in range(2)]
    cyp_samples = [CypSample(pkg, locus, scen[k], cyp_reads, 7 + k) for k in range(len(scen))]
    ctx_c = pkg.Context(device_index)
    cdb_c = pkg.ffi.CypDb(ctx_c, cfg, gene_def, locus.sequence, locus.start)
    if not own_device:
        ctx_c.set_option("k8_persistent", 0)
    genes = list(range(len(fx.genes)))
    ok = [0, 0]

    def hla_work(R, i):
        return db.diplotype_genes(genes, R, db.realign_reads(R))[0]

    def cyp_work(R, i):
        call = cdb_c.diplotype(R)[0]
        ok[0] += int(sorted([call.hap1.decode(), call.hap2.decode()]) == sorted(cyp_samples[i % len(cyp_samples)].expected)); ok[1] += 1
        return call

    def lanes(n):
        return [Lane(pkg, ctx, [s_.payload for s_ in samples], hla_work, n, True), Lane(pkg, ctx_c, [c.payload for c in cyp_samples], cyp_work, n, True)]
    run_lanes(lanes(1))
    ok[0] = ok[1] = 0
    ls = lanes(steps)
    for x in ls:
        x.wait_first()
    barrier(); ctx_c.synchronize()
    t0 = time.perf_counter()
    run_lanes(ls)
    barrier(); ctx_c.synchronize()
    dt = max_over_ranks(time.perf_counter() - t0)
    per_step = samples[0].n + cyp_samples[0].n
    return {"value": world * per_step * steps / dt, "unit": "reads/s", "scaling": "weak", "ms_per_step": 1e3 * dt / steps, "steps": steps, "reads_per_step_per_rank": per_step,
            "cyp2d6_calls_equal_truth_rank0": f"{ok[0]}/{ok[1]}",
            "workload": "the N = 1 headline's stream of samples on every rank (its own samples; HLA-A / -B %d reads + CYP2D6 %d reads per step, the six scenarios in turn), no exchange" % (samples[0].n, cyp_samples[0].n)}


def cohort_leg_in_a_process_of_its_own(n_samples):
    """BASELINE configs[4] on this one GPU, in a process of its own, started like a rank of the N > 1 run and BEFORE this process touches the GPU: by the time the legs run this
    process holds two dozen streams, and the HIP runtime deals streams to its 16 hardware queues in the order they are made -- a cohort call's streams then share queues with
    each other and run their chains one after the other (a 32-sample share: 0.18 s there against 0.14 s in a process that runs nothing else)"""
    try:
        env = dict(os.environ, SP_BENCH_COHORT_SHARES="1")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        import tempfile
        with tempfile.TemporaryDirectory() as tmp:
            side = os.path.join(tmp, "cohort_full.json")
            child = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", "cohort", "--steps", "2", "--warmup", "1", "--no-extra-legs", "--cohort-samples", str(n_samples),
                                    "--full-out", side], env=env, capture_output=True, text=True, timeout=900)
            if child.returncode != 0 or not os.path.exists(side):
                raise RuntimeError("the cohort leg's process ended with %d: %s" % (child.returncode, child.stderr[-500:]))
            return json.load(open(side))["cohort"]
    except Exception as e:                                                      # (a leg, not the headline: say so and go on)
        return {"error": str(e)}


# ---------------------------------------------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)          # four turns through the six CYP2D6 scenarios: with lanes in flight the last long sample of a short run is a tail
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=10000, help="HLA reads of the sample (configs[1])")
    ap.add_argument("--cyp-reads", type=int, default=2000, help="CYP2D6 reads of the sample (configs[2])")
    ap.add_argument("--cohort-samples", type=int, default=256)
    ap.add_argument("--workload", choices=("auto", "sample", "cohort"), default="auto",
                    help="auto: at N = 1 the sample (one sample per step, both loci: BASELINE configs[1] + configs[2]); at N > 1 the cohort (BASELINE configs[4]: 256 samples sharded over "
                         "the ranks, the call records gathered through sp_gather_results over RCCL), with the ranks' independent streams of samples as a second block of the line")
    ap.add_argument("--cyp-lanes", type=int, default=6, help="CYP2D6 samples in flight beside the HLA half of the stream (contexts that share the stream of CYP2D6 samples)")
    ap.add_argument("--hla-lanes", type=int, default=2, help="HLA samples in flight (contexts that share the stream of HLA samples)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-hla-reads", type=int, default=5000, help="HLA reads of sample 0 the CPU leg runs (0: all of them)")
    ap.add_argument("--cpu-cyp-reads", type=int, default=1000, help="reads of every CYP2D6 scenario of the mix the CPU leg runs (0: all of them)")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the resident-HLA, CYP2D6-scenario, cohort and K5 legs")
    ap.add_argument("--full-out", default=os.path.join(ROOT, "bench_full.json"), help="where the whole record goes (legs, critical path, per-scenario CPU blocks); stdout carries ONE compact line")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # N ranks: every rank runs the same stream of samples on its own GPU (its own samples: the seeds carry the rank) -- per-GPU work fixed, nothing shared, `value` the sum over
    # the ranks (weak scaling by independent samples); `--workload cohort` is the other question: ONE 256-sample cohort sharded over the ranks (strong scaling)
    workload = args.workload if args.workload != "auto" else ("sample" if world == 1 else "cohort")
    pkg = ge.load_package()
    from pb_starphase_amd import synth, shard
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cyp_cases_real as cr
    fx = synth.HlaFixture()
    samples = [HlaSample(pkg, fx, args.reads, 1000 + rank + 100 * k) for k in range(2)] if workload == "sample" else []
    cfg, gene_def = cr.load_db()
    locus = synth.Chr22Locus(cfg, gene_def, seed=3)
    scen = cr.scenarios(locus)
    # the steps cycle through ALL SIX configs[2] scenarios (*1/*2, *4/*4, *5/*1, *4+*68/*1, *10+*36/*10, *2x2/*1; round 4 alternated the first two, the cheap third of the config)
    cyp_samples = [CypSample(pkg, locus, scen[k], args.cyp_reads, 7 + k) for k in range(len(scen))] if workload == "sample" else []
    cohort_in_own_process = cohort_leg_in_a_process_of_its_own(args.cohort_samples) if (world == 1 and workload == "sample" and not args.no_extra_legs) else None
    cb, cpu_ref = None, None
    if not args.no_cpu_baseline and world == 1 and workload == "sample":
        # both loci of sample 0 through the reference-call-pattern CPU port; forks workers: must happen before anything touches the GPU
        cb, cpu_ref = cpu_baseline(fx, samples[0].wl.reads, (cfg, gene_def, locus), [(c.name, c.reads) for c in cyp_samples], args.cpu_hla_reads, args.cpu_cyp_reads)
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
    if world > 1 and backend != "nccl":
        os.environ.setdefault("SP_K8_PERSISTENT", "0")      # (ranks that share a device: the header's rule for several processes on one device -- their CU leases cannot see each other)

    ctx = pkg.Context(device_index)
    db = fx.make_db(pkg, ctx)
    cdb = pkg.ffi.CypDb(ctx, cfg, gene_def, locus.sequence, locus.start)
    group, gather_via = None, "single rank"
    group_fallback = None
    if world > 1 and workload == "cohort":                  # (the streams of samples exchange nothing: no communicator)
        group, gather_via, group_fallback = bring_up_group(pkg, shard, backend, coll_dev, device_index, rank, world)

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

    if workload == "cohort":
        line = cohort_line(pkg, ctx, fx, db, cdb, locus, scen, world, rank, group, args, barrier, max_over_ranks, (cfg, gene_def), device_index)
        blocks = {}
        if world > 1 and not args.no_extra_legs:
            # (a) the same cohort on ONE GPU (rank 0 alone, the others wait): what the N ranks' value is to be held against; (b) the ranks' independent streams of samples
            if rank == 0:
                try:
                    one_args = argparse.Namespace(**vars(args)); one_args.steps, one_args.warmup = 1, 1
                    solo = cohort_line(pkg, ctx, fx, db, cdb, locus, scen, 1, 0, shard.SoloGroup(), one_args, lambda: (torch.cuda.synchronize(), ctx.synchronize()), lambda d: d, (cfg, gene_def), device_index)
                    blocks["one_gpu_same_cohort"] = {"value": solo["value"], "samples_per_s": solo["samples_per_s"], "ms_per_step": solo["ms_per_step"],
                                                     "value_over_n_times_this": line["value"] / (world * solo["value"])}
                except Exception as e:                                              # (a second block, not the line's value: say so and go on; no collective inside)
                    blocks["error"] = "one_gpu_same_cohort: " + str(e)
            barrier()
            blocks["independent_streams"] = streams_block(pkg, fx, db, ctx, cfg, gene_def, locus, scen, device_index, rank, world, min(args.steps, 6), args.reads, args.cyp_reads,
                                                          barrier, max_over_ranks, own_device=(backend == "nccl"))
        if rank == 0:
            out = {"metric": "HiFi reads/sec diplotyped (HLA+CYP2D6)", "value": line["value"], "unit": "reads/s", "n_gpus": world, "steps": args.steps,
                   "warmup": args.warmup, "ms_per_step": line["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                   "dtype": "int32", "data": "synthetic",
                   "config": {"workload": line["workload"], "samples": line["samples"], "parallelism": f"samples sharded over {world} GPU(s), one all-gather of the call records per pass through {gather_via}"
                              if backend == "nccl" else f"samples sharded over {world} rank(s) on shared devices, gather through torch.distributed ({backend})"},
                   "cohort": line, "gather_via": gather_via, "group_fallback": group_fallback, "one_gpu_same_cohort": blocks.get("one_gpu_same_cohort"), "independent_streams": blocks.get("independent_streams"), "second_blocks_error": blocks.get("error"),
                   "roofline": None, "cpu_baseline": None,
                   "note": "BASELINE configs[4], strong scaling: the cohort's work is fixed, every rank owns samples / N of it and hands its whole share to the library in one call; the only exchange is ONE "
                           "all-gather of the call records per pass (sp_gather_results).  The calls keep groups of samples in lockstep, so a rank's rate falls with its share (`legs.cohort.by_share_size` "
                           "of the N = 1 line: one GPU on shares of 32 / 64 / 128 samples).  `one_gpu_same_cohort`: the same 256 samples on rank 0 alone, in this run; `independent_streams`: every rank "
                           "its own stream of the N = 1 headline's samples (weak scaling, nothing exchanged).  roofline / cpu_baseline: the N = 1 line carries them"}
            compact = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
            compact["value"], compact["ms_per_step"] = _r(out["value"], 6), _r(out["ms_per_step"], 6)
            compact["config"] = {"workload": "BASELINE configs[4]: %d-sample synthetic cohort (HLA-A/-B, CYP2D6, %d variant genes per sample) sharded by sample over the ranks"
                                             % (line["samples"], line["records_gathered_per_pass"] // max(1, line["samples"]) - 3),
                                 "samples": line["samples"], "reads_per_pass": line["reads_per_pass"], "ranks": world, "gather_via": gather_via, "group_fallback": group_fallback}
            compact["roofline"], compact["cpu_baseline"] = None, None
            compact["cohort"] = {"samples_per_s": _r(line["samples_per_s"]), "calls_equal_truth": line["calls_equal_truth"], "records_gathered_per_pass": line["records_gathered_per_pass"],
                                 "by_share_size_samples_per_s": {k: _r(v["samples_per_s"]) for k, v in (line.get("by_share_size") or {}).items()} or None}
            one, ind = blocks.get("one_gpu_same_cohort"), blocks.get("independent_streams")
            compact["summary"] = {"one_gpu_same_cohort_reads_per_s": _r(one["value"]) if one else None, "value_over_n_times_one_gpu": _r(one["value_over_n_times_this"]) if one else None,
                                  "independent_streams_reads_per_s": _r(ind["value"]) if ind else None, "second_blocks_error": blocks.get("error"),
                                  "note": "roofline / cpu_baseline: the N = 1 line carries them"}
            emit(out, compact, args.full_out)
        if group is not None and hasattr(group, "close"):
            group.close()
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        if "exit without teardown" in ABANDONED:
            sys.stdout.flush(); sys.stderr.flush()
            os._exit(0)
        return

    # ------------------------------------------------------------------------------------------------ the sample: HLA-A / -B + CYP2D6, a new one every step
    # the CYP2D6 half runs beside the HLA half on contexts (stream, pools) of its own: `--cyp-lanes` of them share the stream of CYP2D6 samples (a lane takes the next sample
    # when it is free).  One CYP2D6 sample is a chain of a few hundred dependent consensus steps that keeps a fraction of the device busy (critical_path below); two of them in
    # flight fill the time the HLA half of a step takes
    n_cyp_lanes, n_hla_lanes = max(1, args.cyp_lanes), max(1, args.hla_lanes)
    hla_ctxs = [ctx] + [pkg.Context(device_index) for _ in range(n_hla_lanes - 1)]
    hla_dbs = [db] + [fx.make_db(pkg, c) for c in hla_ctxs[1:]]
    ctx_h = ContextGroup(hla_ctxs)
    cyp_ctxs = [pkg.Context(device_index) for _ in range(n_cyp_lanes)]
    cyp_dbs = [pkg.ffi.CypDb(c, cfg, gene_def, locus.sequence, locus.start) for c in cyp_ctxs]
    ctx_c = ContextGroup(cyp_ctxs)                         # (the group reads the list: lanes a leg adds later count)
    # the consensus mode is the library's own choice (k8_persistent 2 = auto, the default of every context); SP_BENCH_HEADLINE_PERSISTENT = 0 / 1 forces launch pairs / persistent kernels
    forced = os.environ.get("SP_BENCH_HEADLINE_PERSISTENT")
    headline_mode = {"persistent": forced != "0", "fallback": None, "forced": forced}
    if forced is not None:
        ctx_c.set_option("k8_persistent", int(forced))
    genes = list(range(len(fx.genes)))
    last = {}

    def hla_work(R, i):
        db_ = hla_dbs[hla_ctxs.index(R.ctx)]
        o = db_.realign_reads(R)
        calls = db_.diplotype_genes(genes, R, o)[0]
        last["hla"] = (i, o, calls)
        return calls

    cyp_log = []                                            # (step, scenario index, call strings, seconds of the library call) of every CYP2D6 call of the lanes

    def cyp_work(R, i):
        if headline_mode["persistent"] and os.environ.get("SP_BENCH_INJECT_FAILURE"):       # (a test of the fall-back below: profiles/scripts/r04_run67.sh)
            raise RuntimeError("injected failure of the persistent mode")
        t_ = time.perf_counter()
        call, _cons, _labels = cyp_dbs[cyp_ctxs.index(R.ctx)].diplotype(R)
        last["cyp"] = (i, call)
        cyp_log.append((i, i % len(cyp_samples), call.hap1.decode(), call.hap2.decode(), time.perf_counter() - t_))
        return call

    def make_lanes(steps, fresh=True, n_lanes=None):
        shared, shared_h = Tickets(steps), Tickets(steps)
        return [Lane(pkg, c, [s.payload for s in samples], hla_work, steps, fresh, tickets=shared_h) for c in hla_ctxs] + [Lane(pkg, c, [s.payload for s in cyp_samples], cyp_work, steps, fresh, tickets=shared)
                                                                                          for c in cyp_ctxs[:n_lanes if n_lanes else n_cyp_lanes]]

    def agree(failed):
        """has the persistent mode failed on ANY rank?  (one all-reduce: the ranks switch to launch pairs together instead of one of them raising and the others waiting at a barrier)"""
        if world == 1:
            return failed
        t = torch.tensor([1 if failed else 0], dtype=torch.int32, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return bool(int(t.item()))

    def timed_region():
        # the warm-up steps (no barrier inside): where the persistent kernels of a batch cannot run side by side the library ends the batch after its time-out with an error -- here
        err = None
        try:
            run_lanes(make_lanes(max(1, args.warmup)))
        except Exception as e:
            if not headline_mode["persistent"]:
                raise
            err = e
        if headline_mode["persistent"] and agree(err is not None):
            headline_mode.update(persistent=False, fallback=str(err) if err is not None else "another rank's persistent consensus kernels failed")
            sys.stderr.write("bench: persistent consensus kernels failed (%s): the headline runs with a launch pair per step\n" % headline_mode["fallback"])
            ctx_c.set_option("k8_persistent", 0)
            run_lanes(make_lanes(max(1, args.warmup)))
        ctx_h.profile_reset(); ctx_c.profile_reset()
        del cyp_log[:]
        lanes = make_lanes(args.steps)                      # (sample 0's bytes start travelling here: the pipeline is full when the clock starts)
        for x in lanes:
            x.wait_first()
        barrier(); ctx_c.synchronize(); ctx_h.synchronize()
        t0 = time.perf_counter()
        run_lanes(lanes)
        barrier(); ctx_c.synchronize(); ctx_h.synchronize()
        return lanes, max_over_ranks(time.perf_counter() - t0)
    try:
        lanes, dt = timed_region()
    except Exception as e:
        if not headline_mode["persistent"] or world > 1:    # (a failure inside the timed steps of a multi-rank run: the ranks are in lockstep, nothing to agree on any more)
            raise
        # the persistent kernels failed after the warm-up: a launch pair per step instead, from the start
        headline_mode.update(persistent=False, fallback=str(e))
        sys.stderr.write("bench: persistent consensus kernels failed (%s): the headline runs with a launch pair per step\n" % e)
        ctx_c.set_option("k8_persistent", 0)
        lanes, dt = timed_region()
    reads_per_step = samples[0].n + cyp_samples[0].n        # (every scenario has the same number of reads)
    line_value = world * reads_per_step * args.steps / dt
    timed_cyp = list(cyp_log)
    mix = {}
    for _i, k, h1, h2, sec in timed_cyp:
        m = mix.setdefault(cyp_samples[k].name, {"steps": 0, "calls_equal_truth": 0, "call_ms": 0.0})
        m["steps"] += 1; m["calls_equal_truth"] += int(sorted([h1, h2]) == sorted(cyp_samples[k].expected)); m["call_ms"] += 1e3 * sec
    for m in mix.values():
        m["call_ms"] = round(m["call_ms"] / max(1, m["steps"]), 2)
    if world > 1:
        args.no_extra_legs = True                           # (the legs are one-GPU questions; the ranks only run the headline together)
    # a lane's host time per step: waiting for the sample's bytes / starting the next upload / the library calls / closing the sample's read set
    lane_ms = [{k: round(1e3 * getattr(x, "t_" + k, 0.0) / max(1, args.steps), 3) for k in ("wait", "start", "work", "close")} for x in lanes]

    e2e_names = ("anchor_k1", "k1s_seeds", "k1s_groups", "k1s_dp", "k1s_dp_big", "k1s_select", "k1s_cells", "k1s_af_trace", "k1s_af_dp", "anchor_k2", "anchor_type", "k1_finalize", "cons_steps", "type_consensus_ref",
                 "k2_cells_cdna", "k2_cells_dna", "k2_scan")
    kernel_ms = {k: ctx_h.profile_get(k)[0] / max(1, args.steps) for k in e2e_names}
    cyp_kernel_ms = {k: ctx_c.profile_get(k)[0] / max(1, args.steps) for k in ("cons_steps", "k5_pairs", "k9_graph")}

    def critical_path(c):
        """the consensus launch chain of a context per step: how many dependent steps the slowest problem of every batch made, and where their time went -- the step
        kernel from its first workgroup's start to its last one's end, the single-workgroup control step, and what lies between the kernels (100 MHz device clock,
        written by the kernels themselves: sp_consensus.hip, CSearch::step_ticks / gap_ticks / ticks)"""
        ms = lambda name: c.profile_get(name)[2] / 1e5 / max(1, args.steps)
        steps = c.profile_get("cons_path_steps")[2] / max(1, args.steps)
        persistent = c.profile_get("cons_persistent_batches")[2] > 0
        out = {"mode": "persistent kernels: two launches per batch, the step and control workgroups of a problem hand over through one word each in memory" if persistent
                       else "a launch pair per step", "dependent_steps": steps, "launches_per_step": 0 if persistent else 2,
               "step_kernel_ms": ms("cons_path_step_ticks"), "control_kernel_ms": ms("cons_path_control_ticks"),
               "between_kernels_ms": ms("cons_path_gap_ticks"), "chain_ms": c.profile_get("cons_steps")[0] / max(1, args.steps)}
        if out["chain_ms"] > 0:                             # the share of the chain's wall time in which one of its kernels is running at all (the rest: the device waits for the next launch)
            out["kernels_running_fraction"] = (out["step_kernel_ms"] + out["control_kernel_ms"]) / out["chain_ms"]
        if steps > 0:
            out["per_step_us"] = {"step_kernel": 1e3 * out["step_kernel_ms"] / steps, "control_kernel": 1e3 * out["control_kernel_ms"] / steps,
                                  "between_kernels_per_boundary": 1e3 * out["between_kernels_ms"] / (2 * steps)}
            # the two boundaries of a step apart: the step kernel's end -> the control step's start (one workgroup with ~ 130 KB of LDS looking for a CU), the control step's end ->
            # the first step workgroup's start (eight waves at 240 registers looking for a CU)
            gc = ms("cons_path_gap_ctl_ticks")
            out["boundary_us"] = {"step_end_to_control_start": 1e3 * gc / steps, "control_end_to_step_start": 1e3 * (out["between_kernels_ms"] - gc) / steps}
            # the control step by part (the slowest problem of every batch, all steps of the batch -- not only the chain's): loading the work order and adding up the
            # workgroups' votes / the result of the step (columns decided, tapes written) / the search for the next node / the next work order and the write-back
            out["control_parts_us_per_step"] = {k: 1e3 * ms("cons_ticks_" + k) / steps for k in ("reduce", "result", "search", "tail")}
        out["note"] = ("the chain is latency bound: every step is step body -> control body -> next step body; as a launch pair per step each arrow is a dependent same-stream "
                       "kernel boundary (MI355X_MICROARCH.md: 1.45 us on an idle device), in persistent mode a write-through store + drain + flag and a poll (handoff-flag: 1.3-5 us)")
        return out
    crit = {"kernel": "cons_step_kernel + cons_control_kernel (K8, the consensus search)", "cyp2d6": critical_path(ctx_c), "hla": critical_path(ctx_h)}
    batches_per_step = ctx_c.profile_get("cons_persistent_batches")[2] / max(1, args.steps)
    if headline_mode["persistent"] and batches_per_step == 0:
        headline_mode["persistent"] = False                 # (the library chose launch pairs: its condition for the persistent kernels does not hold in this process)
    cons = {k: ctx_h.profile_get(n)[2] / max(1, args.steps) for k, n in (("launch_triples_per_step", "cons_windows"), ("cut_windows_per_step", "cons_cut_windows"),
                                                                         ("expansions_per_step", "cons_expansions"), ("nodes_expanded_per_step", "cons_columns"))}
    host_ms = {k: ctx_h.profile_get("host:" + k)[0] / max(1, args.steps) for k in ("hla_segments", "hla_dual_hpc", "hla_groups", "hla_typing", "k8_loop", "k1_total", "hla_genes_total")}
    cyp_host_ms = {k: ctx_c.profile_get("host:cyp_" + k)[0] / max(1, args.steps) for k in ("regions", "segments", "consensus", "merge", "typing", "weights", "chains", "chain_pair")}
    cyp_host_ms["k8"] = {k: [round(ctx_c.profile_get("host:k8_" + k)[0] / max(1, args.steps), 2), ctx_c.profile_get("host:k8_" + k)[1] // max(1, args.steps)] for k in ("prologue", "loop", "result_wait", "epilogue")}
    # the calls of the last step against the truth the reads were simulated from
    i_h, k1_out, gene_calls = last["hla"]
    smp = samples[i_h % 2]
    ok = sum(all(same_allele(fx, a, b) for a, b in zip(sorted([c.allele1, c.allele2]), sorted((smp.truth[g] * 2)[:2]))) for g, (c, _c1, _c2) in enumerate(gene_calls))
    i_c, cyp_call = last["cyp"]
    cyp_ok = all(m["calls_equal_truth"] == m["steps"] for m in mix.values())
    k1_gene_ok = float(np.mean([k1_out[r]["gene"] == smp.wl.read_truth[r][0] for r in range(smp.n)]))

    # the bytes' way alone: one synchronous upload of the HLA half in either form
    t_up = {}
    for name, fmt, payload in (("bam4", pkg.ffi.SP_SEQ_BAM4, samples[0].payload), ("ascii", pkg.ffi.SP_SEQ_ASCII, pkg.ffi._concat(samples[0].wl.reads) + (None,))):
        best = None
        for _ in range(3):
            t1 = time.perf_counter()
            S = ctx.upload_format(fmt, payload[0], payload[1], payload[2])
            d = time.perf_counter() - t1
            S.close()
            best = d if best is None or d < best else best
        t_up[name] = {"seconds": best, "bytes": int(len(payload[0])), "GBps": len(payload[0]) / best / 1e9, "bases_per_s": samples[0].ascii_bytes / best}

    legs = {}
    other_leg = "headline_with_launch_pairs" if headline_mode["persistent"] else "headline_with_persistent_consensus"
    if not args.no_extra_legs and headline_mode["fallback"] is None:
        # the headline's step once more in the other consensus mode of the CYP2D6 context (sp_ctx_set_option "k8_persistent"): the same samples, uploads and calls.
        # (persistent kernels need every stream of the process on a hardware queue of its own and fail with an error, after a four-second time-out, where it is not --
        #  which is reported here instead of raised)
        try:
            ctx_c.set_option("k8_persistent", 0 if headline_mode["persistent"] else 1)
            run_lanes(make_lanes(max(1, args.warmup)))
            ctx_c.profile_reset()
            lanes_p = make_lanes(args.steps)
            for x in lanes_p:
                x.wait_first()
            ctx_c.synchronize(); ctx.synchronize()
            t1 = time.perf_counter()
            run_lanes(lanes_p)
            ctx_c.synchronize(); ctx.synchronize()
            d_p = time.perf_counter() - t1
            legs[other_leg] = {"value": reads_per_step * args.steps / d_p, "unit": "reads/s", "ms_per_step": 1e3 * d_p / args.steps,
                                                          "cyp2d6_cons_steps_ms": ctx_c.profile_get("cons_steps")[0] / max(1, args.steps),
                                                          "cyp2d6_call_equals_truth": sorted([last["cyp"][1].hap1.decode(), last["cyp"][1].hap2.decode()]) == sorted(cyp_samples[last["cyp"][0] % len(cyp_samples)].expected),
                                                          "host_wall_ms_cyp2d6": {k: round(ctx_c.profile_get("host:cyp_" + k)[0] / max(1, args.steps), 2) for k in ("regions", "segments", "consensus", "merge", "typing", "weights", "chains", "chain_pair")},
                                                          "host_wall_ms_k8": {k: [round(ctx_c.profile_get("host:k8_" + k)[0] / max(1, args.steps), 2), ctx_c.profile_get("host:k8_" + k)[1] // max(1, args.steps)] for k in ("prologue", "loop", "result_wait", "epilogue")},
                                                          "critical_path_cyp2d6": critical_path(ctx_c)}
        except Exception as e:                                                   # (the library's own error text)
            legs[other_leg] = {"error": str(e)}
    ctx_c.set_option("k8_persistent", cyp_persistent())     # (the legs below: many streams, launch pairs unless SP_BENCH_CYP_PERSISTENT asks otherwise)
    if not args.no_extra_legs:
        # HLA alone, reads resident in HBM (round 2's headline): the K1 launch the roofline block describes runs here exactly as in the headline
        res_lane = Lane(pkg, ctx, [s.payload for s in samples], hla_work, args.steps, fresh_upload=False)
        ctx.synchronize()
        t1 = time.perf_counter()
        run_lanes([res_lane])
        ctx.synchronize()
        d_res = time.perf_counter() - t1
        legs["hla_resident"] = {"value": samples[0].n * args.steps / d_res, "unit": "reads/s", "ms_per_step": 1e3 * d_res / args.steps,
                                "workload": "BASELINE configs[1] alone: K1 + sp_hla_diplotype_genes on reads already in HBM, one sample at a time (the headline of round 2)"}
        # K1 both ways on sample 0's resident reads: the reference's call pattern (k1_best_n = 5, the default: seeds, chains, best_n) and the exhaustive search of every allele
        # (k1_best_n = 0, rounds 1-4), whose cells kernel was the roofline's kernel until round 5 (algorithmic bytes of the cells the launch executed)
        Rk = ctx.upload_format(pkg.ffi.SP_SEQ_BAM4, *samples[0].payload)
        k1 = {}
        for mode, label in ((5, "seeded_best_n_5"), (0, "exhaustive")):
            ctx.set_option("k1_best_n", mode)
            o_ = db.realign_reads(Rk)
            ctx.profile_reset(); ctx.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                o_ = db.realign_reads(Rk)
            d_k = (time.perf_counter() - t1) / 3
            k1[label] = {"ms_per_call": 1e3 * d_k, "reads_per_s": samples[0].n / d_k, "best_alleles_crc": int(np.bitwise_xor.reduce(o_["best_allele"].astype(np.int64) * 2654435761 % (1 << 31))),
                         "kernel_ms": {k: round(ctx.profile_get(k)[0] / 3, 3) for k in ("anchor_k1", "k1s_seeds", "k1s_groups", "k1s_dp", "k1s_dp_big", "k1s_select", "k1s_cells", "k1s_af_trace", "k1s_af_dp",
                                                                                       "k1_cells", "k1_cells_deep", "k1_finalize", "k1_af_trace", "k1_af_dp") if ctx.profile_get(k)[1]}}
            if mode == 0:
                ms_cells, launches, _c = ctx.profile_get("k1_cells")
                executed, resumed, active, exec_bytes = (ctx.counter("k1_cells_" + k) for k in ("executed", "resumed", "active", "bytes"))
                avg_ms = ms_cells / max(1, launches)
                k1[label]["cells_kernel"] = {"kernel": "k1_cells_kernel", "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": exec_bytes / max(1, launches),
                                             "algorithmic_GBs": exec_bytes / max(1, launches) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else None,
                                             "cells_executed_per_launch": executed / max(1, launches), "cells_resumed_per_launch": resumed / max(1, launches),
                                             "cells_active_per_launch": active / max(1, launches),
                                             "note": "an integer-DP kernel whose database sits in L2: instruction issue bounds it (profiles/r04/valu_k1_cells.json: 0.49 of the measured VALU peak), "
                                                     "0.19 GB cross HBM per launch (profiles/r04/traffic_k1_cells.json); the algorithmic bytes bill every executed cell its two whole sequences"}
            else:
                pass
        ctx.set_option("k1_best_n", 5)
        Rk.close()
        legs["k1_modes"] = k1
        # the same stream of samples with other numbers of CYP2D6 lanes (CYP2D6 samples in flight beside the HLA lane): one lane = a step is one locus's latency chain
        # (rounds 3-4's headline); more lanes = the chains of several samples fill the device's idle time until the HLA lane bounds the step
        lane_table = {str(n_cyp_lanes): {"value": line_value, "ms_per_step": 1e3 * dt / args.steps}}
        try:
            for n_l in (1, 2, 4):
                if str(n_l) in lane_table:
                    continue
                while len(cyp_ctxs) < n_l:
                    cyp_ctxs.append(pkg.Context(device_index)); cyp_dbs.append(pkg.ffi.CypDb(cyp_ctxs[-1], cfg, gene_def, locus.sequence, locus.start))
                    ctx_c.ctxs = cyp_ctxs
                run_lanes(make_lanes(1, n_lanes=n_l))
                fl = make_lanes(args.steps, n_lanes=n_l)
                for x in fl:
                    x.wait_first()
                ctx.synchronize(); ctx_c.synchronize()
                t1 = time.perf_counter()
                run_lanes(fl)
                ctx.synchronize(); ctx_c.synchronize()
                d_fl = time.perf_counter() - t1
                lane_table[str(n_l)] = {"value": reads_per_step * args.steps / d_fl, "ms_per_step": 1e3 * d_fl / args.steps}
            legs["cyp2d6_lanes"] = dict(lane_table, unit="reads/s", note="the headline's stream of samples with 1 / 2 / 4 CYP2D6 samples in flight beside the HLA lanes (`--cyp-lanes`; the headline's own entry is the line's value)")
        except Exception as e:                                                  # (a leg, not the headline: say so and go on)
            legs["cyp2d6_lanes"] = {"error": str(e)}
        legs["cyp2d6"] = cyp_leg(pkg, ctx, cdb, locus)
        legs["k5_chain_pairs"] = chain_pair_leg(pkg, ctx)
        legs["cohort"] = cohort_in_own_process             # (measured before this process touched the GPU: cohort_leg_in_a_process_of_its_own)

    peaks = {"valu_int_wave_instr_per_s": ctx.microbench("valu_int"), "match16_valu_wave_instr_per_s": ctx.microbench("match16"),
             "hbm_copy_bytes_per_s": ctx.microbench("hbm_copy")}
    # The dominant kernel of the step since round 5 is the consensus step kernel of the CYP2D6 context (K8: cons_step_persist_kernel<8>, cons_step_kernel<8> as launch pairs;
    # 55 % of the kernel time of profiles/r05/rocprof_r05_kernel_stats.csv with its control kernel).  Its time per step is measured live -- the chain between HIP events on the
    # library's stream (cons_steps) and, inside it, the step body by the device clock the kernels stamp (critical_path) --; its instruction and HBM byte counts per bench step come
    # from the committed rocprofv3 PMC passes of this workload (profiles/r05/counters_cons_step.json), quoted only while the kernel's source is the one they were made on
    import hashlib
    cons_sha = hashlib.sha256(open(os.path.join(ROOT, "pb-starphase_amd", "csrc", "sp_consensus.hip"), "rb").read()).hexdigest()[:16]
    stale, pmc = [], None
    cfile = os.path.join(COUNTER_DIR, "counters_cons_step.json")
    if os.path.exists(cfile):
        rec = json.load(open(cfile))
        if rec.get("cons_source_sha16") == cons_sha and args.cyp_reads == 2000:
            pmc = rec
        else:
            stale.append("counters_cons_step.json")
    cp = crit["cyp2d6"]
    step_ms, chain_ms, dep_steps = cp["step_kernel_ms"], cp["chain_ms"], cp["dependent_steps"]
    batches = batches_per_step if batches_per_step > 0 else None
    cyp_packed_bytes = float(np.mean([c.ascii_bytes for c in cyp_samples])) / 4.0
    # SURVEY 8(d)'s streaming model for the consensus: every search level reads the packed bases of its reads once and writes one 32-byte record per read
    searches = batches if batches else 3.0                  # (a batch = the searches of one level of the multi-way consensus, all open groups in lockstep)
    algo_bytes = searches * (cyp_packed_bytes + 32.0 * cyp_samples[0].n)
    achieved = (pmc["sq_insts_valu_per_bench_step"] / (step_ms * 1e-3)) if (pmc and step_ms > 0) else None
    # `peak` is the NOMINAL VALU issue rate (256 CUs x 4 SIMDs x one wave-instruction per 2 cycles at 2.4 GHz = 1,229 G wave-instr/s, MI355X_MICROARCH.md); the measured rate of
    # sp_microbench's integer loop (`measured_peak`) is beside it
    roof = {"bound": "valu", "kernel": "cons_step_wide_kernel<8> / cons_step_kernel<8> (one body at 2 / 4 waves per SIMD: batches of up to 320 / more workgroups)" +
                                       ("" if batches_per_step == 0 else "; %.1f batches per step ran as cons_step_persist_kernel<8>, the library's choice" % batches_per_step),
            "kernel_short": "cons_step_wide_kernel<8> (K8 consensus step)" if batches_per_step == 0 else "cons_step_persist_kernel<8> (K8 consensus step)",
            "achieved": achieved, "peak": VALU_PEAK_WAVE_INSTR, "unit": "wave-instr/s",
            "frac": (achieved / VALU_PEAK_WAVE_INSTR) if achieved else None,
            "measured_peak": peaks["valu_int_wave_instr_per_s"], "frac_of_measured_peak": (achieved / peaks["valu_int_wave_instr_per_s"]) if achieved else None,
            "avg_launch_us": _r(1e3 * step_ms / dep_steps, 4) if dep_steps > 0 else None, "avg_launch_us_rocprof": (pmc or {}).get("rocprof_avg_launch_us"),
            "traffic": pmc["hbm_bytes_per_bench_step"] if pmc else None,
            # (PMC passes: the waves resident on average while a step launch is active / the wave slots of the device at the kernel's register budget, and the share of a
            #  resident wave's cycles in which it issues anything -- a launch lasts as long as its slowest wave, most of its waves are done in a sixth of that)
            "occupancy": (pmc or {}).get("occupancy"),
            "per": "bench step (the consensus searches of one 2,000-read CYP2D6 sample: %.0f dependent window / expansion steps%s)" % (dep_steps, "" if batches is None else ", %.1f persistent launches" % batches),
            "step_kernel_ms_per_step": step_ms, "chain_ms_per_step_hip_events": chain_ms, "nominal_peak": VALU_PEAK_WAVE_INSTR,
            "sq_insts_valu_per_step": pmc["sq_insts_valu_per_bench_step"] if pmc else None, "sq_insts_salu_per_step": pmc["sq_insts_salu_per_bench_step"] if pmc else None,
            "sq_insts_lds_per_step": pmc["sq_insts_lds_per_bench_step"] if pmc else None,
            "hbm": {"algorithmic_bytes_per_step": algo_bytes, "achieved_GBs": algo_bytes / (step_ms * 1e-3) / 1e9 if step_ms > 0 else None, "peak_GBs": HBM_PEAK_GBS,
                    "frac": algo_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if step_ms > 0 else None, "measured_hbm_copy_GBs": peaks["hbm_copy_bytes_per_s"] / 1e9,
                    "note": "SURVEY 8(d)'s streaming model applied to the consensus: every search reads its reads' packed bases once and writes a 32-byte record per read"},
            "counters_from": ("profiles/r06/counters_cons_step.json: " + pmc["method"]) if pmc else "no counter file for this kernel source (profiles/run_rocprof.sh makes it)",
            "note": "the step's dominant kernel is a LATENCY chain, not a throughput kernel: a sample's searches are a few hundred dependent steps (critical_path), each the slowest "
                    "wavefront of a launch that fills a fraction of the device -- the fraction of the VALU peak (and of HBM: the reads are 3 MB) says how little of the machine one "
                    "sample's chain can use; what bounds it is per-step latency (critical_path.cyp2d6.per_step_us) and the number of steps"}
    line = {
        "metric": "HiFi reads/sec diplotyped (HLA+CYP2D6)",
        "value": world * reads_per_step * args.steps / dt, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int32", "dtype_note": "2-bit packed bases (u8 in, 16 per dword), int32 wavefront DP, f64 score ratios", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1] + configs[2] as ONE sample per step: HLA-A / -B, %d synthetic HiFi reads vs the bundled IMGT/HLA DB v0.14.1 (18,461 alleles, 11,199 with "
                               "DNA) and CYP2D6, %d targeted reads (39 templates, 393 variants / 520 star alleles), the steps cycling through ALL SIX configs[2] scenarios (%s); a new "
                               "sample's bytes (BAM 4-bit SEQ) uploaded every step under the previous sample's kernels; reads -> diplotypes of both loci"
                               % (samples[0].n, cyp_samples[0].n, ", ".join(c.name for c in cyp_samples)),
                   "cyp2d6_mix": mix,
                   "reads_per_step": reads_per_step, "hla_reads": samples[0].n, "cyp2d6_reads": cyp_samples[0].n, "alleles": len(fx.ids),
                   "parallelism": ("one GPU: " if world == 1 else "%d GPUs, one process each with its own stream of samples (no exchange; `value` = all ranks' reads / the slowest rank's time); per GPU: " % world) +
                                  "the two loci side by side on contexts of their own (host threads, HIP streams): %d HLA lane(s), %d CYP2D6 lane(s) sharing the stream of CYP2D6 samples "
                                  "(`--hla-lanes`, `--cyp-lanes`: that many samples of the locus in flight; a CYP2D6 sample is a latency chain of dependent consensus steps); uploads on copy streams" % (n_hla_lanes, n_cyp_lanes),
                   "cyp2d6_lanes": n_cyp_lanes, "hla_lanes": n_hla_lanes,
                   "cyp2d6_consensus": ("persistent kernels (%s: two launches per batch, hand-overs through memory; SP_BENCH_HEADLINE_PERSISTENT=0 for launch pairs)"
                                        % ("the library's own choice, k8_persistent = 2 (auto), the default of every context" if forced is None else "forced by SP_BENCH_HEADLINE_PERSISTENT=" + forced)
                                        if headline_mode["persistent"] and batches_per_step > 0 else "a launch pair per step" + ("" if headline_mode["fallback"] is None else " (the persistent kernels failed here: %s)" % headline_mode["fallback"]))},
        "roofline": roof,
        "stale_counter_files": stale or None,
        "kernel_ms": {"hla": kernel_ms, "cyp2d6": cyp_kernel_ms}, "host_wall_ms": {"hla": host_ms, "cyp2d6": cyp_host_ms, "lanes_hla_cyp2d6": lane_ms},
        "critical_path": crit,
        "consensus_hla": cons,
        "concordance": {"hla_diplotypes_equal_truth": f"{ok}/{len(genes)} genes", "cyp2d6_call_equals_truth": bool(cyp_ok), "k1_gene_correct": k1_gene_ok,
                        "cyp2d6_call": [cyp_call.hap1.decode(), cyp_call.hap2.decode()]},
        "upload": {"per_step_bytes": int(len(samples[0].payload[0]) + len(cyp_samples[0].payload[0])), "alone": t_up,
                   "note": "inside `value` every step uploads a new sample (4-bit SEQ bytes, sp_seqset_upload_async); `alone` = one synchronous upload of the HLA half"},
        "legs": legs or None,
        "context": dict(ctx.info(), legs_k8_persistent=cyp_persistent()),          # sp_ctx_get_info: the hardware queues the streams of the run were mapped onto (this script exports GPU_MAX_HW_QUEUES=16 before torch initialises HIP)
    }
    if cb is not None:
        # the GPU on exactly the reads the CPU leg saw: the calls of the two have to be the same, for both loci
        cpu_done, cpu_best, cpu_calls = cpu_ref["hla_done"], cpu_ref["hla_best"], cpu_ref["hla_calls"]
        sub = [samples[0].wl.reads[r] for r in cpu_done]
        Rs = ctx.upload(sub)
        o = db.realign_reads(Rs)
        g_calls = db.diplotype_genes(genes, Rs, o)[0]
        gpu_calls = {g: sorted([int(c.allele1), int(c.allele2)]) for g, (c, _a, _b) in enumerate(g_calls)}
        cpu_c = {g: sorted(int(x) for x in cpu_calls[g]) for g in cpu_calls}
        cb["hla"]["diplotypes_cpu"] = {fx.genes[g]: cpu_c[g] for g in cpu_c}
        cb["hla"]["diplotypes_gpu_same_reads"] = {fx.genes[g]: gpu_calls[g] for g in gpu_calls}
        hla_same = all(all(same_allele(fx, a, b) for a, b in zip(cpu_c[g], gpu_calls[g])) for g in cpu_c)
        agree = sum(1 for k, r in enumerate(cpu_done) if cpu_best[r][0] == int(o[k]["best_allele"]))
        gene_agree = sum(1 for k, r in enumerate(cpu_done) if cpu_best[r][0] >= 0 and int(o[k]["gene"]) == int(fx.gene_of[cpu_best[r][0]]))
        cb["hla"]["k1_same_allele_as_gpu"] = f"{agree}/{len(cpu_done)}"
        # the whole realign_record result: status, the segment cut out of the read, its DNA / HPC offsets on the gene's reference (what the consensus stage is fed)
        recs = cpu_ref.get("hla_records", {})
        same_rec = sum(1 for k, r in enumerate(cpu_done) if r in recs and int(recs[r]["status"]) == int(o[k]["status"]) and (int(recs[r]["status"]) != 0 or all(
            int(recs[r][f]) == int(o[k][f]) for f in ("seg_start", "seg_end", "dna_offset", "hpc_offset"))))
        cb["hla"]["k1_records_identical_to_gpu"] = f"{same_rec}/{len(cpu_done)} (status; segment start / end, DNA and HPC offset of every realigned read)"
        cb["hla"]["k1_same_gene_as_gpu"] = f"{gene_agree}/{len(cpu_done)}"
        cyp_same, cyp_sets_gpu = True, []
        for name, creads, cres in cpu_ref["cyp"]:
            Rc = ctx.upload(creads)
            g_cyp, _cons, _labels = cdb.diplotype(Rc)
            got = [g_cyp.hap1.decode(), g_cyp.hap2.decode()]
            cb["cyp2d6"]["scenarios"][name]["call_gpu_same_reads"] = got
            same_here = int(g_cyp.status) == int(cres["status"]) and sorted(got) == sorted([cres.get("hap1", ""), cres.get("hap2", "")])
            cb["cyp2d6"]["scenarios"][name]["identical"] = bool(same_here)
            cyp_same = cyp_same and same_here
            cyp_sets_gpu.append(Rc)
        cb["diplotypes_identical"] = {"hla": bool(hla_same), "cyp2d6": bool(cyp_same)}
        # like for like: the GPU on the same reads, one sample at a time, reads resident (the headline also uploads a new sample per step)
        ctx.synchronize(); t1 = time.perf_counter()
        o2 = db.realign_reads(Rs); db.diplotype_genes(genes, Rs, o2)
        for Rc in cyp_sets_gpu:
            cdb.diplotype(Rc)
        ctx.synchronize(); d_same = time.perf_counter() - t1
        n_same = len(sub) + sum(len(c[1]) for c in cpu_ref["cyp"])
        cb["gpu_same_reads_one_after_the_other"] = {"value": n_same / d_same, "unit": "reads/s", "seconds": d_same}
        cb["gpu_over_cpu"] = {"one_socket": cb["gpu_same_reads_one_after_the_other"]["value"] / cb["value"],
                              "one_thread": cb["gpu_same_reads_one_after_the_other"]["value"] / cb["single_thread_value"],
                              "note": "same reads, both loci, the GPU running the loci and scenarios one after the other as the CPU leg does; a ratio to a scalar restatement, not to minimap2's SSE build"}
        for Rc in cyp_sets_gpu:
            Rc.close()
        Rs.close()
        line["cpu_baseline"] = cb
    else:
        line["cpu_baseline"] = None
    if rank == 0:
        compact = {k: line[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
        compact["value"], compact["ms_per_step"] = _r(line["value"], 6), _r(line["ms_per_step"], 6)
        compact["config"] = {"workload": "BASELINE configs[1] + configs[2] as ONE sample per step: HLA-A/-B %d HiFi reads vs IMGT/HLA v0.14.1 (11,199 DNA alleles) + CYP2D6 %d targeted reads "
                                         "(39 templates), the six configs[2] scenarios in turn; a new sample uploaded every step; reads -> diplotypes" % (samples[0].n, cyp_samples[0].n),
                             "reads_per_step": reads_per_step, "hla_reads": samples[0].n, "cyp2d6_reads": cyp_samples[0].n, "cyp2d6_lanes": n_cyp_lanes, "hla_lanes": n_hla_lanes,
                             "cyp2d6_consensus": "persistent kernels" if (headline_mode["persistent"] and batches_per_step > 0) else "launch pairs"}
        rf = roof
        compact["roofline"] = {"bound": rf["bound"], "kernel": rf["kernel_short"], "achieved": _r(rf["achieved"]), "peak": rf["peak"], "unit": rf["unit"], "frac": _r(rf["frac"]),
                               "traffic": rf["traffic"], "per": "bench step", "measured_peak": _r(rf["measured_peak"]), "frac_of_measured_peak": _r(rf["frac_of_measured_peak"]),
                               "kernel_ms_per_step_device_clock": _r(rf["step_kernel_ms_per_step"]), "kernel_avg_launch_us": rf.get("avg_launch_us"),
                               "kernel_avg_launch_us_rocprof": rf.get("avg_launch_us_rocprof"),
                               "mean_resident_waves_over_slots": _r((rf.get("occupancy") or {}).get("mean_resident_over_slots")),
                               "hbm": {"algorithmic_bytes_per_step": _r(rf["hbm"]["algorithmic_bytes_per_step"]), "achieved_GBs": _r(rf["hbm"]["achieved_GBs"]), "peak_GBs": HBM_PEAK_GBS,
                                       "frac": _r(rf["hbm"]["frac"])},
                               "counters_from": ("profiles/%s/counters_cons_step.json" % os.path.basename(COUNTER_DIR)) if rf["traffic"] is not None else None}
        cbf = line["cpu_baseline"]
        if cbf is not None:
            compact["cpu_baseline"] = {"value": _r(cbf["value"]), "unit": cbf["unit"], "cores": cbf["cores"], "kind": cbf["kind"], "sample": cbf["sample_short"],
                                       "single_thread_value": _r(cbf["single_thread_value"]), "physical_cores": cbf.get("physical_cores"), "wall_s": _r(cbf["wall_s"]),
                                       "diplotypes_identical": cbf.get("diplotypes_identical"), "k1_records_identical_to_gpu": cbf["hla"].get("k1_records_identical_to_gpu", "").split(" ")[0],
                                       "k1_same_allele_as_gpu": cbf["hla"].get("k1_same_allele_as_gpu"),
                                       "gpu_same_reads_value": _r(cbf.get("gpu_same_reads_one_after_the_other", {}).get("value"))}
        else:
            compact["cpu_baseline"] = None
        compact["concordance"] = {"hla_diplotypes_equal_truth": line["concordance"]["hla_diplotypes_equal_truth"], "cyp2d6_calls_equal_truth": "%d/%d" % (
            sum(m["calls_equal_truth"] for m in mix.values()), sum(m["steps"] for m in mix.values()))}
        coh = (legs or {}).get("cohort") or {}
        if "samples_per_s" in coh:
            compact["cohort"] = {"samples_per_s": _r(coh["samples_per_s"]), "samples": coh["samples"], "calls_equal_truth": coh["calls_equal_truth"],
                                 "share_rate_over_cohort_rate": {k: _r(v["samples_per_s"] / coh["samples_per_s"], 3) for k, v in (coh.get("by_share_size") or {}).items()}}
        cpc = crit["cyp2d6"]
        compact["summary"] = {"cyp2d6_chain": {"dependent_steps": _r(cpc["dependent_steps"]), "chain_ms": _r(cpc["chain_ms"]), "per_step_us": {k: _r(v, 3) for k, v in cpc.get("per_step_us", {}).items()},
                                               "kernels_running_fraction": _r(cpc.get("kernels_running_fraction"), 3)},
                              "one_lane_reads_per_s": _r((legs.get("cyp2d6_lanes") or {}).get("1", {}).get("value")) if legs else None,
                              "hla_resident_reads_per_s": _r((legs.get("hla_resident") or {}).get("value")) if legs else None,
                              "cyp2d6_scenarios_ms": {k: _r(v["ms"], 3) for k, v in ((legs.get("cyp2d6") or {}).get("scenarios") or {}).items()} if legs else None}
        emit(line, compact, args.full_out)
    if world > 1:
        dist.barrier()
        if group is not None and hasattr(group, "close") and not isinstance(group, shard.TorchGroup):
            group.close()
        dist.destroy_process_group()
    if stale and rank == 0:                                                 # (the blocks they feed were left out of the line above)
        print("bench.py: the counter files " + ", ".join(stale) + " were collected on other kernel sources: re-run profiles/run_rocprof.sh and copy counters_cons_step.json into profiles/r06", file=sys.stderr)


if __name__ == "__main__":
    main()
