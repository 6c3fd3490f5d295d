"""bench.py's N > 1 path on a one-GPU box: two ranks (child processes) sharing device 0, the gather over gloo instead of RCCL
(SP_BENCH_BACKEND=gloo; RCCL refuses two ranks on one device).  Checks the contract of the JSON line, not the speed."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def run_bench(extra, env=None, tmp=None):
    """-> the whole record (the side file `--full-out` names); the ONE line on stdout is the compact one the driver parses: under 4 KB, every key of the contract"""
    import tempfile
    e = dict(os.environ, **(env or {}))
    e.pop("RANK", None); e.pop("WORLD_SIZE", None); e.pop("LOCAL_RANK", None)
    with tempfile.TemporaryDirectory() as d:
        side = os.path.join(d, "full.json")
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--full-out", side] + extra, env=e, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        printed = [l for l in out.stdout.splitlines() if l.strip()]
        lines = [l for l in printed if l.startswith("{")]                  # (gloo announces its peers on stdout; the JSON line is the last line and the only one of its kind)
        assert len(lines) == 1 and printed[-1] == lines[0], out.stdout[-2000:]
        assert len(lines[0]) < 4096, len(lines[0])
        line = json.loads(lines[0])
        assert set(line) >= set(REQUIRED), sorted(set(REQUIRED) - set(line))
        assert "workload" in line["config"] and "model" not in line["config"]
        full = json.load(open(side))
    for k in ("metric", "unit", "n_gpus", "steps", "warmup", "scaling", "higher_is_better", "vs_baseline", "dtype", "data"):
        assert line[k] == full[k], k
    assert abs(line["value"] - full["value"]) <= 1e-5 * full["value"] and abs(line["ms_per_step"] - full["ms_per_step"]) <= 1e-5 * full["ms_per_step"]
    full["_line"] = line
    return full


def test_sample_line_and_two_rank_cohort_over_gloo():
    """N = 1: one sample with both loci, a new upload every step.  N = 2: the cohort sharded by sample (two ranks sharing device 0, the gather over
    gloo: RCCL refuses two ranks on one device; the RCCL gather of one rank runs in tests/test_gpu_upload.py)."""
    one = run_bench(["--gpus", "1", "--steps", "2", "--warmup", "1", "--reads", "2000", "--cyp-reads", "400", "--no-cpu-baseline", "--no-extra-legs"])
    assert one["n_gpus"] == 1 and one["steps"] == 2 and one["warmup"] == 1 and one["scaling"] == "weak"
    assert one["metric"] == "HiFi reads/sec diplotyped (HLA+CYP2D6)" and one["unit"] == "reads/s" and one["higher_is_better"] is True
    assert one["value"] > 0 and one["ms_per_step"] > 0 and one["vs_baseline"] is None
    assert set(one["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert set(one["_line"]["roofline"]) >= {"bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "hbm"} and one["_line"]["roofline"]["peak"] == 256 * 4 * 2.4e9 / 2
    assert one["_line"]["config"]["reads_per_step"] == one["config"]["reads_per_step"] and one["_line"]["concordance"]["cyp2d6_calls_equal_truth"] == "2/2"
    assert one["config"]["hla_reads"] == 2000 and 380 <= one["config"]["cyp2d6_reads"] <= 420
    assert one["concordance"]["hla_diplotypes_equal_truth"] == "2/2 genes" and one["concordance"]["cyp2d6_call_equals_truth"] is True
    assert one["upload"]["per_step_bytes"] > 5_000_000 and one["upload"]["alone"]["bam4"]["GBps"] > 0
    for n in (1, 2):
        line = run_bench(["--gpus", str(n), "--steps", "1", "--warmup", "1", "--cohort-samples", "8", "--reads", "1500", "--cyp-reads", "300"] + (["--workload", "cohort"] if n == 1 else []),
                         env={"SP_BENCH_BACKEND": "gloo"})              # (N > 1: the cohort is the default workload)
        assert line["n_gpus"] == n and line["scaling"] == "strong" and line["metric"] == one["metric"] and line["value"] > 0
        c = line["cohort"]
        assert c["samples"] == 8 and c["records_gathered_per_pass"] == 8 * (2 + 1 + 18)              # HLA-A, HLA-B, CYP2D6, 18 variant genes per sample
        assert c["calls_equal_truth"]["hla"] == "16/16" and c["calls_equal_truth"]["cyp2d6"] == "8/8"
        assert c["calls_equal_truth"]["variant_genes_truth_among_reported"] == "144/144"
        assert c["samples_per_s"] > 0 and c["rank0_host_seconds_per_pass"]["cyp2d6"] > 0
        if n == 2:
            # the second blocks of the N > 1 line: the same cohort on rank 0 alone, and the ranks' independent streams of samples (weak scaling, nothing exchanged)
            assert line["second_blocks_error"] is None, line["second_blocks_error"]
            assert line["one_gpu_same_cohort"]["value"] > 0 and line["one_gpu_same_cohort"]["value_over_n_times_this"] > 0
            st = line["independent_streams"]
            assert st["scaling"] == "weak" and st["value"] > 0 and st["cyp2d6_calls_equal_truth_rank0"].split("/")[0] == st["cyp2d6_calls_equal_truth_rank0"].split("/")[1]
    # shares that differ by one sample (7 samples over 2 ranks: 4 + 3): every record arrives once, every call equals the truth
    line = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "1", "--cohort-samples", "7", "--no-extra-legs"], env={"SP_BENCH_BACKEND": "gloo"})
    c = line["cohort"]
    assert c["samples"] == 7 and c["records_gathered_per_pass"] == 7 * (2 + 1 + 18)
    assert c["calls_equal_truth"]["hla"] == "14/14" and c["calls_equal_truth"]["cyp2d6"] == "7/7"
    assert line["group_fallback"] is None and line["gather_via"] == "torch.distributed"          # (gloo: the torch group is the path asked for, not a fall-back)


def test_a_group_that_never_comes_up_falls_back_to_the_torch_group():
    """`bench.py --gpus N` brings the ranks' group up -- and tries it with one small gather -- on a helper thread with a time limit before anything is timed; a rank whose
    group never returns (injected here: SP_BENCH_INJECT_GROUP_HANG) makes every rank gather through torch.distributed, and the line says so"""
    line = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "1", "--cohort-samples", "4", "--no-extra-legs"],
                     env={"SP_BENCH_BACKEND": "gloo", "SP_BENCH_INJECT_GROUP_HANG": "1", "SP_BENCH_GROUP_TIMEOUT_S": "2"})
    assert "time limit" in line["group_fallback"] and line["gather_via"] == "torch.distributed"
    c = line["cohort"]
    assert c["samples"] == 4 and c["records_gathered_per_pass"] == 4 * (2 + 1 + 18)
    assert c["calls_equal_truth"]["hla"] == "8/8" and c["calls_equal_truth"]["cyp2d6"] == "4/4"


@pytest.mark.parametrize("inject", ["1:before", "0:after", "1:error"])
def test_one_rank_whose_group_hangs_or_fails_takes_every_rank_to_the_torch_group(inject):
    """only ONE rank's group bring-up hangs (before or after the group is made) or raises: the id travels through the rendezvous store and the agreement is the first collective
    every rank issues, so the healthy rank is not left waiting in a collective the other never joins; every rank gathers through torch.distributed and the line is complete"""
    line = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "1", "--cohort-samples", "4", "--no-extra-legs"],
                     env={"SP_BENCH_BACKEND": "gloo", "SP_BENCH_INJECT_GROUP_HANG": inject, "SP_BENCH_GROUP_TIMEOUT_S": "2"})
    assert line["group_fallback"] and line["gather_via"] == "torch.distributed"
    c = line["cohort"]
    assert c["samples"] == 4 and c["records_gathered_per_pass"] == 4 * (2 + 1 + 18)
    assert c["calls_equal_truth"]["hla"] == "8/8" and c["calls_equal_truth"]["cyp2d6"] == "4/4"


def test_eight_rank_cohort_of_sixteen_samples_over_gloo():
    """the round-end scaling run's shape on one device: `bench.py --gpus 8` with a 16-sample cohort, two samples per rank, the records of all eight ranks in rank 0's table"""
    line = run_bench(["--gpus", "8", "--steps", "1", "--warmup", "1", "--cohort-samples", "16", "--no-extra-legs"], env={"SP_BENCH_BACKEND": "gloo"})
    assert line["n_gpus"] == 8 and line["scaling"] == "strong" and line["value"] > 0
    c = line["cohort"]
    assert c["samples"] == 16 and c["records_gathered_per_pass"] == 16 * (2 + 1 + 18)
    assert c["calls_equal_truth"]["hla"] == "32/32" and c["calls_equal_truth"]["cyp2d6"] == "16/16"
    assert c["calls_equal_truth"]["variant_genes_truth_among_reported"] == "288/288"


def test_two_ranks_each_with_its_own_stream_of_samples():
    """`bench.py --gpus 2 --workload sample`: each rank runs the headline's stream of samples -- its own samples, both loci, a new upload every step --, the
    ranks meet at the barriers around the timed steps, `value` is the sum over the ranks and rank 0 prints the one line.  Two ranks share device 0 here (gloo for the
    barrier and the maximum; launch pairs: two processes' persistent batches would wait for each other's CUs on one device)."""
    two = run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--reads", "1500", "--cyp-reads", "300", "--no-cpu-baseline", "--workload", "sample"],
                    env={"SP_BENCH_BACKEND": "gloo", "SP_BENCH_HEADLINE_PERSISTENT": "0"})
    assert two["n_gpus"] == 2 and two["steps"] == 2 and two["scaling"] == "weak" and two["metric"] == "HiFi reads/sec diplotyped (HLA+CYP2D6)"
    assert two["config"]["parallelism"].startswith("2 GPUs") and two["config"]["cyp2d6_consensus"].startswith("a launch pair per step")
    per_rank_reads = two["config"]["reads_per_step"] * two["steps"]
    assert abs(two["value"] - 2 * per_rank_reads / (two["ms_per_step"] * 1e-3 * two["steps"])) < 1e-6 * two["value"]      # all ranks' reads / the slowest rank's time
    assert not two["legs"] and two["cpu_baseline"] is None
    assert two["concordance"]["hla_diplotypes_equal_truth"] == "2/2 genes" and two["concordance"]["cyp2d6_call_equals_truth"] is True


def test_the_headline_falls_back_to_launch_pairs_when_the_persistent_mode_fails():
    """The headline's CYP2D6 context runs its consensus as persistent kernels; if the library reports that a batch's kernels could not run side by side (its time-out, an
    error) the bench starts the region over with a launch pair per step and says so in the line.  The failure is injected here (SP_BENCH_INJECT_FAILURE)."""
    line = run_bench(["--gpus", "1", "--steps", "2", "--warmup", "1", "--reads", "1500", "--cyp-reads", "300", "--no-cpu-baseline", "--no-extra-legs"],
                     env={"SP_BENCH_INJECT_FAILURE": "1"})
    assert line["config"]["cyp2d6_consensus"].startswith("a launch pair per step (the persistent kernels failed here: injected failure")
    assert line["value"] > 0 and line["concordance"]["cyp2d6_call_equals_truth"] is True and line["concordance"]["hla_diplotypes_equal_truth"] == "2/2 genes"
    ok = run_bench(["--gpus", "1", "--steps", "2", "--warmup", "1", "--reads", "1500", "--cyp-reads", "300", "--no-cpu-baseline", "--no-extra-legs"])
    assert ok["config"]["cyp2d6_consensus"].startswith("persistent kernels") and ok["critical_path"]["cyp2d6"]["launches_per_step"] == 0
