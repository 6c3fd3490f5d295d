"""bench.py's N > 1 path on a one-GPU box: two ranks (child processes) sharing device 0, the gather over gloo instead of RCCL
(SP_BENCH_BACKEND=gloo; RCCL refuses two ranks on one device).  Checks the contract of the JSON line, not the speed."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(extra, env=None):
    e = dict(os.environ, **(env or {}))
    e.pop("RANK", None); e.pop("WORLD_SIZE", None); e.pop("LOCAL_RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=e, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_over_gloo_on_one_device():
    one = run_bench(["--gpus", "1", "--steps", "2", "--warmup", "1", "--reads", "2000", "--no-cpu-baseline", "--no-extra-legs"])
    two = run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--reads", "2000", "--no-cpu-baseline", "--no-extra-legs"],
                    env={"SP_BENCH_BACKEND": "gloo"})
    for line, n in ((one, 1), (two, 2)):
        assert line["n_gpus"] == n and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak"
        assert line["metric"] == one["metric"] and line["unit"] == one["unit"] and line["higher_is_better"] is True
        assert line["value"] > 0 and line["ms_per_step"] > 0 and line["vs_baseline"] is None
        assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    # every rank types its own 2,000-read sample: the job's reads double, both samples are called correctly
    assert two["config"]["reads_per_gpu"] == one["config"]["reads_per_gpu"] == 2000
    assert two["concordance"]["diplotypes_equal_truth"] == one["concordance"]["diplotypes_equal_truth"] == "2/2 genes"
