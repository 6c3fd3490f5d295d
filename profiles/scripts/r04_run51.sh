# whole GPU suite on the committed state, with the process's exit code
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/suite.log 2>&1; echo "pytest exit code $?"
grep -E "passed|failed|error" gpurun_out/suite.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit code $?"
python bench.py --no-cpu-baseline --no-extra-legs > gpurun_out/b.json 2> gpurun_out/b.err; echo "bench exit code $?"
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo "full bench exit code $?"
