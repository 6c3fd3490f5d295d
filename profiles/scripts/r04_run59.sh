# the round's last checks: whole GPU suite, then the whole bench line (cpu_baseline included)
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -5
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo "rc $?"; tail -2 gpurun_out/bench_final.err | cut -c1-300
