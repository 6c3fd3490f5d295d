timeout 900 python -m pytest tests/test_gpu_cyp_real.py -x -q 2>&1 | grep -E "Error|error|passed|failed" | head -20
