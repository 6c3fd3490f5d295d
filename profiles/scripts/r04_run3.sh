timeout 900 python -m pytest tests/test_gpu_hla.py tests/test_gpu_affine.py tests/test_gpu_concordance.py -q -s 2>&1 | grep -E "K1 |passed|failed|Error|assert" | head -20
