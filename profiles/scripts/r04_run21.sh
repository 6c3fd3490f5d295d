timeout 900 python -m pytest tests/test_gpu_cyp.py tests/test_gpu_cyp_real.py -x -q -k "not stated_size" 2>&1 | tail -2
for s in 0 1; do python profiles/scripts/cyp_kernels.py $s | grep -E "total|k9_graph|align_trace|merge"; done
