timeout 600 python profiles/scripts/af_windows_check.py 2>&1 | tail -40
timeout 1200 python -m pytest tests/test_gpu_affine.py tests/test_gpu_hla.py tests/test_gpu_concordance.py tests/test_gpu_panel.py -x -q 2>&1 | tail -5
