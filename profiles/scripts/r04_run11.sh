timeout 900 python -m pytest tests/test_gpu_consensus.py tests/test_gpu_cyp_real.py tests/test_gpu_cyp.py tests/test_gpu_hla_pipeline.py -x -q 2>&1 | tail -3
timeout 900 python profiles/scripts/k8fuzz.py 2>&1 | tail -2
timeout 600 python profiles/scripts/k8persist_dbg3.py "*1/*2" "*4+*68/*1" 2>&1 | grep -E "classic"
bash profiles/scripts/k8_cyp_timing.sh 0
