mkdir -p gpurun_out/r06d
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r06d/pytest_gpu.txt; cat gpurun_out/r06d/pytest_gpu.txt
timeout 600 python -m pytest tests/test_gpu_concordance.py -m gpu -q -s -k "score_read_numbers" 2>&1 | grep "K2 " > gpurun_out/r06d/k2_numbers.txt; cat gpurun_out/r06d/k2_numbers.txt
python profiles/scripts/k8_side_orders.py 2000 > gpurun_out/r06d/k8_side_orders_default.txt 2>&1; cat gpurun_out/r06d/k8_side_orders_default.txt
python profiles/scripts/k3_residue_probe.py '*5/*1' '*4+*68/*1' > gpurun_out/r06d/k3_residue.txt 2>&1; head -c 6000 gpurun_out/r06d/k3_residue.txt
