# consensus parity (oracle bit-exact incl. nodes expanded), the fuzz, then the six configs[2] scenarios' times with 0 / 3 side orders
set -u
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_consensus.py tests/test_gpu_cyp.py tests/test_gpu_cyp_real.py tests/test_gpu_hla_pipeline.py -x -q -m gpu 2>&1 | tail -5
timeout 900 python profiles/scripts/k8fuzz.py 2>&1 | tail -3
for so in 0 1 3; do
  echo "== SP_K8_SIDE_ORDERS=$so"
  SP_K8_SIDE_ORDERS=$so python profiles/scripts/k8_side_orders.py 2000 2>&1 | tail -8
done
