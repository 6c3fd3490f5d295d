SP_K8_COMPOUND=1 SP_K8_SIDE_ORDERS=1 timeout 900 python -m pytest tests/test_gpu_consensus.py tests/test_gpu_cyp_real.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python profiles/scripts/k8fuzz.py 2>&1 | tail -2
for cp in 15 25 40; do for so in 0 1; do
  echo "== compound $cp side orders $so"
  SP_K8_COMPOUND=$cp SP_K8_SIDE_ORDERS=$so python profiles/scripts/k8_side_orders.py 2000 '*1/*2' '*4+*68/*1' HLA 2>&1 | grep -v path | tail -3
done; done
