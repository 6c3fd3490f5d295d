for wm in default 0; do for so in 0 1 2 3; do
  echo "== side orders $so, SP_K8_WIDE_MAX=$wm"
  if [ $wm = default ]; then SP_K8_SIDE_ORDERS=$so python profiles/scripts/k8_side_orders.py 2000 '*1/*2' '*4+*68/*1' 2>&1 | tail -4
  else SP_K8_WIDE_MAX=$wm SP_K8_SIDE_ORDERS=$so python profiles/scripts/k8_side_orders.py 2000 '*1/*2' '*4+*68/*1' 2>&1 | tail -4; fi
done; done
