# what the agent-scope fences of the persistent consensus kernels cost: measurement builds without some of them (results not to be trusted)
for v in 15 14 10 0; do
  L=$PWD/build/variants/lib_pf$v.so; [ $v = 15 ] && L=$PWD/pb-starphase_amd/libstarphase_hip.so
  echo "== fences $v"
  SP_LIB_PATH=$L timeout 120 python profiles/scripts/k8persist_dbg3.py "*1/*2" 2>&1 | grep -E "persistent|classic|rror" | tail -3
done
