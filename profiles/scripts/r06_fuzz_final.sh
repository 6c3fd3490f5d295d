# the consensus fuzzer on the final code: default options, and the settings that speculate most (three side orders, branching windows wherever they are foreseen)
mkdir -p gpurun_out/r06k
S1=$(python -c "print(','.join(str(x) for x in list(range(41,81))+list(range(1041,1061))+list(range(2041,2061))))")
S2=$(python -c "print(','.join(str(x) for x in list(range(81,101))+list(range(1061,1081))+list(range(2061,2081))))")
timeout 1500 python profiles/scripts/k8fuzz.py $S1 2>&1 | tail -3
SP_K8_SIDE_ORDERS=3 SP_K8_COMPOUND=3 timeout 1500 python profiles/scripts/k8fuzz.py $S2 2>&1 | tail -3
SP_K8_SIDE_ORDERS=2 SP_K8_COMPOUND=50 SP_K8_PERSISTENT=0 timeout 1500 python profiles/scripts/k8fuzz.py $S2 2>&1 | tail -3
timeout 1200 python profiles/scripts/pipeline_fuzz.py 16 6 606 2>&1 | tail -3
