# more of the differential fuzzers on the final sources (second batch of seeds)
S3=$(python -c "print(','.join(str(x) for x in list(range(101,161))+list(range(1081,1121))+list(range(2081,2121))))")
timeout 2400 python profiles/scripts/k8fuzz.py $S3 2>&1 | tail -3
timeout 2400 python profiles/scripts/pipeline_fuzz.py 30 30 707 2>&1 | tail -2
timeout 900 python profiles/scripts/k6fuzz.py 3000 606 2>&1 | tail -2
