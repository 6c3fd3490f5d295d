timeout 1500 python profiles/scripts/pipeline_fuzz.py 40 12 404 2>&1 | tail -4
timeout 600 python profiles/scripts/thread_stress.py 4 60 2>&1 | tail -3
timeout 600 python profiles/scripts/k8fuzz.py 2000 2>&1 | tail -2
timeout 600 python profiles/scripts/k8fuzz.py 1000 2>&1 | tail -2
