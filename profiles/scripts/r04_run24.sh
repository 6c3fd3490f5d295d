timeout 600 python -m pytest tests/test_gpu_upload.py tests/test_gpu_sample.py -x -q 2>&1 | tail -2
python bench.py --no-cpu-baseline --no-extra-legs > gpurun_out/r04_pk.json 2> gpurun_out/r04_pk.err
python -c "
import json;d=json.loads(open('gpurun_out/r04_pk.json').read().strip().splitlines()[-1]);print(round(d['value']), round(d['ms_per_step'],2), d['upload'])"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
export GPU_MAX_HW_QUEUES=16
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pk -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs > gpurun_out/prof_pk.log 2>&1
grep -h "sp_pack4" gpurun_out/prof_pk/*/*kernel_stats.csv | cut -c1-200
