python -m pytest tests/test_gpu_upload.py -x -q > gpurun_out/up1.log 2>&1; echo "upload tests rc $?"; tail -3 gpurun_out/up1.log | cut -c1-200
python -m pytest tests/test_gpu_upload.py -x -q -k "not rccl" > gpurun_out/up2.log 2>&1; echo "upload tests without the rccl one rc $?"; tail -2 gpurun_out/up2.log | cut -c1-200
python -m pytest tests/test_gpu_upload.py -x -q -k "not freed_sets" > gpurun_out/up3.log 2>&1; echo "upload tests without the new one rc $?"; tail -2 gpurun_out/up3.log | cut -c1-200
python -m pytest tests/test_gpu_align.py -x -q > gpurun_out/up4.log 2>&1; echo "align tests rc $?"; tail -2 gpurun_out/up4.log | cut -c1-200
