mkdir -p gpurun_out/r3ag
timeout -k 10 500 python -m pytest tests/test_gpu_ground.py tests/test_gpu_loader.py -x -q -m gpu > gpurun_out/r3ag/tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -15 gpurun_out/r3ag/tests.log
