#!/bin/bash
mkdir -p gpurun_out/x
timeout -k 10 500 python -m pytest tests/test_gpu_tracking.py -x -q -m gpu > gpurun_out/x/tests.log 2>&1 || { tail -30 gpurun_out/x/tests.log; exit 1; }
tail -2 gpurun_out/x/tests.log
PCS_BENCH_TRACE=1 timeout -k 10 500 python bench.py --batch 12 --steps 20 --warmup 1 --no-cpu-baseline --no-stages > gpurun_out/x/b12long.json 2> gpurun_out/x/b12long.err
echo rc=$?
grep "^step" gpurun_out/x/b12long.err | awk '{print $3}' | sort -n | tr '\n' ' '
