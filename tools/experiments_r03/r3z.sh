mkdir -p gpurun_out/r3z
timeout -k 10 500 python -m pytest tests/test_gpu_tracking.py tests/test_gpu_voxel_conv.py -x -q -m gpu > gpurun_out/r3z/tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -3 gpurun_out/r3z/tests.log
[ $rc -ne 0 ] && exit $rc
run() { name=$1; st=$2; shift; shift
  env "$@" PCS_BENCH_TRACE=1 timeout -k 10 300 python bench.py --warmup 2 --no-cpu-baseline --no-stages --steps $st > gpurun_out/r3z/$name.json 2> gpurun_out/r3z/$name.err; rc=$?
  echo "$name rc=$rc $(python -c "
import json
d=json.loads(open('gpurun_out/r3z/$name.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms'], d['config'].get('one_sequence_alone_ms'))")"
  [ $rc -ne 0 ] && exit $rc
}
run e4 6 PCS_TRACK_ENGINES=4
run e3 40 PCS_TRACK_ENGINES=3
true
