mkdir -p gpurun_out/r3o
timeout -k 10 600 python -m pytest tests/test_gpu_tracking.py tests/test_gpu_graph_icp.py tests/test_gpu_full_size.py -x -q -m gpu > gpurun_out/r3o/tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -5 gpurun_out/r3o/tests.log
[ $rc -ne 0 ] && exit $rc
run() { # name, env...
  name=$1; shift
  env "$@" PCS_BENCH_TRACE=1 timeout -k 10 240 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-stages > gpurun_out/r3o/$name.json 2> gpurun_out/r3o/$name.err; rc=$?
  echo "$name rc=$rc : $(python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r3o/$name.json').read().strip().splitlines()[-1])
    print('value', d['value'], 'ms', d['ms_per_step'], 'steps', d.get('step_ms'))
except Exception as e: print('no line', e)
PY
)"
  [ $rc -ne 0 ] && exit $rc
}
run both
run none PCS_ICP_LDS_STEPS=0 PCS_ICP_SEED_FIRST=0
run ldsonly PCS_ICP_SEED_FIRST=0
run seedonly PCS_ICP_LDS_STEPS=0
run both2
run none2 PCS_ICP_LDS_STEPS=0 PCS_ICP_SEED_FIRST=0
