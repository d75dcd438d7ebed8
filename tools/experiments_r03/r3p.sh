mkdir -p gpurun_out/r3p
./tools/xcd_probe > gpurun_out/r3p/xcd_probe.txt 2>&1; echo "probe rc=$?"; cat gpurun_out/r3p/xcd_probe.txt
PCS_ICP_XCD=1 timeout -k 10 400 python -m pytest tests/test_gpu_tracking.py tests/test_gpu_graph_icp.py -x -q -m gpu > gpurun_out/r3p/tests.log 2>&1; rc=$?
echo "tests(xcd) rc=$rc"; tail -3 gpurun_out/r3p/tests.log
[ $rc -ne 0 ] && exit $rc
run() { # name, extra args..., env via ENVV
  name=$1; shift
  env $ENVV PCS_BENCH_TRACE=1 timeout -k 10 240 python bench.py --warmup 2 --no-cpu-baseline --no-stages "$@" > gpurun_out/r3p/$name.json 2> gpurun_out/r3p/$name.err; rc=$?
  echo "$name rc=$rc : $(python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r3p/$name.json').read().strip().splitlines()[-1])
    print('value', d['value'], 'ms', d['ms_per_step'], 'steps', d.get('step_ms'), 'roof', d['roofline'].get('achieved'), d['roofline'].get('kernel_us'))
except Exception as e: print('no line', e)
PY
)"
  [ $rc -ne 0 ] && exit $rc
}
ENVV="PCS_ICP_XCD=0" run b12_x0 --steps 8
ENVV="PCS_ICP_XCD=1" run b12_x1 --steps 8
ENVV="PCS_ICP_XCD=0" run b1_x0 --batch 1 --steps 20
ENVV="PCS_ICP_XCD=1" run b1_x1 --batch 1 --steps 20
ENVV="PCS_ICP_XCD=1" run b12_x1b --steps 8
ENVV="PCS_ICP_XCD=0" run b12_x0b --steps 8
true
