mkdir -p gpurun_out/r3ah
run() { name=$1; shift
  PCS_BENCH_TRACE=1 timeout -k 10 400 python bench.py --warmup 2 --no-cpu-baseline --no-stages "$@" > gpurun_out/r3ah/$name.json 2> gpurun_out/r3ah/$name.err; rc=$?
  echo "$name rc=$rc $(python -c "
import json
d=json.loads(open('gpurun_out/r3ah/$name.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['step_ms'])" 2>&1 | tail -1)"
  [ $rc -ne 0 ] && exit $rc
}
run b12 --steps 6
run b18 --steps 4 --batch 18 --inflight 18
run b24 --steps 3 --batch 24 --inflight 24
run b24i12 --steps 3 --batch 24 --inflight 12
true
