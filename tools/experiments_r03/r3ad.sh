mkdir -p gpurun_out/r3ad
run() { name=$1; st=$2; shift; shift
  env "$@" PCS_BENCH_TRACE=1 timeout -k 10 200 python bench.py --warmup 2 --no-cpu-baseline --no-stages --steps $st > gpurun_out/r3ad/$name.json 2> gpurun_out/r3ad/$name.err; rc=$?
  echo "$name rc=$rc $(python -c "
import json
d=json.loads(open('gpurun_out/r3ad/$name.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms'], d['config'].get('one_sequence_alone_ms'))" 2>&1 | tail -1)"
}
run q4 6
run q5 10 GPU_MAX_HW_QUEUES=5
run q6 10 GPU_MAX_HW_QUEUES=6
run q3 6 GPU_MAX_HW_QUEUES=3
true
