mkdir -p gpurun_out/r3ac
run() { name=$1; shift
  env "$@" PCS_TICK_PROFILE=1 timeout -k 10 240 python bench.py --warmup 2 --no-cpu-baseline --no-stages --steps 6 > gpurun_out/r3ac/$name.json 2> gpurun_out/r3ac/$name.err; rc=$?
  echo "$name rc=$rc $(python -c "
import json
d=json.loads(open('gpurun_out/r3ac/$name.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms']['median'], d['config'].get('one_sequence_alone_ms'))") $(grep 'tick profile' gpurun_out/r3ac/$name.err | cut -c15-)"
  [ $rc -ne 0 ] && exit $rc
}
run e3_own PCS_TRACK_ENGINES=3
run e2_pool2 PCS_TRACK_ENGINES=2 PCS_WORKER_STREAM_POOL=2
run e3_pool1 PCS_TRACK_ENGINES=3 PCS_WORKER_STREAM_POOL=1
run e2_pool1 PCS_TRACK_ENGINES=2 PCS_WORKER_STREAM_POOL=1
run e1_pool3 PCS_TRACK_ENGINES=1 PCS_WORKER_STREAM_POOL=3
run e3_pool2 PCS_TRACK_ENGINES=3 PCS_WORKER_STREAM_POOL=2
true
