mkdir -p gpurun_out/r3w
run() { name=$1; shift
  env "$@" PCS_TICK_PROFILE=1 timeout -k 10 240 python bench.py --warmup 2 --no-cpu-baseline --no-stages --steps 6 > gpurun_out/r3w/$name.json 2> gpurun_out/r3w/$name.err; rc=$?
  echo "$name rc=$rc $(python -c "
import json
d=json.loads(open('gpurun_out/r3w/$name.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms']['median'], d['config'].get('one_sequence_alone_ms'))") $(grep 'tick profile' gpurun_out/r3w/$name.err | cut -c15-)"
  [ $rc -ne 0 ] && exit $rc
}
run e2c8 PCS_TRACK_ENGINES=2 PCS_TRACK_CHUNK=8
run e2c6 PCS_TRACK_ENGINES=2 PCS_TRACK_CHUNK=6
run e2c12 PCS_TRACK_ENGINES=2 PCS_TRACK_CHUNK=12
run e2c16 PCS_TRACK_ENGINES=2 PCS_TRACK_CHUNK=16
run e2c8xcd PCS_TRACK_ENGINES=2 PCS_ICP_XCD=1
true
