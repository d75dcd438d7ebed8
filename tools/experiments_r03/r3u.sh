mkdir -p gpurun_out/r3u
run() { name=$1; shift
  env "$@" PCS_TICK_PROFILE=1 timeout -k 10 240 python bench.py --warmup 2 --no-cpu-baseline --no-stages --steps 6 > gpurun_out/r3u/$name.json 2> gpurun_out/r3u/$name.err; rc=$?
  echo "$name rc=$rc $(python -c "
import json
d=json.loads(open('gpurun_out/r3u/$name.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms'], d['config'].get('one_sequence_alone_ms'))") $(grep 'tick profile' gpurun_out/r3u/$name.err)"
  [ $rc -ne 0 ] && exit $rc
}
run e3 PCS_TRACK_ENGINES=3
run e2 PCS_TRACK_ENGINES=2
run e1 PCS_TRACK_ENGINES=1
run e4 PCS_TRACK_ENGINES=4
run e3b PCS_TRACK_ENGINES=3
true
