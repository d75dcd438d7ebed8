mkdir -p gpurun_out/r3y
run() { name=$1; shift
  env "$@" timeout -k 10 240 python bench.py --warmup 2 --no-cpu-baseline --no-stages --steps 6 > gpurun_out/r3y/$name.json 2> gpurun_out/r3y/$name.err; rc=$?
  echo "$name rc=$rc $(python -c "
import json
d=json.loads(open('gpurun_out/r3y/$name.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms']['median'], d['config'].get('one_sequence_alone_ms'))")"
  [ $rc -ne 0 ] && exit $rc
}
run e3_early PCS_TRACK_ENGINES=3
run e3_late PCS_TRACK_ENGINES=3 PCS_NO_EARLY_ENGINES=1
run e2_early PCS_TRACK_ENGINES=2
run e2_late PCS_TRACK_ENGINES=2 PCS_NO_EARLY_ENGINES=1
run e3_early2 PCS_TRACK_ENGINES=3
true
