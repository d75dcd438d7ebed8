mkdir -p gpurun_out/r3v
(hostname; uname -r; cat /sys/module/amdgpu/version 2>/dev/null; rocm-smi --showfwinfo 2>/dev/null | grep -i "MEC\|SMC\|SDMA\|RLC " | head -6; rocm-smi --showserial --showuniqueid 2>/dev/null | grep -i "unique\|serial" | head -3) > gpurun_out/r3v/box.txt 2>&1
cat gpurun_out/r3v/box.txt
run() { name=$1; shift
  env "$@" timeout -k 10 240 python bench.py --warmup 2 --no-cpu-baseline --no-stages --steps 6 > gpurun_out/r3v/$name.json 2> gpurun_out/r3v/$name.err; rc=$?
  echo "$name rc=$rc $(python -c "
import json
d=json.loads(open('gpurun_out/r3v/$name.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms']['median'], d['config'].get('one_sequence_alone_ms'))")"
  [ $rc -ne 0 ] && exit $rc
}
run e2 PCS_TRACK_ENGINES=2
run e3 PCS_TRACK_ENGINES=3
run e2b PCS_TRACK_ENGINES=2
run e3b PCS_TRACK_ENGINES=3
true
