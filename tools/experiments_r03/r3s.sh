mkdir -p gpurun_out/r3s
run() { # name, args
  name=$1; shift
  PCS_BENCH_TRACE=1 timeout -k 10 240 python bench.py --warmup 2 --no-cpu-baseline --no-stages "$@" > gpurun_out/r3s/$name.json 2> gpurun_out/r3s/$name.err; rc=$?
  echo "$name rc=$rc : $(python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r3s/$name.json').read().strip().splitlines()[-1])
    print('value', d['value'], 'ms', d['ms_per_step'], 'alone', d['config'].get('one_sequence_alone_ms'), 'pair us', d['roofline']['avg_us_per_launch'])
except Exception as e: print('no line', e)
PY
)"
  [ $rc -ne 0 ] && exit $rc
}
cp pcseqlearning_amd/libpcs_hip.so /tmp/cur.so
run cur_b1 --batch 1 --steps 20
run cur_b12 --steps 6
cp pcseqlearning_amd/libpcs_hip_noxcd.so pcseqlearning_amd/libpcs_hip.so
run old_b1 --batch 1 --steps 20
run old_b12 --steps 6
cp /tmp/cur.so pcseqlearning_amd/libpcs_hip.so
run cur_b1_again --batch 1 --steps 20
true
