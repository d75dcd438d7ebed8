mkdir -p gpurun_out/r3ai
timeout -k 10 700 python -m pytest tests/test_gpu_ground.py tests/test_gpu_backbone_pipeline.py -x -q -m gpu > gpurun_out/r3ai/tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -3 gpurun_out/r3ai/tests.log
[ $rc -ne 0 ] && exit $rc
for n in a b; do
PCS_BENCH_TRACE=1 timeout -k 10 300 python bench.py --warmup 2 --no-cpu-baseline --steps 8 > gpurun_out/r3ai/b12$n.json 2> gpurun_out/r3ai/b12$n.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('gpurun_out/r3ai/b12$n.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms'], d['config'].get('one_sequence_alone_ms'), d['roofline']['avg_us_per_launch'], d['roofline']['frac']); print([ (s['stage'], s['ms']) for s in d['roofline']['stages'][:4]])"
done
