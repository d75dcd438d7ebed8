mkdir -p gpurun_out/r3t
lscpu | grep -i "model name\|mhz\|^cpu(s)" > gpurun_out/r3t/cpu.txt; nproc >> gpurun_out/r3t/cpu.txt; cat /proc/loadavg >> gpurun_out/r3t/cpu.txt
rocm-smi --showclocks --showpower --showperflevel > gpurun_out/r3t/smi_before.txt 2>&1
PCS_TICK_PROFILE=1 PCS_BENCH_TRACE=1 timeout -k 10 240 python bench.py --warmup 2 --no-cpu-baseline --no-stages --steps 6 > gpurun_out/r3t/b12.json 2> gpurun_out/r3t/b12.err; echo "b12 rc=$?"; grep "tick profile" gpurun_out/r3t/b12.err
rocm-smi --showclocks --showpower --showperflevel > gpurun_out/r3t/smi_after.txt 2>&1
PCS_TICK_PROFILE=1 timeout -k 10 240 python bench.py --warmup 2 --no-cpu-baseline --no-stages --batch 1 --steps 20 > gpurun_out/r3t/b1.json 2> gpurun_out/r3t/b1.err; echo "b1 rc=$?"; grep "tick profile" gpurun_out/r3t/b1.err
python -c "
import json
for n in ('b12','b1'):
    d=json.loads(open('gpurun_out/r3t/%s.json'%n).read().strip().splitlines()[-1]); print(n, d['ms_per_step'], d['config'].get('one_sequence_alone_ms'))"
cat gpurun_out/r3t/cpu.txt; cat /proc/loadavg
grep -i "sclk\|mclk\|fclk\|power\|perf" gpurun_out/r3t/smi_after.txt | head -12
