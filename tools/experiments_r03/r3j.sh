mkdir -p gpurun_out/r3j
python3 -c "import bench; bench.load_sequences(list(range(12)), 20, 160000)"; echo "sequences ready"
PCS_BENCH_TRACE=1 PCS_TICK_PROFILE=1 timeout -k 10 500 python bench.py --steps 120 --warmup 2 --no-cpu-baseline --no-stages > gpurun_out/r3j/soak.json 2> gpurun_out/r3j/soak.err; echo "soak rc=$?"
grep -E '^step' gpurun_out/r3j/soak.err | awk '{print $3}' | sort -n | awk '{a[NR]=$1} END{print "steps",NR,"min",a[1],"median",a[int((NR+1)/2)],"p95",a[int(NR*0.95)],"max",a[NR]}'
grep "\[pcs\]\|tick profile" gpurun_out/r3j/soak.err | head -5
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3j/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -4 gpurun_out/r3j/gpu_tests.log
