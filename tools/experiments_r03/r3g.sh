mkdir -p gpurun_out/r3g
python3 -c "import bench; bench.load_sequences(list(range(24)), 20, 160000)"; echo "sequences ready"
timeout -k 10 300 python tools/host_times.py 45 > gpurun_out/r3g/host_times.txt 2>&1; echo "host_times rc=$?"; head -20 gpurun_out/r3g/host_times.txt
PCS_BENCH_TRACE=1 timeout -k 10 400 python bench.py --steps 100 --warmup 2 --no-cpu-baseline --no-stages > gpurun_out/r3g/soak100.json 2> gpurun_out/r3g/soak100.err; echo "soak100 rc=$?"
grep -E '^step' gpurun_out/r3g/soak100.err | awk '{print $3}' | sort -n | awk '{a[NR]=$1} END{print "steps",NR,"min",a[1],"median",a[int((NR+1)/2)],"p95",a[int(NR*0.95)],"max",a[NR]}'
grep "\[pcs\]" gpurun_out/r3g/soak100.err | head -5
for B in 24; do
  PCS_BENCH_TRACE=1 PCS_TICK_PROFILE=1 timeout -k 10 300 python bench.py --batch $B --steps 5 --warmup 1 --no-cpu-baseline --no-stages > gpurun_out/r3g/b$B.json 2> gpurun_out/r3g/b$B.err; echo "batch $B rc=$?"; grep -E '^step|tick profile' gpurun_out/r3g/b$B.err | tail -7
done
PCS_BENCH_TRACE=1 PCS_TICK_PROFILE=1 timeout -k 10 300 python bench.py --batch 24 --inflight 16 --steps 5 --warmup 1 --no-cpu-baseline --no-stages > gpurun_out/r3g/b24i16.json 2> gpurun_out/r3g/b24i16.err; echo "batch 24 inflight 16 rc=$?"; grep -E '^step|tick profile' gpurun_out/r3g/b24i16.err | tail -7
timeout -k 10 600 python -m pytest tests/test_gpu_full_size.py -x -q > gpurun_out/r3g/full.log 2>&1; echo "full-size tests rc=$?"; tail -5 gpurun_out/r3g/full.log
