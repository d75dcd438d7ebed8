mkdir -p gpurun_out/r3e
python3 -c "import bench; bench.load_sequences(list(range(12)), 20, 160000)"; echo "sequences ready"
PCS_BENCH_TRACE=1 timeout -k 10 400 python bench.py --steps 40 --warmup 2 --no-cpu-baseline --no-stages > gpurun_out/r3e/soak_a.json 2> gpurun_out/r3e/soak_a.err; echo "soak A rc=$?"
grep -E "^step" gpurun_out/r3e/soak_a.err | awk '{print $3}' | sort -n | awk '{a[NR]=$1} END{print "A steps",NR,"min",a[1],"median",a[int((NR+1)/2)],"max",a[NR]}'
grep "\[pcs\]" gpurun_out/r3e/soak_a.err | head -5
grep -E "^step (0|1|2|10|20|39):" gpurun_out/r3e/soak_a.err
HSA_ENABLE_INTERRUPT=0 PCS_BENCH_TRACE=1 timeout -k 10 400 python bench.py --steps 40 --warmup 2 --no-cpu-baseline --no-stages > gpurun_out/r3e/soak_b.json 2> gpurun_out/r3e/soak_b.err; echo "soak B (HSA_ENABLE_INTERRUPT=0) rc=$?"
grep -E "^step" gpurun_out/r3e/soak_b.err | awk '{print $3}' | sort -n | awk '{a[NR]=$1} END{print "B steps",NR,"min",a[1],"median",a[int((NR+1)/2)],"max",a[NR]}'
grep "\[pcs\]" gpurun_out/r3e/soak_b.err | head -5
