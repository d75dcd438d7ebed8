mkdir -p gpurun_out/r3f
python3 -c "import bench; bench.load_sequences(list(range(12)), 20, 160000)"; echo "sequences ready"
run() {  # name, env...
  name=$1; shift
  env "$@" PCS_BENCH_TRACE=1 timeout -k 10 170 python bench.py --steps 30 --warmup 2 --no-cpu-baseline --no-stages > gpurun_out/r3f/$name.json 2> gpurun_out/r3f/$name.err
  echo "$name rc=$? : $(grep -E '^step' gpurun_out/r3f/$name.err | awk '{print $3}' | sort -n | awk '{a[NR]=$1} END{print "steps",NR,"min",a[1],"median",a[int((NR+1)/2)],"max",a[NR]}') stalls>3s: $(grep -E '^step' gpurun_out/r3f/$name.err | awk '$3>3000' | wc -l)"
}
run noevents PCS_BENCH_NO_ICP_EVENTS=1
run nodefer PCS_BENCH_NO_DEFER=1
run hwq4 GPU_MAX_HW_QUEUES=4
run hwq24 GPU_MAX_HW_QUEUES=24
run base PCS_DUMMY=1
