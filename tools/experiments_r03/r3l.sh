mkdir -p gpurun_out/r3l
python3 -c "import bench; bench.load_sequences(list(range(12)), 20, 160000)"; echo "sequences ready"
run() { name=$1; shift; env "$@" PCS_BENCH_TRACE=1 PCS_TICK_PROFILE=1 timeout -k 10 200 python bench.py --steps 10 --warmup 1 --no-cpu-baseline --no-stages $EXTRA > gpurun_out/r3l/$name.json 2> gpurun_out/r3l/$name.err; echo "$name rc=$? : $(grep -E '^step' gpurun_out/r3l/$name.err | awk '{print $3}' | sort -n | awk '{a[NR]=$1} END{print "steps",NR,"min",a[1],"median",a[int((NR+1)/2)],"max",a[NR]}') $(grep 'tick profile' gpurun_out/r3l/$name.err | tail -1)"; }
EXTRA="" run q4e3 PCS_DUMMY=1
EXTRA="" run q24e3 GPU_MAX_HW_QUEUES=24
EXTRA="" run q24e6 GPU_MAX_HW_QUEUES=24 PCS_TRACK_ENGINES=6
EXTRA="" run q24e6s GPU_MAX_HW_QUEUES=24 PCS_TRACK_ENGINES=6 PCS_ICP_MOM_STREAM=1
EXTRA="" run q32e8 GPU_MAX_HW_QUEUES=32 PCS_TRACK_ENGINES=8
EXTRA="" run q4e6 PCS_TRACK_ENGINES=6
