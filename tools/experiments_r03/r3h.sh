mkdir -p gpurun_out/r3h gpurun_out/prof
timeout -k 10 400 python -m pytest tests/test_gpu_tracking.py tests/test_gpu_graph_icp.py -x -q -m gpu > gpurun_out/r3h/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3h/tests.log
python3 -c "import bench; bench.load_sequences(list(range(12)), 20, 160000)"; echo "sequences ready"
run() { name=$1; shift; env "$@" PCS_BENCH_TRACE=1 PCS_TICK_PROFILE=1 timeout -k 10 200 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-stages $EXTRA > gpurun_out/r3h/$name.json 2> gpurun_out/r3h/$name.err; echo "$name rc=$? : $(grep -E '^step' gpurun_out/r3h/$name.err | awk '{print $3}' | sort -n | awk '{a[NR]=$1} END{print "steps",NR,"min",a[1],"median",a[int((NR+1)/2)],"max",a[NR]}') $(grep 'tick profile' gpurun_out/r3h/$name.err | tail -1)"; }
EXTRA="" run base PCS_DUMMY=1
EXTRA="--stagger-ms 0" run stagger0 PCS_DUMMY=1
EXTRA="--stagger-ms 5" run stagger5 PCS_DUMMY=1
EXTRA="--stagger-ms 5" run prio PCS_TRACK_PRIORITY=1
EXTRA="--stagger-ms 5" run e1 PCS_TRACK_ENGINES=1
EXTRA="--stagger-ms 5" run e3 PCS_TRACK_ENGINES=3
EXTRA="--stagger-ms 5" run chunk6 PCS_TRACK_CHUNK=6
EXTRA="--stagger-ms 5" run chunk12 PCS_TRACK_CHUNK=12
