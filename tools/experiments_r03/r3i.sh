mkdir -p gpurun_out/r3i
timeout -k 10 300 python -m pytest tests/test_gpu_tracking.py tests/test_gpu_graph_icp.py -x -q -m gpu > gpurun_out/r3i/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3i/tests.log
python3 -c "import bench; bench.load_sequences(list(range(12)), 20, 160000)"; echo "sequences ready"
run() { name=$1; shift; env "$@" PCS_BENCH_TRACE=1 PCS_TICK_PROFILE=1 timeout -k 10 200 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-stages $EXTRA > gpurun_out/r3i/$name.json 2> gpurun_out/r3i/$name.err; echo "$name rc=$? : $(grep -E '^step' gpurun_out/r3i/$name.err | awk '{print $3}' | sort -n | awk '{a[NR]=$1} END{print "steps",NR,"min",a[1],"median",a[int((NR+1)/2)],"max",a[NR]}') $(grep 'tick profile' gpurun_out/r3i/$name.err | tail -1)"; }
EXTRA="" run staged PCS_DUMMY=1
EXTRA="" run direct PCS_ICP_SOLVE_STAGED=0
EXTRA="" run staged2 PCS_DUMMY=1
EXTRA="" run direct2 PCS_ICP_SOLVE_STAGED=0
EXTRA="" run e4 PCS_TRACK_ENGINES=4
EXTRA="--batch 18" run b18 PCS_DUMMY=1
