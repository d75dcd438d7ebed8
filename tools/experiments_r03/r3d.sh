mkdir -p gpurun_out/r3d
timeout -k 10 600 python -m pytest tests/test_gpu_tracking.py tests/test_gpu_graph_icp.py tests/test_gpu_ground.py -x -q -m gpu > gpurun_out/r3d/tests.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/r3d/tests.log
python3 -c "import bench; bench.load_sequences(list(range(12)), 20, 160000)"; echo "sequences ready"
for E in 1 2 3; do
  PCS_TRACK_ENGINES=$E PCS_TICK_PROFILE=1 PCS_BENCH_TRACE=1 timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-stages > gpurun_out/r3d/bench_e$E.json 2> gpurun_out/r3d/bench_e$E.err; echo "engines $E rc=$?"; grep -E "^step|tick profile" gpurun_out/r3d/bench_e$E.err | tail -5
done
PCS_TRACK_ENGINES=2 PCS_TRACK_CHUNK=4 PCS_TICK_PROFILE=1 PCS_BENCH_TRACE=1 timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-stages > gpurun_out/r3d/bench_e2c4.json 2> gpurun_out/r3d/bench_e2c4.err; echo "e2 chunk4 rc=$?"; grep -E "^step|tick profile" gpurun_out/r3d/bench_e2c4.err | tail -5
