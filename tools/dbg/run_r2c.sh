set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r2f
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_graph_icp.py tests/test_gpu_tracking.py tests/test_gpu_backbone_pipeline.py -q > $O/t.txt 2>&1
echo "tests rc=$?" >> $O/t.txt
timeout -k 10 300 python tools/stage_times.py --repeat 3 > $O/stage.txt 2>&1
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o st -- python3 $GRAFT_REPO_ROOT/tools/stage_times.py --repeat 3 > $O/prof_run.txt 2>&1
cp /tmp/prof/st_kernel_stats.csv $O/ 2>/dev/null
tail -5 $O/t.txt; tail -8 $O/stage.txt
