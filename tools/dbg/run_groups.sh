O=$GRAFT_REPO_ROOT/gpurun_out/r2g
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_graph_icp.py tests/test_gpu_tracking.py tests/test_gpu_backbone_pipeline.py -q > $O/t.txt 2>&1
echo "tests rc=$?" >> $O/t.txt
tail -3 $O/t.txt
for G in 1 2 3 4 5; do
echo "groups $G"; PCS_TRACK_GROUPS=$G timeout -k 10 300 python tools/stage_times.py --repeat 3 2>&1 | grep -A3 "run 2" | tail -1
done
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o st -- python3 $GRAFT_REPO_ROOT/tools/stage_times.py --repeat 3 > $O/prof_run.txt 2>&1
cp /tmp/prof/st_kernel_stats.csv $O/ 2>/dev/null
