O=$GRAFT_REPO_ROOT/gpurun_out/r2l
mkdir -p $O
cd $GRAFT_REPO_ROOT
for Q in 8 16; do
for G in 4 6 9; do
echo "legacy queues $Q groups $G"; GPU_MAX_HW_QUEUES=$Q PCS_ICP_ONE_LAUNCH=0 PCS_TRACK_GROUPS=$G timeout -k 10 200 python tools/stage_times.py --repeat 3 2>&1 | grep -A3 "run 2" | tail -1
done
done
for G in 4 9; do
echo "one-launch groups $G"; GPU_MAX_HW_QUEUES=16 PCS_TRACK_GROUPS=$G timeout -k 10 200 python tools/stage_times.py --repeat 3 2>&1 | grep -A3 "run 2" | tail -1
done
