cd $GRAFT_REPO_ROOT
for Q in 4 8; do for G in 2 3 4; do
echo "queues $Q groups $G"; GPU_MAX_HW_QUEUES=$Q PCS_TRACK_GROUPS=$G timeout -k 10 300 python tools/stage_times.py --repeat 3 2>&1 | grep -A3 "run 2" | tail -1
done; done
