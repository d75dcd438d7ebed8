cd $GRAFT_REPO_ROOT
for S in 0 1; do for G in 3 4 5 6; do
echo "by_span $S groups $G"; PCS_TRACK_GROUP_BY_SPAN=$S PCS_TRACK_GROUPS=$G timeout -k 10 300 python tools/stage_times.py --repeat 3 2>&1 | grep -A3 "run 2" | tail -1
done; done
