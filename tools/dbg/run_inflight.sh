cd $GRAFT_REPO_ROOT
for M in 0 1; do
for N in 1 2 3; do
echo "one_launch=$M"; PCS_ICP_ONE_LAUNCH=$M timeout -k 10 200 python tools/dbg/inflight.py $N 4 2>&1 | tail -1
done
done
