O=$GRAFT_REPO_ROOT/gpurun_out/r2i
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"
tail -c 3000 $O/bench.json
tail -5 $O/bench.err
