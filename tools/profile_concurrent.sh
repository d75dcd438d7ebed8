#!/bin/bash
# Kernel trace of the batch mode (four sequences in flight) and how much the kernels overlap:
#   bash tools/profile_concurrent.sh <tag>   ->  gpurun_out/prof/<tag>_concurrency.txt, <tag>_concurrent_kernel_stats.csv
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c
export PCS_PROFILE_CONCURRENT=1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c -o ks -- python3 $ROOT/bench.py --batch 12 --steps 2 --warmup 1 --no-cpu-baseline --no-stages > $O/${TAG}_concurrent_bench.json 2> $O/conc.err || { echo "failed"; tail -5 $O/conc.err; exit 1; }
cp /tmp/prof_c/ks_kernel_stats.csv $O/${TAG}_concurrent_kernel_stats.csv
python3 $ROOT/tools/trace_concurrency.py /tmp/prof_c/ks_kernel_trace.csv | tee $O/${TAG}_concurrency.txt
