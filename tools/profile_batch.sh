#!/bin/bash
# Kernel trace of the default bench schedule (a batch of sequences in flight, their tracking units in the
# device's one lock-step engine), summarised on the box (the trace itself is too large to bring back):
#   bash tools/profile_batch.sh <tag> [extra bench args]  ->  gpurun_out/prof/<tag>_batch_kernel_stats.csv,
#   <tag>_batch_trace_summary.txt, <tag>_batch_bench_under_rocprofv3.json
set -o pipefail
# (rocprofv3's tool library dies now and then in the first seconds -- host SIGSEGV, or a malformed packet that aborts
#  a queue and leaves the process waiting: ONE attempt under a short timeout; a failed run is started again by hand)
TAG=${1:-r04}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/prof
mkdir -p $O
cd $ROOT
# the synthetic sequences are generated (child processes) in a plain run first: not under the profiler
python3 -c "import bench; bench.load_sequences(list(range(12)), 20, 160000)" || exit 1
echo "sequences ready" | tee -a $O/batch.progress
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_b
# (bench.py --trace: step lines and Python stacks of a step that takes > 5 s)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o ks -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-stages --trace "$@" > $O/${TAG}_batch_bench_under_rocprofv3.json 2> $O/batch.err || { echo "failed"; tail -5 $O/batch.err; exit 1; }
cp /tmp/prof_b/ks_kernel_stats.csv $O/${TAG}_batch_kernel_stats.csv
# the summary's window = the timed steps of the line just printed (they are the end of the run)
LAST_MS=$(python3 -c "import json,sys; d=json.loads(open('$O/${TAG}_batch_bench_under_rocprofv3.json').read().strip().splitlines()[-1]); print(d['steps'] * d['ms_per_step'])")
python3 $ROOT/tools/trace_summary.py /tmp/prof_b/ks_kernel_trace.csv 45 $LAST_MS | tee $O/${TAG}_batch_trace_summary.txt
