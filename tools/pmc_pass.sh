#!/bin/bash
# One rocprofv3 --pmc pass over a one-sequence bench step, summarised per kernel:
#   bash tools/pmc_pass.sh <tag> "<counter> [<counter> ...]" [bench.py arguments ...]   ->  gpurun_out/prof/<tag>.json
# e.g.  bash tools/pmc_pass.sh r04_pmc_tcp_tiles "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum" --engines 1 --schedule resident_tiles
set -o pipefail
# (rocprofv3's tool library dies now and then in the first seconds: ONE attempt under a timeout; a failed run is
#  started again by hand)

TAG=$1; COUNTERS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
d=/tmp/pmc_$TAG
rm -rf $d
timeout -k 10 600 rocprofv3 --pmc $COUNTERS --kernel-trace --output-format csv -d $d -o p -- python3 $ROOT/bench.py --batch 1 --steps 1 --warmup 0 --no-cpu-baseline --no-stages "$@" > $O/$TAG.out 2> $O/$TAG.err || { echo "pmc pass $TAG failed"; tail -5 $O/$TAG.err; exit 1; }
python3 $ROOT/tools/summarize_pmc.py $d/p_counter_collection.csv > $O/$TAG.json
echo "pmc $TAG: $(wc -l < $d/p_counter_collection.csv) rows"
