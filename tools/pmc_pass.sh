#!/bin/bash
# One rocprofv3 --pmc pass over a one-sequence bench step, summarised per kernel:
#   bash tools/pmc_pass.sh <tag> <counter> [<counter> ...]   ->  gpurun_out/prof/<tag>.json
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
d=/tmp/pmc_$TAG
rm -rf $d
timeout -k 10 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $d -o p -- python3 $ROOT/bench.py --batch 1 --steps 1 --warmup 0 --no-cpu-baseline --no-stages > $O/$TAG.out 2> $O/$TAG.err || { echo "pmc pass $TAG failed"; tail -5 $O/$TAG.err; exit 1; }
python3 $ROOT/tools/summarize_pmc.py $d/p_counter_collection.csv > $O/$TAG.json
echo "pmc $TAG: $(wc -l < $d/p_counter_collection.csv) rows"
