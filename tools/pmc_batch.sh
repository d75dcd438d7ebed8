#!/bin/bash
# HBM-side traffic of the ICP iteration kernels IN THE SCHEDULE THE BENCH LINE IS QUOTED ON (a batch of 12
# sequences in flight, their units in the device's lock-step engines): two rocprofv3 --pmc passes (FETCH_SIZE
# and WRITE_SIZE do not fit one; --pmc with --kernel-trace only, as gpurun demands), summarised per kernel,
# then (fetch + write) / algorithmic of the four iteration kernels per chunk:
#   bash tools/pmc_batch.sh <tag>  ->  gpurun_out/prof/<tag>_batch_pmc_{FETCH,WRITE}_SIZE.json, <tag>_icp_hbm_pmc.json
set -o pipefail
# (rocprofv3's tool library dies now and then in the first seconds -- host SIGSEGV, or a malformed packet that aborts
#  a queue and leaves the process waiting: ONE attempt under a short timeout; a failed run is started again by hand)
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/prof
mkdir -p $O
cd $ROOT
python3 -c "import bench; bench.load_sequences(list(range(12)), 20, 160000)" || exit 1
echo "sequences ready" | tee -a $O/pmc.progress
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/pmcb_$C
  rm -rf $d
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $d -o p -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-stages --trace > $O/${TAG}_batch_pmc_$C.bench.json 2> $O/pmcb_$C.err || { echo "pmc pass $C failed"; tail -5 $O/pmcb_$C.err; exit 1; }
  python3 $ROOT/tools/summarize_pmc.py $d/p_counter_collection.csv > $O/${TAG}_batch_pmc_$C.json
  echo "pmc $C: $(wc -l < $d/p_counter_collection.csv) rows" | tee -a $O/pmc.progress
done
python3 $ROOT/tools/derive_icp_traffic.py $O/${TAG}_batch_pmc_FETCH_SIZE.json $O/${TAG}_batch_pmc_WRITE_SIZE.json $O/${TAG}_batch_pmc_FETCH_SIZE.bench.json > $O/${TAG}_icp_hbm_pmc.json
cat $O/${TAG}_icp_hbm_pmc.json
