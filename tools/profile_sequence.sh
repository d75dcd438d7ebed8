#!/bin/bash
# How the committed profiles of the sequence workload are produced (run on the GPU box from the repo root):
#   bash tools/profile_sequence.sh <tag>      e.g. r03  ->  gpurun_out/prof/<tag>_*
# 1. rocprofv3 --kernel-trace --stats over `bench.py --batch 1` (one sequence per step: its 9 tracking units
#    alone in the device's lock-step engine -- per-kernel numbers without anything beside them);
# 2./3. separate --pmc passes for FETCH_SIZE and WRITE_SIZE (MI355X_MICROARCH.md, HBM section: the two
#    counters do not fit one pass). The batch schedule the bench line is quoted on is profiled by
#    tools/profile_batch.sh.
set -o pipefail
# (rocprofv3's tool library dies now and then in the first seconds: ONE attempt under a timeout; a failed run is
#  started again by hand)

TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/prof
mkdir -p $O
cd $ROOT
python3 -c "import bench; bench.load_sequences([0], 20, 160000)" || exit 1
echo "sequence ready" | tee -a $O/seq.progress
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ks
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks -o ks -- python3 $ROOT/bench.py --batch 1 --steps 3 --warmup 1 --no-cpu-baseline --no-stages > $O/${TAG}_sequence_bench_under_rocprofv3.json 2> $O/ks.err || { echo "kernel-stats pass failed"; tail -5 $O/ks.err; exit 1; }
cp /tmp/prof_ks/ks_kernel_stats.csv $O/${TAG}_sequence_kernel_stats.csv
echo "kernel stats: $(wc -l < $O/${TAG}_sequence_kernel_stats.csv) rows" | tee -a $O/seq.progress
for C in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/prof_$C
  rm -rf $d
  timeout -k 10 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $d -o p -- python3 $ROOT/bench.py --batch 1 --steps 1 --warmup 0 --no-cpu-baseline --no-stages > $O/pmc_$C.out 2> $O/pmc_$C.err || { echo "pmc pass $C failed"; tail -5 $O/pmc_$C.err; exit 1; }
  python3 $ROOT/tools/summarize_pmc.py $d/p_counter_collection.csv > $O/${TAG}_pmc_$C.json
  echo "pmc $C: $(wc -l < $d/p_counter_collection.csv) rows" | tee -a $O/seq.progress
done
