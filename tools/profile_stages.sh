#!/bin/bash
# Kernel stats of ONE sequence per step (bench.py --batch 1) under rocprofv3, only the rows of the pre-tracking stages
# (ground removal, proposal): bash tools/profile_stages.sh <tag>  ->  gpurun_out/prof/<tag>_stage_kernels.txt
set -o pipefail
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/prof
mkdir -p $O
cd $ROOT
python3 -c "import bench; bench.load_sequences([0], 20, 160000)" || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_st
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_st -o ks -- python3 $ROOT/bench.py --batch 1 --steps 3 --warmup 1 --no-cpu-baseline --no-stages > $O/${TAG}_stage_bench.json 2> $O/st.err || { echo "failed"; tail -5 $O/st.err; exit 1; }
cp /tmp/prof_st/ks_kernel_stats.csv $O/${TAG}_sequence_kernel_stats.csv
python3 - <<PY | tee $O/${TAG}_stage_kernels.txt
import csv
rows = list(csv.DictReader(open("/tmp/prof_st/ks_kernel_stats.csv")))
pat = ("knn_", "cc_", "radius_search", "edge_", "irls", "l1_min", "prune", "ransac", "grid_insert", "segment_", "hyp", "pillar",
       "minmax4", "onesweep", "rocprim", "at::native", "plane", "ground", "select", "nn1_")
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"all kernels: {tot / 1e6 / 4:.2f} ms per step (4 steps)")
for r in rows:
    n = r["Name"]
    if any(p in n for p in pat):
        print(f'{n[:90]:90s} calls {int(r["Calls"]):6d}  total {float(r["TotalDurationNs"]) / 1e6:9.3f} ms  avg {float(r["AverageNs"]) / 1e3:9.1f} us')
PY
