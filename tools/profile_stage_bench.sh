#!/bin/bash
# rocprofv3 kernel stats of tools/stage_bench.py (ground removal and proposal of one sequence, alone on the device):
#   bash tools/profile_stage_bench.sh <tag>  ->  gpurun_out/prof/<tag>_stagebench_kernels.txt
set -o pipefail
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/prof
mkdir -p $O
cd $ROOT
python3 -c "import bench; bench.load_sequences([0], 20, 160000)" || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_sb
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sb -o ks -- python3 $ROOT/tools/stage_bench.py --reps 4 > $O/${TAG}_stagebench.json 2> $O/sb.err || { echo "failed"; tail -5 $O/sb.err; exit 1; }
python3 - <<PY | tee $O/${TAG}_stagebench_kernels.txt
import csv
rows = list(csv.DictReader(open("/tmp/prof_sb/ks_kernel_stats.csv")))
for r in rows[:45]:
    print(f'{r["Name"][:100]:100s} calls {int(r["Calls"]):6d}  total {float(r["TotalDurationNs"]) / 1e6:9.3f} ms  avg {float(r["AverageNs"]) / 1e3:9.1f} us')
PY
