#!/bin/bash
# A/B of two builds of libpcs_hip.so on ONE box (run-to-run and box-to-box spread is a few per cent: only
# alternating runs on the same box tell builds apart):  bash tools/ab_bench.sh A.so B.so [rounds] [bench args...]
# The library in the tree is replaced in turn (and put back at the end); every run is `bench.py --steps 3`.
A=$1; B=$2; ROUNDS=${3:-2}; shift 3 2>/dev/null
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
LIB=$ROOT/pcseqlearning_amd/libpcs_hip.so
KEEP=$(mktemp /tmp/ab_keep.XXXXXX.so)
cp $LIB $KEEP
trap 'cp $KEEP $LIB; rm -f $KEEP' EXIT  # (an interrupted run puts the tree's own build back too)
mkdir -p $ROOT/gpurun_out/ab
python3 -c "import bench; bench.load_sequences(list(range(12)), 20, 160000)" >/dev/null 2>&1
for r in $(seq 1 $ROUNDS); do
  for v in A B; do
    src=$A; [ $v = B ] && src=$B
    cp $src $LIB
    python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stages "$@" > $ROOT/gpurun_out/ab/${v}_$r.json 2> $ROOT/gpurun_out/ab/${v}_$r.err
    python3 - <<PY
import json
try:
    d=json.loads(open("$ROOT/gpurun_out/ab/${v}_$r.json").read().strip().splitlines()[-1])
    print("$v round $r:", d["value"], "seq/s", d["step_ms"], "icp chunk us", d["roofline"]["avg_us_per_launch"], flush=True)
except Exception as e:
    print("$v round $r: failed", e, flush=True)
PY
  done
done
