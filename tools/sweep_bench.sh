#!/bin/bash
# One bench.py argument swept on ONE box, alternating (box-to-box spread is larger than most effects):
#   bash tools/sweep_bench.sh <flag> "<v1> <v2> ..." [rounds] [more bench args]   e.g.  --icp-widening "16 8 4" 2
FLAG=$1; VALUES=$2; ROUNDS=${3:-2}; shift 3 2>/dev/null
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/sweep
mkdir -p $O
cd $ROOT
for r in $(seq 1 $ROUNDS); do
  for v in $VALUES; do
    python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stages $FLAG $v "$@" > $O/${v}_$r.json 2> $O/${v}_$r.err
    python3 - <<PY
import json
try:
    d = json.loads(open("$O/${v}_$r.json").read().strip().splitlines()[-1])
    print("$FLAG $v round $r:", d["value"], "seq/s", d["ms_per_step"], d.get("step_ms"), "alone", d["config"].get("one_sequence_alone_ms"), flush=True)
except Exception as e:
    print("$FLAG $v round $r: failed", e, flush=True)
PY
  done
done
