#!/bin/bash
# A/B of two source TREES on ONE box (e.g. last round's final commit checked out and built under build/<name>
# against the working tree):  bash tools/ab_trees.sh build/r03tree . [rounds] [bench args...]
# Each run is that tree's own `bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stages`.
A=$1; B=$2; ROUNDS=${3:-2}; shift 3 2>/dev/null
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/abt
for r in $(seq 1 $ROUNDS); do
  for v in A B; do
    tree=$A; [ $v = B ] && tree=$B
    ( cd $ROOT/$tree && python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stages "$@" ) \
        > $ROOT/gpurun_out/abt/${v}_$r.json 2> $ROOT/gpurun_out/abt/${v}_$r.err
    python3 - <<PY
import json
try:
    d=json.loads(open("$ROOT/gpurun_out/abt/${v}_$r.json").read().strip().splitlines()[-1])
    print("$v ($tree) round $r:", d["value"], "seq/s", d["ms_per_step"], d.get("step_ms"), flush=True)
except Exception as e:
    print("$v round $r: failed", e, flush=True)
PY
  done
done
