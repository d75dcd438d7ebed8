#!/bin/bash
# Several builds of libpcs_hip.so (and, optionally, another source tree with its own bench.py) alternating on ONE box:
#   bash tools/ab_multi.sh ROUNDS [tree:build/r03tree] build/ab/a.so build/ab/b.so ...
# Every run is `bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stages`; the tree's library is put back at the end.
ROUNDS=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
LIB=$ROOT/pcseqlearning_amd/libpcs_hip.so
KEEP=$(mktemp /tmp/ab_keep.XXXXXX.so)
cp $LIB $KEEP
trap 'cp $KEEP $LIB; rm -f $KEEP' EXIT  # (an interrupted run puts the tree's own build back too)
mkdir -p $ROOT/gpurun_out/abm
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    tag=$(basename ${v#tree:} .so)
    if [ "${v#tree:}" != "$v" ]; then
      ( cd $ROOT/${v#tree:} && python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stages ) > $ROOT/gpurun_out/abm/${tag}_$r.json 2> $ROOT/gpurun_out/abm/${tag}_$r.err
    else
      cp $ROOT/$v $LIB
      python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stages > $ROOT/gpurun_out/abm/${tag}_$r.json 2> $ROOT/gpurun_out/abm/${tag}_$r.err
    fi
    python3 - <<PY
import json
try:
    d=json.loads(open("$ROOT/gpurun_out/abm/${tag}_$r.json").read().strip().splitlines()[-1])
    print("$tag round $r:", d["value"], "seq/s", d["step_ms"], "icp us", d["roofline"].get("avg_us_per_launch"), flush=True)
except Exception as e:
    print("$tag round $r: failed", e, flush=True)
PY
  done
done
