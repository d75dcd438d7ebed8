#!/bin/bash
# start stagger of the sequence workers (alternating on one box)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/exp5; mkdir -p $O
python3 -c "import bench; bench.load_sequences(list(range(12)), 20, 160000)" >/dev/null 2>&1
for r in 1 2; do
  for w in 5 0 15 30 60; do
    python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stages --stagger-ms $w > $O/s${w}_$r.json 2> $O/s${w}_$r.err
    python3 - <<PY
import json
try:
    d=json.loads(open("$O/s${w}_$r.json").read().strip().splitlines()[-1])
    print("stagger $w round $r:", d["value"], "seq/s", d["step_ms"], flush=True)
except Exception as ex:
    print("stagger $w round $r: failed", ex, flush=True)
PY
  done
done
