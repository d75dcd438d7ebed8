#!/bin/bash
# sequence workers on streams of their own (spread over the four hardware queues, the engines' included) or all on
# the caller's stream (one queue that no engine uses); alternating on one box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/exp4; mkdir -p $O
python3 -c "import bench; bench.load_sequences(list(range(12)), 20, 160000)" >/dev/null 2>&1
for r in 1 2 3; do
  for w in own caller; do
    python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stages --worker-streams $w > $O/${w}_$r.json 2> $O/${w}_$r.err
    python3 - <<PY
import json
try:
    d=json.loads(open("$O/${w}_$r.json").read().strip().splitlines()[-1])
    print("workers $w round $r:", d["value"], "seq/s", d["step_ms"], "icp us", d["roofline"].get("avg_us_per_launch"), flush=True)
except Exception as ex:
    print("workers $w round $r: failed", ex, flush=True)
PY
  done
done
