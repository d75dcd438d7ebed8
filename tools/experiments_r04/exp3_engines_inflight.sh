#!/bin/bash
# engines / sequences in flight after the in-place moving grid (alternating on one box)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/exp3; mkdir -p $O
python3 -c "import bench; bench.load_sequences(list(range(16)), 20, 160000)" >/dev/null 2>&1
run() {  # tag, args
  tag=$1; shift
  python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stages "$@" > $O/$tag.json 2> $O/$tag.err
  python3 - <<PY
import json
try:
    d=json.loads(open("$O/$tag.json").read().strip().splitlines()[-1])
    print("$tag:", d["value"], "seq/s", d["ms_per_step"], "icp us", d["roofline"].get("avg_us_per_launch"), flush=True)
except Exception as ex:
    print("$tag: failed", ex, flush=True)
PY
}
for r in 1 2; do
  run e3_b12_$r --engines 3
  run e2_b12_$r --engines 2
  run e4_b12_$r --engines 4
  run e3_b16_$r --engines 3 --batch 16 --inflight 16
  run e3_b9_$r --engines 3 --batch 9 --inflight 9
done
