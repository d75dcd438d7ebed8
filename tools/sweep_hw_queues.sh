#!/bin/bash
# GPU_MAX_HW_QUEUES (HIP's hardware queues per process and priority; default 4: three carry the tracker's engines) swept
# on ONE box, alternating:   bash tools/sweep_hw_queues.sh "4 6 8" [rounds] [more bench args]
VALUES=$1; ROUNDS=${2:-2}; shift 2 2>/dev/null
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/hwq
mkdir -p $O
cd $ROOT
for r in $(seq 1 $ROUNDS); do
  for v in $VALUES; do
    GPU_MAX_HW_QUEUES=$v timeout -k 10 150 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stages "$@" > $O/${v}_$r.json 2> $O/${v}_$r.err || { echo "GPU_MAX_HW_QUEUES $v round $r: failed or timed out"; tail -3 $O/${v}_$r.err; exit 1; }
    python3 - <<PY
import json
d = json.loads(open("$O/${v}_$r.json").read().strip().splitlines()[-1])
print("GPU_MAX_HW_QUEUES $v round $r:", d["value"], "seq/s", d["ms_per_step"], d.get("step_ms"), "alone", d["config"].get("one_sequence_alone_ms"), flush=True)
PY
  done
done
