mkdir -p gpurun_out/r4q
for us in 3000 10000 0; do
  for tiles in 0 1; do
    PCS_TRACK_CHUNK_US=$us PCS_ICP_TILES=$tiles python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-stages > gpurun_out/r4q/b12_us${us}_t${tiles}.json 2> gpurun_out/r4q/b12_us${us}_t${tiles}.err
    python - <<PY
import json
d=json.loads(open('gpurun_out/r4q/b12_us${us}_t${tiles}.json').read().strip().splitlines()[-1])
print('us',$us,'tiles',$tiles,'seq/s',d['value'],'ms/step',d['ms_per_step'],'alone ms',d['config']['one_sequence_alone_ms'], flush=True)
PY
  done
done
