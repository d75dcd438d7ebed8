import os, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from pcseqlearning_amd.pcdet import models
from tools.run_sequence import build_model
N = int(sys.argv[1]); K = int(sys.argv[2])
batches = [models.load_data_to_gpu(bench.load_sequence(i, 20, 160000)) for i in range(N)]
mods = [build_model() for _ in range(N)]
streams = [torch.cuda.Stream() for _ in range(N)]
def work(i, k):
    with torch.cuda.stream(streams[i]), torch.no_grad():
        for _ in range(k):
            mods[i](batches[i])
        streams[i].synchronize()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i, K)) for i in range(N)]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"inflight {N} rep {rep}: {N*K} sequences in {dt*1e3:.0f} ms -> {N*K/dt:.2f} seq/s", flush=True)
