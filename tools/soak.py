"""Memory soak: three sequences in turn through one model; allocator figures per step."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pcseqlearning_amd.pcdet import models
from tools.run_sequence import build_model
model = build_model(save=False)
seqs = [models.load_data_to_gpu(bench.load_sequence(s, 20, 160000)) for s in (0, 1, 2)]
torch.cuda.synchronize()
for it in range(12):
    b = seqs[it % 3]
    t0 = time.perf_counter()
    with torch.no_grad():
        model(b)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = model.last_outputs[0]
    n_ex = sum(v["fxyz"].shape[0] for v in out["cluster_tracking_results"].values())
    print(f"it {it} seq {it%3}: {dt*1e3:7.1f} ms  alloc {torch.cuda.memory_allocated()/2**20:8.1f} MiB  reserved {torch.cuda.memory_reserved()/2**20:8.1f} MiB  extracted {n_ex}  iou {float(out['gt_box_best_iou_tracking'].mean()):.4f}", flush=True)
