import sys, os
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch, bench
from pcseqlearning_amd import ops, lib as L
from pcseqlearning_amd.pcdet import models
from tools.run_sequence import build_model
orig = L.workspace
sizes = []
def ws(nbytes, dev):
    sizes.append(int(nbytes)); return orig(nbytes, dev)
L.workspace = ws
ops.L.workspace = ws
batch = models.load_data_to_gpu(bench.load_sequence(0, 20, 160000))
model = build_model(save=False)
with torch.no_grad():
    model(batch); torch.cuda.synchronize()
    sizes.clear(); torch.cuda.reset_peak_memory_stats()
    model(batch); torch.cuda.synchronize()
big = sorted(sizes, reverse=True)[:8]
print('workspaces (MiB), largest:', [round(b / 2**20, 1) for b in big], 'count', len(sizes))
print('peak allocated GiB', round(torch.cuda.max_memory_allocated() / 2**30, 2), 'reserved GiB', round(torch.cuda.memory_reserved() / 2**30, 2))
print('inputs GiB', round(sum(v.numel() * v.element_size() for v in batch.values() if torch.is_tensor(v)) / 2**30, 3))
