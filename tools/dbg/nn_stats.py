import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from pcseqlearning_amd import lib as L
from pcseqlearning_amd.pcdet import models
from tools.run_sequence import build_model
batch = models.load_data_to_gpu(bench.load_sequence(0, 20, 160000))
model = build_model()
lib = L.load()
out = (ctypes.c_ulonglong * 16)()
with torch.no_grad():
    model(batch)
torch.cuda.synchronize()
lib.pcs_dbg_read(out)
f, b = max(out[0], 1), max(out[1], 1)
print("fwd queries %d: confirmed %.3f, no partner %.3f, pushed to coop %.3f" % (out[0], out[2] / f, out[4] / f, out[6] / f))
print("bwd queries %d: confirmed %.3f, no partner %.3f, pushed to coop %.3f" % (out[1], out[3] / b, out[5] / b, out[7] / b))
