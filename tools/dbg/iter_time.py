"""ICP loop kernel time per iteration over one synthetic sequence (HIP events on the launches)."""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from pcseqlearning_amd import lib as L
from pcseqlearning_amd.pcdet import models
from tools.run_sequence import build_model
batch = models.load_data_to_gpu(bench.load_sequence(0, 20, 160000))
model = build_model()
lib = L.load()
for r in range(3):
    lib.pcs_profile_icp(ctypes.c_int(1))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        model(batch)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    prof = (ctypes.c_double * 9)()
    lib.pcs_profile_icp_read(prof)
    lib.pcs_profile_icp(ctypes.c_int(0))
    ms, launches, nbytes, _, iters = prof[0], prof[1], prof[2], prof[3], prof[4]
    print(f"run {r}: total {dt*1e3:.1f} ms | loop launches {launches:.0f}, kernel ms {ms:.1f}, "
          f"iterations(max per launch summed) {iters:.0f}, us/iteration {ms*1e3/max(iters,1):.1f}, "
          f"GB/s {nbytes/max(ms,1e-9)/1e6:.1f}, MB/launch {nbytes/max(launches,1)/1e6:.2f}")
