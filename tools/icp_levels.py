"""Per-level cost of the fused ICP on one real frame pair of the synthetic sequence: sizes,
iterations, mean icp_nn_kernel time (HIP events) and wall time per iteration."""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pcseqlearning_amd import lib as L, ops  # noqa: E402
from pcseqlearning_amd.pcdet import models  # noqa: E402
from pcseqlearning_amd.pcdet.models.model_utils.grid_sampling import GridSampling3D  # noqa: E402
from pcseqlearning_amd.pcdet.models.registration.preprocessors import cluster_tracking as CT  # noqa: E402
from pcseqlearning_amd.pcdet.utils.common_utils import EasyDict  # noqa: E402
from tools.run_sequence import build_model  # noqa: E402

batch = models.load_data_to_gpu(bench.load_sequence(0, 20, 160000))
model = build_model(streams=1)
with torch.no_grad():
    model(batch)
out = model.last_outputs[0]
lib = L.load()
fxyz = out["point_fxyz"]
for key in ("component_rad1x25", "component_rad0x25"):
    comp_all = out["point_" + key]
    diam = CT.component_diameter(EasyDict(fxyz=fxyz, component=comp_all))
    stat_all = diam[comp_all] > 12.5
    for (fa, fb) in ((8, 9), (8, 12)):
        ma, mb = fxyz[:, 0] == fa, fxyz[:, 0] == fb
        A = EasyDict(fxyz=fxyz[ma].contiguous(), component=comp_all[ma] - comp_all[ma].min(),
                     stationary=stat_all[ma], frame=fxyz[ma][:, 0].long())
        B = EasyDict(fxyz=fxyz[mb].contiguous(), component=comp_all[mb], stationary=stat_all[mb],
                     frame=fxyz[mb][:, 0].long())
        C = int(A.component.max()) + 1
        for radius, vs in ((2.5, [0.4, 0.4, 0.6]), (1.25, [0.2, 0.2, 0.3]), (1.0, [0.1, 0.1, 0.15])):
            sampler = GridSampling3D(vs)
            sa, sb = CT.sample_frame(sampler, A), CT.sample_frame(sampler, B)
            mov = sa.fxyz[~sa.stationary].contiguous()
            mcomp = sa.component[~sa.stationary]
            ref = sb.fxyz[~sb.stationary].contiguous()
            deg = torch.bincount(sa.component.long(), minlength=C)
            best = None
            for rep in range(3):
                m = mov.clone()
                lib.pcs_profile_icp_nn(ctypes.c_int(1))
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                T, l1, ratio, iters = ops.register_to_next_frame(m, mcomp, ref, sb.fxyz, deg, C, radius, fb - fa,
                                                                 10.0, 80, 0.05)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                prof = (ctypes.c_double * 3)()
                lib.pcs_profile_icp_nn_read(prof)
                lib.pcs_profile_icp_nn(ctypes.c_int(0))
                cur = (dt, prof[0], prof[1])
                best = cur if best is None or cur[0] < best[0] else best
            dt, ms, n = best
            print(f"{key} {fa}->{fb} r={radius}: C={C} Mn={mov.shape[0]} Rn={ref.shape[0]} Ra={sb.fxyz.shape[0]} "
                  f"iters={iters}  nn {ms * 1e3 / max(n, 1):6.1f} us/launch  call {dt * 1e3:7.2f} ms "
                  f"({dt * 1e6 / max(iters, 1):6.1f} us/iter)", flush=True)
