"""How the fused ICP scales with concurrent identical problems: K host threads each register the
same real frame pair (own clone, own stream), (a) as independent launches, (b) with the shared
launcher (pcs_icp_share_launches) stacking the K problems into one launch per kernel. Wall time
per call tells whether the chip has room for more problems per launch."""
import ctypes
import os
import sys
import threading
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

MASK_MODE = os.environ.get("PCS_CU_MASK", "")   # "", "block" (contiguous 256/K bits) or "xcd" (bit i -> XCD i % 8)


def make_streams(K):
    if not MASK_MODE or K == 1:
        return [torch.cuda.Stream() for _ in range(K)]
    lib_ = L.load()
    out = []
    per = 256 // K
    for w in range(K):
        bits = [0] * 8
        for i in range(256):
            if MASK_MODE == "block":
                on = w * per <= i < (w + 1) * per
            else:
                on = (i % 8) * K // 8 == w
            if on:
                bits[i // 32] |= 1 << (i % 32)
        arr = (ctypes.c_uint32 * 8)(*bits)
        h = ctypes.c_void_p()
        rc = lib_.pcs_stream_create_cu_mask(arr, ctypes.c_int(8), ctypes.byref(h))
        assert rc == 0, rc
        out.append(torch.cuda.ExternalStream(h.value))
    return out


batch = models.load_data_to_gpu(bench.load_sequence(0, 20, 160000))
model = build_model(streams=1)
with torch.no_grad():
    model(batch)
out = model.last_outputs[0]
lib = L.load()
fxyz = out["point_fxyz"]
key = "component_rad1x25"
comp_all = out["point_" + key]
diam = CT.component_diameter(EasyDict(fxyz=fxyz, component=comp_all))
stat_all = diam[comp_all] > 12.5
fa, fb = 8, 9
ma, mb = fxyz[:, 0] == fa, fxyz[:, 0] == fb
A = EasyDict(fxyz=fxyz[ma].contiguous(), component=comp_all[ma] - comp_all[ma].min(),
             stationary=stat_all[ma], frame=fxyz[ma][:, 0].long())
B = EasyDict(fxyz=fxyz[mb].contiguous(), component=comp_all[mb], stationary=stat_all[mb],
             frame=fxyz[mb][:, 0].long())
C = int(A.component.max()) + 1
for radius, vs in ((2.5, [0.4, 0.4, 0.6]), (1.0, [0.1, 0.1, 0.15])):
    sampler = GridSampling3D(vs)
    sa, sb = CT.sample_frame(sampler, A), CT.sample_frame(sampler, B)
    mov = sa.fxyz[~sa.stationary].contiguous()
    mcomp = sa.component[~sa.stationary]
    ref = sb.fxyz[~sb.stationary].contiguous()
    deg = torch.bincount(sa.component.long(), minlength=C)
    print(f"r={radius}: C={C} Mn={mov.shape[0]} Rn={ref.shape[0]} Ra={sb.fxyz.shape[0]}", flush=True)
    for share in ((0,) if MASK_MODE else (0, 1)):
        for K in ((1, 2, 4, 8) if not MASK_MODE else (2, 4, 8)):
            streams = make_streams(K)
            res = [None] * K
            bar = threading.Barrier(K)

            def job(i, reps=6):
                with torch.cuda.stream(streams[i]):
                    if share:
                        # PCS_SHARE_PAIRS=1: launcher groups of two (threads 2g, 2g+1): K/2 streams of
                        # two-problem launches instead of one stream of K-problem launches
                        group = 1 + i // 2 if os.environ.get("PCS_SHARE_PAIRS") else 1
                        lib.pcs_icp_share_launches(ctypes.c_int(group))
                    ts = []
                    for rep in range(reps):
                        m = mov.clone()
                        streams[i].synchronize()
                        bar.wait()
                        t0 = time.perf_counter()
                        T, l1, ratio, iters = ops.register_to_next_frame(m, mcomp, ref, sb.fxyz, deg, C, radius,
                                                                         fb - fa, 10.0, 80, 0.05)
                        streams[i].synchronize()
                        ts.append(time.perf_counter() - t0)
                        bar.wait()
                    if share:
                        lib.pcs_icp_share_launches(ctypes.c_int(0))
                    res[i] = (min(ts), iters, T.clone())

            th = [threading.Thread(target=job, args=(i,)) for i in range(K)]
            [t.start() for t in th]
            [t.join() for t in th]
            # the per-component sums are f64 hardware atomics: their order, hence the last bits of T,
            # varies from call to call
            dev = max(float((res[0][2] - r[2]).abs().max()) for r in res)
            dt = max(r[0] for r in res)
            print(f"   share={share} K={K}: {dt * 1e3:7.2f} ms per round of K calls, {dt * 1e6 / res[0][1]:6.1f} us per "
                  f"iteration round, {dt * 1e6 / res[0][1] / K:5.1f} us per problem-iteration, iters={res[0][1]}, "
                  f"max |dT| across calls {dev:.1e}", flush=True)
