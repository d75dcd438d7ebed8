"""Where the time of icp_nn_kernel goes: the search cut short after stage s (PCS_NN_PROBE_STAGE=s:
1 launch + head loads, 2 + frame range and coarse masks, 3 + near cells, 0 the full search), timed
per launch with dispatch-attached HIP events on real frame pairs, alone on the chip. Results of the
cut runs are meaningless; only their launch time is read.

  python tools/icp_stages.py --prepare /tmp/icp_probe.pt
  for s in 1 2 3 0; do PCS_NN_PROBE_STAGE=$s python tools/icp_stages.py /tmp/icp_probe.pt; done
  # the apply launch (stage 11 launch + grid reset + head loads, 12 + moments / transform loads and the
  # solve, 13 + move and stores, 0 + grid insert and loss reduction) and the moments launch:
  for s in 11 12 13 0; do PCS_PROF_KERNEL=apply PCS_NN_PROBE_STAGE=$s python tools/icp_stages.py /tmp/icp_probe.pt; done
  PCS_PROF_KERNEL=moments python tools/icp_stages.py /tmp/icp_probe.pt
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pcseqlearning_amd import lib as L, ops  # noqa: E402


def prepare(path):
    import bench
    from pcseqlearning_amd.pcdet import models
    from pcseqlearning_amd.pcdet.models.model_utils.grid_sampling import GridSampling3D
    from pcseqlearning_amd.pcdet.models.registration.preprocessors import cluster_tracking as CT
    from pcseqlearning_amd.pcdet.utils.common_utils import EasyDict
    from tools.run_sequence import build_model
    batch = models.load_data_to_gpu(bench.load_sequence(0, 20, 160000))
    model = build_model(streams=1)
    with torch.no_grad():
        model(batch)
    out = model.last_outputs[0]
    fxyz = out["point_fxyz"]
    comp_all = out["point_component_rad1x25"]
    diam = CT.component_diameter(EasyDict(fxyz=fxyz, component=comp_all))
    stat_all = diam[comp_all] > 12.5
    fa, fb = 8, 9
    ma, mb = fxyz[:, 0] == fa, fxyz[:, 0] == fb
    A = EasyDict(fxyz=fxyz[ma].contiguous(), component=comp_all[ma] - comp_all[ma].min(),
                 stationary=stat_all[ma], frame=fxyz[ma][:, 0].long())
    B = EasyDict(fxyz=fxyz[mb].contiguous(), component=comp_all[mb], stationary=stat_all[mb],
                 frame=fxyz[mb][:, 0].long())
    C = int(A.component.max()) + 1
    probs = []
    for radius, vs in ((2.5, [0.4, 0.4, 0.6]), (1.25, [0.2, 0.2, 0.3]), (1.0, [0.1, 0.1, 0.15])):
        sampler = GridSampling3D(vs)
        sa, sb = CT.sample_frame(sampler, A), CT.sample_frame(sampler, B)
        probs.append(dict(radius=radius, C=C, mov=sa.fxyz[~sa.stationary].contiguous().cpu(),
                          mcomp=sa.component[~sa.stationary].cpu(), ref=sb.fxyz[~sb.stationary].contiguous().cpu(),
                          ref_all=sb.fxyz.cpu(), deg=torch.bincount(sa.component.long(), minlength=C).cpu()))
    torch.save(probs, path)
    print("saved", [(p["radius"], p["mov"].shape[0], p["ref"].shape[0]) for p in probs])


def main():
    if sys.argv[1] == "--prepare":
        return prepare(sys.argv[2])
    stage = os.environ.get("PCS_NN_PROBE_STAGE", "0")
    lib = L.load()
    probs = torch.load(sys.argv[1])
    for p in probs:
        mov, mcomp, ref, ref_all, deg = (p[k].cuda() for k in ("mov", "mcomp", "ref", "ref_all", "deg"))
        best = None
        for rep in range(4):
            m = mov.clone()
            lib.pcs_profile_icp_nn(ctypes.c_int(1))
            torch.cuda.synchronize()
            _, _, _, iters = ops.register_to_next_frame(m, mcomp, ref, ref_all, deg, p["C"], p["radius"], 1, 10.0, 80, 0.05)
            torch.cuda.synchronize()
            prof = (ctypes.c_double * 3)()
            lib.pcs_profile_icp_nn_read(prof)
            lib.pcs_profile_icp_nn(ctypes.c_int(0))
            us = prof[0] * 1e3 / max(prof[1], 1)
            best = us if best is None else min(best, us)
        which = {"m": "icp_moments_kernel", "a": "icp_apply_kernel"}.get(os.environ.get("PCS_PROF_KERNEL", "n")[0], "icp_nn_kernel")
        print(f"stage {stage} r={p['radius']}: Mn={mov.shape[0]} Rn={ref.shape[0]}  {which} {best:6.2f} us per launch "
              f"({int(prof[1])} launches timed, {iters} iterations)", flush=True)


if __name__ == "__main__":
    main()
