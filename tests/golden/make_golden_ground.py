"""Golden vectors for the ground stage, produced by IMPORTING the reference's own
pcdet/models/registration/preprocessors/preprocessor_utils.py in the authoring container and
running `ground_plane_removal` (grid_sample, format_pillars, compute_min_height_from_ransac with
its 30 x iterative_reweighted_ransac, the "truncated least squares" pruning, the propagation loop,
l1_minimization with torch.optim.AdamW) on CPU tensors of a small synthetic sequence.

Modules absent from this image are provided in memory for the import only, as in
make_golden_track.py: torch_scatter, easydict; torch_cluster.knn (brute force: the k nearest by
fp32 squared distance, ties by index -- upstream's tie order is unspecified); GridSampling3D (the
oracle's restatement, pinned separately). The reference's `.cuda()` calls are neutralised for the
duration of the run (torch.Tensor.cuda / nn.Module.cuda -> identity). The reference never travels:
only inputs and outputs are stored.

  ground_small.npz  point_fxyz [n,4] of the 8 cm subsample + point_height / point_horizon /
                    point_error / pillar_height / pillar_min_z as the reference returns them, and the
                    intermediate hypotheses (super-pillar plane centres / normals / hit counts before
                    and after pruning) captured by wrapping the leaves the reference calls
Run: python tests/golden/make_golden_ground.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from tests.golden import make_golden_reg as MG  # noqa: E402
from tests.golden.make_golden_track import OracleGridSampling3D  # noqa: E402
from pcseqlearning_amd import synthetic  # noqa: E402

GROUND_CFG = dict(PILLAR_SIZE=[2, 2], RANSAC=True, SIGMA2=0.0025, K=8, JointOpt=True, LR=0.01,
                  DECAY_STEPS=[1600], RIGID_WEIGHT=0.5, MAX_NUM_ITERS=10000, TRUNCATE_HEIGHT=[0.5])


def knn_brute(x, y, k):
    """torch_cluster.knn(x, y, k): for every row of y its k nearest rows of x, as [2, len(y) * k]
    (row 0 = index into y, row 1 = index into x), grouped by query."""
    d = ((y[:, None, :] - x[None, :, :]) ** 2).sum(-1)
    nbr = torch.argsort(d, dim=1, stable=True)[:, :k]
    q = torch.arange(y.shape[0])[:, None].expand(-1, nbr.shape[1])
    return torch.stack([q.reshape(-1), nbr.reshape(-1)], 0)


def load_reference_ground():
    EasyDict = MG._install_shims()
    tc = types.ModuleType("torch_cluster")
    tc.knn = knn_brute
    tc.grid_cluster = None
    sys.modules["torch_cluster"] = tc
    for name in ("pcdet.models", "pcdet.models.model_utils", "pcdet.ops", "pcdet.ops.voxel"):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    gs = types.ModuleType("pcdet.models.model_utils.grid_sampling")
    gs.GridSampling3D = OracleGridSampling3D
    sys.modules["pcdet.models.model_utils.grid_sampling"] = gs
    vm = types.ModuleType("pcdet.ops.voxel.voxel_modules")
    vm.VoxelAggregation = object
    sys.modules["pcdet.ops.voxel.voxel_modules"] = vm
    pkg = types.ModuleType("refpre")
    pkg.__path__ = [f"{REF}/pcdet/models/registration/preprocessors"]
    sys.modules["refpre"] = pkg
    spec = importlib.util.spec_from_file_location(
        "refpre.preprocessor_utils", f"{REF}/pcdet/models/registration/preprocessors/preprocessor_utils.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules["refpre.preprocessor_utils"] = mod
    spec.loader.exec_module(mod)
    return mod, EasyDict


def small_sequence(seed=11, num_frames=4, n_target=16000):
    """8 cm subsample of a small synthetic sequence (ground included), as SimpleReg hands it on."""
    batch = synthetic.gen_sequence(seed, num_frames, n_target, n_static=10, n_vehicle=6, n_ped=3)
    fxyz = np.concatenate([batch["point_sweep"].astype(np.float32), batch["point_bxyz"][:, 1:]], 1)
    _, _, order, seg = O.grid_sample(fxyz, [0.08, 0.08, 0.08])
    pick = order[seg[1:] - 1]
    return fxyz[pick], batch["segmentation_label"][pick]


def main():
    mod, EasyDict = load_reference_ground()
    fxyz, seg_label = small_sequence()
    print("points", fxyz.shape)
    cfg = EasyDict(GROUND_CFG)
    captured = {}
    # the reference keeps its intermediate hypotheses in locals: capture them at the leaves it calls
    orig_knn = mod.knn
    calls = []

    def knn_spy(x, y, k):
        calls.append(x.detach().clone())
        return orig_knn(x, y, k)

    mod.knn = knn_spy
    orig_irls = mod.iterative_reweighted_ransac
    irls_rounds = []

    def irls_spy(points, pillars, w0, num_pillars, sigma2, stopping_delta=1e-2):
        out = orig_irls(points, pillars, w0, num_pillars, sigma2, stopping_delta)
        irls_rounds.append((out[1].center.detach().clone(), out[1].normal.detach().clone(),
                            out[0].plane_fitting_error.detach().clone()))
        return out

    mod.iterative_reweighted_ransac = irls_spy
    t_cuda, m_cuda = torch.Tensor.cuda, nn.Module.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    try:
        with torch.enable_grad():
            height, horizon, err, p_height, p_min_z = mod.ground_plane_removal(torch.from_numpy(fxyz.copy()), cfg)
    finally:
        torch.Tensor.cuda, nn.Module.cuda = t_cuda, m_cuda
    captured["hyp_xyz_first"] = calls[0].numpy()                 # hypotheses entering the pruning
    captured["hyp_xyz_last"] = calls[-1].numpy()                 # ... and the set of the last k-NN call
    captured["irls_center_r0"] = irls_rounds[0][0].numpy()       # first and last of the 30 ratio fits
    captured["irls_normal_r0"] = irls_rounds[0][1].numpy()
    captured["irls_center_r29"] = irls_rounds[-1][0].numpy()
    captured["irls_normal_r29"] = irls_rounds[-1][1].numpy()
    neg = [(r[1][:, 2] < 0).float().mean().item() for r in irls_rounds]
    print(f"k-NN calls {len(calls)}, hypotheses {calls[0].shape[0]} -> {calls[-1].shape[0]}, "
          f"share of fitted normals with n_z < 0 per ratio: min {min(neg):.3f} max {max(neg):.3f}")
    h = height.detach().numpy()
    print("kept (height >= 0.5):", int((h >= 0.5).sum()), "of", h.size,
          "| ground-labelled kept:", int(((h >= 0.5) & (seg_label == 17)).sum()))
    np.savez_compressed(os.path.join(HERE, "ground_small.npz"), point_fxyz=fxyz, seg_label=seg_label,
                        point_height=h, point_horizon=horizon.numpy(), point_error=err.detach().numpy(),
                        pillar_height=p_height.detach().numpy(), pillar_min_z=p_min_z.detach().numpy(),
                        neg_normal_share=np.array(neg, np.float32), **captured)


if __name__ == "__main__":
    main()
