"""Golden vectors for VoxelAggregation, produced by IMPORTING the reference's own module
(pcdet/ops/voxel/voxel_modules.py) in the authoring container and running its forward on CPU tensors:

  voxel_agg_free.npz    no POINT_CLOUD_RANGE (grid anchored at the data minimum), medians of two
                        integer attributes (one with min 0, one with min 3)
  voxel_agg_range.npz   with POINT_CLOUD_RANGE (points outside the range are dropped), two batches

torch_scatter (absent from this image) is provided in memory for the import only (scatter ->
scatter_add_ / count, the documented semantics); see make_golden_reg.py. The reference never travels:
only inputs and outputs are stored. Run: python tests/golden/make_golden_voxel_agg.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
from tests.golden.make_golden_reg import _install_shims  # noqa: E402

CFG_FREE = dict(VOXEL_SIZE=[0.4, 0.4, 0.3])
CFG_RANGE = dict(VOXEL_SIZE=[0.5, 0.5, 0.25], POINT_CLOUD_RANGE=[-8.0, -6.0, -1.0, 8.0, 6.0, 2.0])


def make_points(seed, n, batches):
    rng = np.random.default_rng(seed)
    b = rng.integers(0, batches, n).astype(np.float32)
    xyz = np.concatenate([rng.uniform(-10, 10, (n, 2)), rng.uniform(-1.5, 2.5, (n, 1))], 1).astype(np.float32)
    # clusters, so that voxels hold several points
    xyz[: n // 2] = (xyz[rng.integers(0, 40, n // 2)] + rng.normal(0, 0.15, (n // 2, 3))).astype(np.float32)
    feat = rng.normal(0, 1, (n, 3)).astype(np.float32)
    lab0 = rng.integers(0, 7, n).astype(np.int64)
    lab3 = rng.integers(3, 23, n).astype(np.int64)
    return np.concatenate([b[:, None], xyz], 1), feat, lab0, lab3


def main():
    _install_shims()
    spec = importlib.util.spec_from_file_location("ref_voxel_modules", f"{REF}/pcdet/ops/voxel/voxel_modules.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for name, cfg, batches, seed in (("voxel_agg_free", CFG_FREE, 1, 3), ("voxel_agg_range", CFG_RANGE, 2, 4)):
        bxyz, feat, lab0, lab3 = make_points(seed, 6000, batches)
        va = mod.VoxelAggregation(dict(cfg), {})
        mean_in = dict(point_bxyz=torch.from_numpy(bxyz), point_feat=torch.from_numpy(feat))
        med_in = dict(label0=torch.from_numpy(lab0), label3=torch.from_numpy(lab3))
        vw, pw, nv, oob = va(mean_in, med_in)
        out = dict(point_bxyz=bxyz, point_feat=feat, label0=lab0, label3=lab3, num_voxels=np.int64(nv),
                   out_of_boundary_mask=oob.numpy())
        for k, v in vw.items():
            out["vw_" + k] = v.numpy()
        for k, v in pw.items():
            out["pw_" + k] = v.numpy()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "voxels", nv, "dropped", int(oob.sum()), sorted(out.keys()))


if __name__ == "__main__":
    main()
