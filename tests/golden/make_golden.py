"""Generates the golden vectors under tests/golden/ from the REFERENCE's own code, in the
authoring container (where /root/reference exists):

  rulebook_*.npz   inputs + outputs of the reference CPU rulebook builders
                   (pcdet/ops/spconv/include/spconv/geometry.h, compiled by oracle/Makefile into
                   oracle/_ref/libref_spconv.so)
  hardvox_*.npz    inputs + outputs of the reference hard voxelizer
                   (pcdet/ops/spconv/include/spconv/point2voxel.h, oracle/_ref/ref_point2voxel*.so)

Fixtures are data only (inputs and expected outputs). Run: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

# (name, spatial_shape zyx, ksize, stride, padding, dilation, subm)  -- the VoxelBackBone8x layer
# geometries (spconv_backbone.py:55-95) on a reduced grid, plus edge cases
RULEBOOK_CASES = [
    ("subm3", [11, 40, 40], [3, 3, 3], [1, 1, 1], [1, 1, 1], [1, 1, 1], True),
    ("spconv_s2p1", [11, 40, 40], [3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1], False),
    ("spconv_s2p011", [11, 40, 40], [3, 3, 3], [2, 2, 2], [0, 1, 1], [1, 1, 1], False),
    ("spconv_k311", [5, 24, 24], [3, 1, 1], [2, 1, 1], [0, 0, 0], [1, 1, 1], False),
    ("spconv_s1p0", [7, 12, 12], [3, 3, 3], [1, 1, 1], [0, 0, 0], [1, 1, 1], False),
    ("subm_k133", [6, 20, 20], [1, 3, 3], [1, 1, 1], [0, 1, 1], [1, 1, 1], True),
]


def random_indices(rng, shape, n, batch):
    vol = int(np.prod(shape))
    out = []
    for b in range(batch):
        flat = rng.choice(vol, size=min(n, vol), replace=False)
        z, r = np.divmod(flat, shape[1] * shape[2])
        y, x = np.divmod(r, shape[2])
        out.append(np.stack([np.full_like(z, b), z, y, x], axis=1))
    idx = np.concatenate(out).astype(np.int32)
    return idx[rng.permutation(idx.shape[0])]


def main():
    assert O.ref_available(), "oracle/_ref not built (needs /root/reference)"
    rng = np.random.default_rng(12345)
    for name, shape, ks, st, pd, dl, subm in RULEBOOK_CASES:
        idx = random_indices(rng, shape, 600, batch=2)
        out_shape = shape if subm else O.conv_out_shape(shape, ks, st, pd, dl)
        out_ids, pairs, num = O.ref_get_indice_pairs(idx, 2, out_shape, ks, st, pd, dl, subm)
        # store pairs compactly: only the first num[k] entries per offset
        np.savez_compressed(os.path.join(HERE, f"rulebook_{name}.npz"), indices=idx,
                            spatial_shape=np.array(shape), out_shape=np.array(out_shape),
                            ksize=np.array(ks), stride=np.array(st), padding=np.array(pd),
                            dilation=np.array(dl), subm=np.array(int(subm)), out_ids=out_ids,
                            pairs=pairs, num=num)
        print(name, "V", idx.shape[0], "out", out_ids.shape[0], "pairs", int(num.sum()))
    # hard voxelizer: clustered points so voxels fill up, some out of range, cap on voxels
    for name, n, max_points, max_voxels in [("a", 4000, 5, 150000), ("capped", 4000, 3, 500)]:
        pts = np.concatenate([rng.normal(0, 3.0, (n, 3)), rng.uniform(0, 1, (n, 2))], axis=1).astype(np.float32)
        pc_range = np.array([-8, -8, -2, 8, 8, 4], np.float32)
        vs = np.array([0.4, 0.4, 0.6], np.float32)
        voxels, coords, num = O.ref_hard_voxelize(pts, pc_range, vs, max_points, max_voxels)
        np.savez_compressed(os.path.join(HERE, f"hardvox_{name}.npz"), points=pts, pc_range=pc_range,
                            voxel_size=vs, max_points=np.array(max_points),
                            max_voxels=np.array(max_voxels), voxels=voxels, coords=coords, num=num)
        print("hardvox", name, voxels.shape)


if __name__ == "__main__":
    main()
