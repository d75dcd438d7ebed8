"""Pins the CPU oracle (oracle/pcs_oracle.cpp) to the reference:
 - against golden vectors produced by the reference's own compiled CPU code
   (tests/golden/make_golden.py), available everywhere;
 - against the compiled reference itself on fresh random inputs, where oracle/_ref exists.
CPU only."""
import glob
import os

import numpy as np
import pytest

from tests.golden.make_golden import RULEBOOK_CASES, random_indices


def _check_rulebook(got, want):
    out_ids, pairs, num = got
    r_out, r_pairs, r_num = want
    np.testing.assert_array_equal(num, r_num)
    np.testing.assert_array_equal(out_ids, r_out)
    np.testing.assert_array_equal(pairs, r_pairs)  # includes the -1 padding


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "rulebook_*.npz"))))
def test_rulebook_golden(oracle, path):
    g = np.load(path)
    got = oracle.get_indice_pairs(g["indices"], g["out_shape"], g["ksize"], g["stride"],
                                  g["padding"], g["dilation"], bool(g["subm"]))
    _check_rulebook(got, (g["out_ids"], g["pairs"], g["num"]))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "hardvox_*.npz"))))
def test_hard_voxelizer_golden(oracle, path):
    g = np.load(path)
    voxels, coords, num = oracle.hard_voxelize(g["points"], g["pc_range"], g["voxel_size"],
                                               int(g["max_points"]), int(g["max_voxels"]))
    np.testing.assert_array_equal(coords, g["coords"])
    np.testing.assert_array_equal(num, g["num"])
    np.testing.assert_array_equal(voxels, g["voxels"])


@pytest.mark.parametrize("case", RULEBOOK_CASES, ids=[c[0] for c in RULEBOOK_CASES])
def test_rulebook_vs_compiled_reference(oracle, case):
    if not oracle.ref_available():
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    name, shape, ks, st, pd, dl, subm = case
    rng = np.random.default_rng(abs(hash(name)) % 2**31)
    for n in (1, 50, 2500):
        idx = random_indices(rng, shape, n, batch=3)
        out_shape = shape if subm else oracle.conv_out_shape(shape, ks, st, pd, dl)
        want = oracle.ref_get_indice_pairs(idx, 3, out_shape, ks, st, pd, dl, subm)
        got = oracle.get_indice_pairs(idx, out_shape, ks, st, pd, dl, subm)
        _check_rulebook(got, want)


def test_hard_voxelizer_vs_compiled_reference(oracle):
    if not oracle.ref_available():
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    rng = np.random.default_rng(7)
    pts = np.concatenate([rng.normal(0, 2.0, (20000, 3)), rng.uniform(0, 1, (20000, 1))], axis=1).astype(np.float32)
    rngs = np.array([-6, -6, -1.5, 6, 6, 1.5], np.float32)
    vs = np.array([0.2, 0.2, 0.3], np.float32)
    for mp, mv in ((5, 100000), (2, 700)):
        want = oracle.ref_hard_voxelize(pts, rngs, vs, mp, mv)
        got = oracle.hard_voxelize(pts, rngs, vs, mp, mv)
        for a, b in zip(got, want):
            np.testing.assert_array_equal(a, b)


def test_dynamic_voxelize_matches_numpy_restatement(oracle):
    """dynamic_mean_vfe.py:51-76 restated with numpy (floor, mask, merge key, unique, mean)."""
    rng = np.random.default_rng(3)
    n = 5000
    pts = np.concatenate([rng.integers(0, 2, (n, 1)), rng.uniform(-5, 5, (n, 3)), rng.uniform(0, 1, (n, 2))], axis=1).astype(np.float32)
    rngs = np.array([-4, -4, -1, 4, 4, 1], np.float32)
    vs = np.array([0.5, 0.5, 0.25], np.float32)
    grid = np.round((rngs[3:] - rngs[:3]) / vs).astype(np.int64)
    feat, coords, inv = oracle.dynamic_voxelize(pts, rngs, vs, grid)
    c = np.floor((pts[:, 1:4] - rngs[:3]) / vs).astype(np.int32)
    mask = ((c >= 0) & (c < grid)).all(1)
    merge = pts[mask, 0].astype(np.int64) * grid.prod() + c[mask, 0] * grid[1] * grid[2] + c[mask, 1] * grid[2] + c[mask, 2]
    uq, uinv = np.unique(merge, return_inverse=True)
    assert feat.shape[0] == uq.shape[0]
    np.testing.assert_array_equal(inv[mask], uinv)
    assert (inv[~mask] == -1).all()
    want_coords = np.stack([uq // grid.prod(), uq % grid[2], (uq % (grid[1] * grid[2])) // grid[2],
                            (uq % grid.prod()) // (grid[1] * grid[2])], axis=1)
    np.testing.assert_array_equal(coords, want_coords)
    sums = np.zeros((uq.shape[0], 5), np.float64)
    np.add.at(sums, uinv, pts[mask, 1:].astype(np.float64))
    cnt = np.bincount(uinv, minlength=uq.shape[0])[:, None]
    np.testing.assert_allclose(feat, sums / cnt, rtol=1e-5, atol=1e-6)


def test_grid_sampling_properties(oracle):
    """GridSampling3D restatement: ids group points of one cell, voxel containment
    (voxel_modules.py:113-117 invariant) and idempotence of the inverse."""
    rng = np.random.default_rng(11)
    n = 8000
    pts = np.concatenate([rng.integers(0, 3, (n, 1)).astype(np.float32), rng.uniform(-10, 10, (n, 3)).astype(np.float32)], axis=1)
    means, inv, order, seg = oracle.grid_sample(pts, [0.4, 0.4, 0.6])
    size = np.array([1, 0.4, 0.4, 0.6], np.float32)
    start = pts.min(0); start[0] -= 0.5
    cell = np.floor((pts - start) / size).astype(np.int64)
    # same cell <=> same voxel
    _, cinv = np.unique(cell, axis=0, return_inverse=True)
    pairs = set(zip(cinv.tolist(), inv.tolist()))
    assert len(pairs) == len(set(cinv.tolist())) == means.shape[0]
    # mean lies inside the cell of its members
    mcell = np.floor((means - start) / size + 1e-4).astype(np.int64)
    assert (np.abs((means - start) / size - (cell[order[seg[:-1]]] + 0.5)) <= 0.5 + 1e-3).all()
    # sorted: voxel rank increases with id
    ids, _ = oracle.grid_cluster_ids(pts, size, 0.5)
    assert (np.diff(ids[order]) >= 0).all()
