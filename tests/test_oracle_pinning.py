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


@pytest.mark.parametrize("subm,stride,pad", [(True, 1, 1), (False, 2, 1), (False, 1, 0)])
def test_sparse_conv_forward_backward_vs_dense_autograd(oracle, subm, stride, pad):
    """The oracle's indice_conv / indice_conv_backward (spconv_ops.h:260-449) against torch's dense
    conv3d and its autograd: a sparse conv without bias is the dense conv of the zero-filled grid
    read at the output sites, and gradients flow only between active sites."""
    import torch
    rng = np.random.default_rng(4)
    shape, cin, cout = [6, 9, 10], 5, 7
    lin = rng.choice(int(np.prod(shape)), 120, replace=False)
    idx = np.concatenate([np.zeros((120, 1), np.int64), np.stack(np.unravel_index(lin, shape), 1)], 1).astype(np.int32)
    feat = rng.normal(size=(120, cin)).astype(np.float32)
    W = rng.normal(size=(3, 3, 3, cin, cout)).astype(np.float32)
    ks, st, pd = [3] * 3, [stride] * 3, [pad] * 3
    oshape = shape if subm else oracle.conv_out_shape(shape, ks, st, pd, [1, 1, 1])
    out_idx, pairs, num = oracle.get_indice_pairs(idx, oshape, ks, st, pd, [1, 1, 1], subm)
    out = oracle.indice_conv(feat, W, pairs, num, out_idx.shape[0], False, subm)
    og = rng.normal(size=out.shape).astype(np.float32)
    in_grad, w_grad = oracle.indice_conv_backward(feat, W, og, pairs, num, False, subm)
    # dense twin in f64
    dense = torch.zeros(1, cin, *shape, dtype=torch.float64)
    dense[0, :, idx[:, 1], idx[:, 2], idx[:, 3]] = torch.from_numpy(feat.T.astype(np.float64))
    dense.requires_grad_(True)
    Wt = torch.from_numpy(W.astype(np.float64)).permute(4, 3, 0, 1, 2).contiguous().requires_grad_(True)
    y = torch.nn.functional.conv3d(dense, Wt, stride=stride, padding=pad)
    assert list(y.shape[2:]) == list(oshape)
    o = out_idx.astype(np.int64)
    y_sites = y[0, :, o[:, 1], o[:, 2], o[:, 3]].T
    np.testing.assert_allclose(out, y_sites.detach().numpy(), rtol=1e-4, atol=1e-4)
    (y_sites * torch.from_numpy(og.astype(np.float64))).sum().backward()
    want_in = dense.grad[0, :, idx[:, 1], idx[:, 2], idx[:, 3]].T.numpy()
    want_w = Wt.grad.permute(2, 3, 4, 1, 0).numpy()
    np.testing.assert_allclose(in_grad, want_in, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(w_grad, want_w, rtol=1e-4, atol=1e-4)
    # inverse conv (outputs -> inputs over the same rulebook) is the adjoint gather / scatter
    x2 = rng.normal(size=(out_idx.shape[0], cout)).astype(np.float32)
    W2 = rng.normal(size=(3, 3, 3, cout, cin)).astype(np.float32)
    y2 = oracle.indice_conv(x2, W2, pairs, num, idx.shape[0], True, False)
    g2 = rng.normal(size=y2.shape).astype(np.float32)
    ig2, wg2 = oracle.indice_conv_backward(x2, W2, g2, pairs, num, True, False)
    # <y2, g2> is bilinear in (x2, W2): its gradients are the two partial contractions
    eps_x, eps_w = rng.normal(size=x2.shape).astype(np.float32), rng.normal(size=W2.shape).astype(np.float32)
    lhs_x = float((oracle.indice_conv(eps_x, W2, pairs, num, idx.shape[0], True, False) * g2).sum())
    lhs_w = float((oracle.indice_conv(x2, eps_w, pairs, num, idx.shape[0], True, False) * g2).sum())
    assert abs(lhs_x - float((ig2 * eps_x).sum())) < 1e-2 * max(1.0, abs(lhs_x))
    assert abs(lhs_w - float((wg2 * eps_w).sum())) < 1e-2 * max(1.0, abs(lhs_w))


@pytest.mark.parametrize("name", ["maxpool_ties", "maxpool_float"])
def test_maxpool_oracle_vs_reference(oracle, name):
    """oracle.indice_maxpool / _backward against the reference's own CPU functors: committed golden
    vectors (tests/golden/make_golden_maxpool.py) and, when oracle/_ref is built, the compiled
    functors on fresh random data. Bit-exact."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    out = oracle.indice_maxpool(g["features"], g["pairs"], g["num"], g["out_ids"].shape[0])
    np.testing.assert_array_equal(out, g["out"])
    din = oracle.indice_maxpool_backward(g["features"], out, g["out_grad"], g["pairs"], g["num"])
    np.testing.assert_array_equal(din, g["in_grad"])
    lib = oracle.ref_maxpool()
    if lib is None:
        return
    from tests.golden.make_golden_maxpool import make_case, reference_maxpool
    c = make_case(99 if name == "maxpool_ties" else 98, name == "maxpool_ties")
    want_out, want_din = reference_maxpool(lib, c["features"], c["pairs"], c["num"], c["out_ids"].shape[0], c["out_grad"])
    got = oracle.indice_maxpool(c["features"], c["pairs"], c["num"], c["out_ids"].shape[0])
    np.testing.assert_array_equal(got, want_out)
    np.testing.assert_array_equal(oracle.indice_maxpool_backward(c["features"], got, c["out_grad"], c["pairs"], c["num"]), want_din)


from tests.golden.make_golden_deconv import DECONV_CASES  # noqa: E402


@pytest.mark.parametrize("case", DECONV_CASES, ids=[c[0] for c in DECONV_CASES])
def test_deconv_rulebook_vs_compiled_reference_and_golden(oracle, case):
    """Transposed-conv rulebook: the oracle against the reference's own getIndicePairsDeConv compiled in
    place (where oracle/_ref exists; that run also (re)writes nothing) and against the committed golden
    vectors it produced (tests/golden/deconvbook_*.npz)."""
    name, shape, ks, st, pd, dl, opad = case
    out_shape = oracle.deconv_out_shape(shape, ks, st, pd, dl, opad)
    rng = np.random.default_rng(abs(hash(name)) % 2**31)
    if oracle.ref_available():
        for n in (1, 40, 1500):
            idx = random_indices(rng, shape, n, batch=3)
            want = oracle.ref_get_indice_pairs_deconv(idx, 3, out_shape, ks, st, pd, dl)
            got = oracle.get_indice_pairs(idx, out_shape, ks, st, pd, dl, False, transpose=True)
            _check_rulebook(got, want)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", f"deconvbook_{name}.npz"))
    got = oracle.get_indice_pairs(g["indices"], g["out_shape"], g["ksize"], g["stride"], g["padding"],
                                  g["dilation"], False, transpose=True)
    _check_rulebook(got, (g["out_ids"], g["pairs"], g["num"]))
