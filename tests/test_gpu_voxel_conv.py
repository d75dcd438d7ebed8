"""Parity of the HIP voxelizers / rulebook / sparse conv (through the C ABI) against the CPU
oracle and the golden vectors. Integer outputs bit-exact; conv within 1e-5 (fp32 accumulation
order differs from a BLAS GEMM, the tolerance the reference itself would see)."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from pcseqlearning_amd import ops, synthetic  # noqa: E402

DEV = "cuda:0"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _rulebook_gpu(idx, batch, out_shape, shape, ks, st, pd, dl, subm):
    out_ids, pairs, num = ops.get_indice_pairs(dev(idx), batch, list(out_shape), list(shape),
                                               list(ks), list(st), list(pd), list(dl), subm)
    return out_ids.cpu().numpy(), pairs.cpu().numpy(), num.cpu().numpy()


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "rulebook_*.npz"))))
def test_rulebook_golden(path):
    g = np.load(path)
    out_ids, pairs, num = _rulebook_gpu(g["indices"], 2, g["out_shape"], g["spatial_shape"],
                                        g["ksize"], g["stride"], g["padding"], g["dilation"],
                                        bool(g["subm"]))
    np.testing.assert_array_equal(num, g["num"])
    np.testing.assert_array_equal(out_ids, g["out_ids"])
    np.testing.assert_array_equal(pairs, g["pairs"])


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "hardvox_*.npz"))))
def test_hard_voxelizer_golden(path):
    g = np.load(path)
    voxels, coords, num = ops.hard_voxelize(dev(g["points"]), g["pc_range"], g["voxel_size"],
                                            int(g["max_points"]), int(g["max_voxels"]))
    np.testing.assert_array_equal(coords.cpu().numpy(), g["coords"])
    np.testing.assert_array_equal(num.cpu().numpy(), g["num"])
    np.testing.assert_array_equal(voxels.cpu().numpy(), g["voxels"])


def test_hard_voxelizer_edge_cases(oracle):
    rngs = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    vs = np.array([0.5, 0.5, 0.5], np.float32)
    # every point out of range -> zero voxels; single point; duplicates
    for pts in (np.full((10, 4), 5.0, np.float32), np.zeros((1, 4), np.float32),
                np.zeros((9, 4), np.float32)):
        v, c, n = ops.hard_voxelize(dev(pts), rngs, vs, 4, 10)
        ov, oc, on = oracle.hard_voxelize(pts, rngs, vs, 4, 10)
        np.testing.assert_array_equal(v.cpu().numpy(), ov)
        np.testing.assert_array_equal(c.cpu().numpy(), oc)
        np.testing.assert_array_equal(n.cpu().numpy(), on)


def test_grid_sampling_vs_oracle(oracle):
    rng = np.random.default_rng(5)
    for n, gs in ((1, [0.4, 0.4, 0.6]), (777, [0.2, 0.2, 0.3]), (200000, [0.08, 0.08, 0.08])):
        pts = np.concatenate([rng.integers(0, 4, (n, 1)).astype(np.float32),
                              rng.uniform(-30, 30, (n, 3)).astype(np.float32)], axis=1)
        o_means, o_inv, o_order, o_seg = oracle.grid_sample(pts, gs)
        p = dev(pts)
        ids = ops.grid_cluster_ids(p, [1.0] + gs, 0.5)
        o_ids, _ = oracle.grid_cluster_ids(pts, [1.0] + gs, 0.5)
        np.testing.assert_array_equal(ids.cpu().numpy(), o_ids)
        seg = ops.unique_i64(ids)
        assert seg.num == o_means.shape[0]
        np.testing.assert_array_equal(seg.inv.cpu().numpy(), o_inv)
        np.testing.assert_array_equal(seg.order.cpu().numpy(), o_order)
        np.testing.assert_array_equal(seg.seg_start.cpu().numpy(), o_seg)
        means = ops.segment_reduce(p, seg, mean=True)
        np.testing.assert_array_equal(means.cpu().numpy(), o_means)  # same summation order
        last = ops.segment_pick(seg, last=True).cpu().numpy()
        np.testing.assert_array_equal(last, o_order[o_seg[1:] - 1])


def test_dynamic_voxelize_vs_oracle(oracle):
    pts = synthetic.gen_voxel_frame(1, 50000)
    pts[:100, 1] = 1000.0  # out of range rows
    feat, coords, inv = ops.dynamic_voxelize(dev(pts), synthetic.PC_RANGE, synthetic.VOXEL_SIZE,
                                             synthetic.GRID_SIZE)
    o_feat, o_coords, o_inv = oracle.dynamic_voxelize(pts, synthetic.PC_RANGE,
                                                      synthetic.VOXEL_SIZE, synthetic.GRID_SIZE)
    np.testing.assert_array_equal(coords.cpu().numpy(), o_coords)
    np.testing.assert_array_equal(inv.cpu().numpy(), o_inv)
    np.testing.assert_array_equal(feat.cpu().numpy(), o_feat)


BACKBONE_LAYERS = [  # (ksize, stride, padding, subm) as spconv_backbone.py:55-95
    ([3, 3, 3], [1, 1, 1], [1, 1, 1], True),
    ([3, 3, 3], [2, 2, 2], [1, 1, 1], False),
    ([3, 3, 3], [2, 2, 2], [0, 1, 1], False),
    ([3, 1, 1], [2, 1, 1], [0, 0, 0], False),
]


def test_rulebook_and_conv_full_frame(oracle):
    """BASELINE config 2: one 160k-pt frame -> dynamic voxelization -> rulebooks of every layer
    geometry of VoxelBackBone8x -> conv, bit-exact indices vs the oracle, conv within 1e-5."""
    pts = synthetic.gen_voxel_frame(0, 160000)
    feat, coords, _ = ops.dynamic_voxelize(dev(pts), synthetic.PC_RANGE, synthetic.VOXEL_SIZE,
                                           synthetic.GRID_SIZE)
    shape = [41, 1504, 1504]
    idx = coords.cpu().numpy()
    rng = np.random.default_rng(7)
    for ks, st, pd, subm in BACKBONE_LAYERS:
        dl = [1, 1, 1]
        out_shape = shape if subm else oracle.conv_out_shape(shape, ks, st, pd, dl)
        got = _rulebook_gpu(idx, 1, out_shape, shape, ks, st, pd, dl, subm)
        want = oracle.get_indice_pairs(idx, out_shape, ks, st, pd, dl, subm)
        np.testing.assert_array_equal(got[2], want[2])
        np.testing.assert_array_equal(got[0], want[0])
        np.testing.assert_array_equal(got[1], want[1])
        # conv on this rulebook
        cin, cout = 5, 16
        W = rng.normal(0, 0.05, (*ks, cin, cout)).astype(np.float32)
        num_out = got[0].shape[0]
        o = ops.indice_conv(feat, dev(W), dev(got[1]), dev(got[2]), num_out, False, subm)
        ref = oracle.indice_conv(feat.cpu().numpy(), W, want[1], want[2], num_out, False, subm)
        np.testing.assert_allclose(o.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("ks", [[3, 3, 3], [3, 1, 1], [1, 3, 5], [2, 3, 3]])
def test_submanifold_rulebook_leaves_the_neighbour_table(oracle, ks):
    """The submanifold rulebook's own probing pass writes the conv kernels' neighbour table: the same table
    pcs_build_neighbor_table makes of the pairs; the pairs themselves stay bit-exact against the oracle.
    Even kernel sizes have no mirrored offset: no table (the caller builds it)."""
    pts = synthetic.gen_voxel_frame(3, 60000)
    _, coords, _ = ops.dynamic_voxelize(dev(pts), synthetic.PC_RANGE, synthetic.VOXEL_SIZE, synthetic.GRID_SIZE)
    shape = [41, 1504, 1504]
    idx = coords.cpu().numpy()
    idx = idx[np.random.default_rng(2).permutation(len(idx))]  # (not in key order)
    dl, st = [1, 1, 1], [1, 1, 1]
    pd = [k // 2 for k in ks]
    out_ids, pairs, num, table = ops.get_indice_pairs(dev(idx), 1, shape, shape, ks, st, pd, dl, True, with_table=True)
    want = oracle.get_indice_pairs(idx, shape, ks, st, pd, dl, True)
    np.testing.assert_array_equal(num.cpu().numpy(), want[2])
    np.testing.assert_array_equal(pairs.cpu().numpy(), want[1])
    if any(k % 2 == 0 for k in ks):
        assert table is None
        return
    built = ops.build_neighbor_table(pairs, num, idx.shape[0])
    np.testing.assert_array_equal(table.cpu().numpy(), built.cpu().numpy())
    assert (table >= 0).sum().item() == int(want[2].sum())


def test_submanifold_rulebook_with_duplicate_coordinates(oracle):
    """Duplicate input coordinates (the reference's grid keeps the LAST index, geometry.h:262-266): pairs
    bit-exact against the oracle; a voxel that lost its coordinate is never an output, so its row of the
    neighbour table is empty, and the winners' rows are what pcs_build_neighbor_table makes of the pairs."""
    rng = np.random.default_rng(4)
    shape = [12, 30, 30]
    base = np.unique(np.stack([rng.integers(0, 2, 3000), rng.integers(0, 12, 3000), rng.integers(0, 30, 3000),
                               rng.integers(0, 30, 3000)], 1), axis=0).astype(np.int32)
    idx = np.concatenate([base, base[rng.permutation(len(base))[:400]]])
    idx = idx[rng.permutation(len(idx))]
    ks, st, pd, dl = [3, 3, 3], [1, 1, 1], [1, 1, 1], [1, 1, 1]
    out_ids, pairs, num, table = ops.get_indice_pairs(dev(idx), 2, shape, shape, ks, st, pd, dl, True, with_table=True)
    want = oracle.get_indice_pairs(idx, shape, ks, st, pd, dl, True)
    np.testing.assert_array_equal(num.cpu().numpy(), want[2])
    np.testing.assert_array_equal(pairs.cpu().numpy(), want[1])
    t = table.cpu().numpy()
    keys = (idx[:, 0].astype(np.int64) * 12 + idx[:, 1]) * 900 + idx[:, 2] * 30 + idx[:, 3]
    last = {}
    for j, k in enumerate(keys):
        last[k] = j
    loser = np.array([last[k] != j for j, k in enumerate(keys)])
    assert loser.sum() == 400
    assert (t[loser] == -1).all()
    assert (t[~loser, 13] == np.nonzero(~loser)[0]).all()  # every winner is its own centre partner


@pytest.mark.parametrize("case", ["crowded", "isolated", "grid_beyond_32_bits", "one_voxel", "no_valid_output"])
def test_strided_rulebook_table_regimes(oracle, case):
    """The strided rulebook's three table regimes, bit-exact against the oracle: a first table of 4-8 slots
    per input (crowded: adjacent voxels, 0.2 outputs per input), the retry with two slots per POSSIBLE output
    (isolated odd coordinates: 8 outputs per input), and the two-array form of grids whose keys pass 32 bits."""
    rng = np.random.default_rng(11)
    ks, st, pd, dl = [3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1]
    batch = 3
    if case == "crowded":
        shape = [24, 40, 40]
        full = np.stack(np.meshgrid(np.arange(batch), np.arange(24), np.arange(40), np.arange(40), indexing="ij"), -1).reshape(-1, 4)
        idx = full[rng.permutation(len(full))[: len(full) // 2]].astype(np.int32)
    elif case == "isolated":
        shape = [41, 400, 400]
        n = 30000
        idx = np.unique(np.stack([rng.integers(0, batch, n), 4 * rng.integers(0, 10, n) + 1, 4 * rng.integers(1, 99, n) + 1,
                                  4 * rng.integers(1, 99, n) + 1], 1), axis=0).astype(np.int32)
        idx = idx[rng.permutation(len(idx))]
    elif case == "grid_beyond_32_bits":
        shape = [400, 4000, 4000]  # 200 x 2000 x 2000 outputs x 8 batches = 6.4e9 keys
        batch = 8
        n = 20000
        idx = np.unique(np.stack([rng.integers(0, batch, n), rng.integers(0, 400, n), rng.integers(0, 4000, n),
                                  rng.integers(0, 4000, n)], 1), axis=0).astype(np.int32)
        idx = idx[rng.permutation(len(idx))]
    elif case == "one_voxel":
        shape = [8, 8, 8]
        idx = np.array([[1, 3, 3, 3]], np.int32)
    else:  # every emission falls outside the output grid
        shape = [8, 8, 8]
        ks, st, pd = [1, 1, 1], [4, 4, 4], [0, 0, 0]
        idx = np.array([[0, 1, 1, 1], [2, 3, 2, 1]], np.int32)
    out_shape = oracle.conv_out_shape(shape, ks, st, pd, dl)
    got = _rulebook_gpu(idx, batch, out_shape, shape, ks, st, pd, dl, False)
    want = oracle.get_indice_pairs(idx, out_shape, ks, st, pd, dl, False)
    np.testing.assert_array_equal(got[2], want[2])
    np.testing.assert_array_equal(got[0], want[0])
    np.testing.assert_array_equal(got[1], want[1])
    if case == "isolated":
        assert got[0].shape[0] == 8 * idx.shape[0]
    if case == "no_valid_output":
        assert got[0].shape[0] == 0


@pytest.mark.parametrize("cin,cout", [(16, 16), (16, 32), (32, 32), (32, 64), (64, 64), (64, 128), (128, 128), (128, 64), (96, 32), (7, 10)])
def test_conv_channel_configs(oracle, cin, cout):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "rulebook_subm3.npz"))
    rng = np.random.default_rng(cin * 131 + cout)
    V = g["indices"].shape[0]
    feat = rng.normal(0, 1, (V, cin)).astype(np.float32)
    W = rng.normal(0, 0.05, (3, 3, 3, cin, cout)).astype(np.float32)
    o = ops.indice_conv(dev(feat), dev(W), dev(g["pairs"]), dev(g["num"]), V, False, True)
    ref = oracle.indice_conv(feat, W, g["pairs"], g["num"], V, False, True)
    np.testing.assert_allclose(o.cpu().numpy(), ref, rtol=1e-5, atol=2e-5)
    # bitwise reproducible run to run (no atomics on the data path)
    o2 = ops.indice_conv(dev(feat), dev(W), dev(g["pairs"]), dev(g["num"]), V, False, True)
    assert torch.equal(o, o2)


def test_conv_wide_kernel_external_rulebook(oracle):
    """kv = 45 (a 3x3x5 kernel) from a rulebook built outside pcs_get_indice_pairs (the conv entry
    points take any kv; the matrix-core kernel stages only 32 offsets in LDS, so wider kernels must
    be routed to the tile kernels) -- against the oracle, MFMA-shaped channel counts included."""
    rng = np.random.default_rng(21)
    V, kv = 3000, 45
    pairs = np.full((kv, 2, V), -1, np.int32)
    num = np.zeros(kv, np.int32)
    for k in range(kv):
        n = int(rng.integers(0, V // 2))
        src = rng.choice(V, n, replace=False)
        dst = rng.choice(V, n, replace=False)      # each output at most once per offset
        pairs[k, 0, :n], pairs[k, 1, :n], num[k] = src, dst, n
    for cin, cout in ((32, 64), (16, 16), (5, 16)):
        feat = rng.normal(0, 1, (V, cin)).astype(np.float32)
        W = rng.normal(0, 0.05, (kv, cin, cout)).astype(np.float32)
        o = ops.indice_conv(dev(feat), dev(W), dev(pairs), dev(num), V, False, False)
        ref = oracle.indice_conv(feat, W, pairs, num, V, False, False)
        np.testing.assert_allclose(o.cpu().numpy(), ref, rtol=1e-5, atol=3e-5)


def test_conv_epilogue_and_inverse(oracle):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "rulebook_spconv_s2p1.npz"))
    rng = np.random.default_rng(9)
    V, Vo = g["indices"].shape[0], g["out_ids"].shape[0]
    cin, cout = 16, 32
    feat = rng.normal(0, 1, (V, cin)).astype(np.float32)
    W = rng.normal(0, 0.05, (3, 3, 3, cin, cout)).astype(np.float32)
    bias = rng.normal(0, 1, cout).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.normal(0, 1, cout).astype(np.float32)
    pairs, num = dev(g["pairs"]), dev(g["num"])
    table = ops.build_neighbor_table(pairs, num, Vo)
    o = ops.indice_conv_table(dev(feat), dev(W), table, Vo, False, dev(bias), dev(scale), dev(shift), True)
    ref = oracle.indice_conv(feat, W, g["pairs"], g["num"], Vo, False, False)
    ref = np.maximum((ref + bias) * scale + shift, 0)
    np.testing.assert_allclose(o.cpu().numpy(), ref, rtol=1e-5, atol=2e-5)
    # inverse conv: gather from outputs, scatter to inputs (spconv_ops.h `inverse`)
    fo = rng.normal(0, 1, (Vo, cin)).astype(np.float32)
    oi = ops.indice_conv(dev(fo), dev(W), pairs, num, V, True, False)
    refi = oracle.indice_conv(fo, W, g["pairs"], g["num"], V, True, False)
    np.testing.assert_allclose(oi.cpu().numpy(), refi, rtol=1e-5, atol=2e-5)


def test_empty_inputs():
    idx = torch.zeros(0, 4, dtype=torch.int32, device=DEV)
    out_ids, pairs, num = ops.get_indice_pairs(idx, 1, [5, 5, 5], [5, 5, 5], [3, 3, 3], [1, 1, 1],
                                               [1, 1, 1], [1, 1, 1], True)
    assert pairs.shape == (27, 2, 0) and int(num.sum()) == 0


@pytest.mark.parametrize("cin,cout", [(5, 16), (16, 16), (32, 64), (64, 128), (128, 128), (7, 9)])
def test_conv_backward_vs_oracle(oracle, cin, cout):
    """pcs_indice_conv_backward_f32 against the oracle (itself pinned to torch's
    dense conv3d autograd, tests/test_oracle_pinning.py) for subM, strided and inverse convs."""
    rng = np.random.default_rng(cin * 17 + cout)
    gs = np.load(os.path.join(os.path.dirname(__file__), "golden", "rulebook_subm3.npz"))
    gc = np.load(os.path.join(os.path.dirname(__file__), "golden", "rulebook_spconv_s2p1.npz"))
    cases = [(gs["pairs"], gs["num"], gs["indices"].shape[0], gs["indices"].shape[0], False, True),
             (gc["pairs"], gc["num"], gc["indices"].shape[0], gc["out_ids"].shape[0], False, False),
             (gc["pairs"], gc["num"], gc["out_ids"].shape[0], gc["indices"].shape[0], True, False)]
    for pairs, num, n_in, n_out, inverse, subm in cases:
        feat = rng.normal(0, 1, (n_in, cin)).astype(np.float32)
        W = rng.normal(0, 0.05, (3, 3, 3, cin, cout)).astype(np.float32)
        og = rng.normal(0, 1, (n_out, cout)).astype(np.float32)
        ib, wb = ops.indice_conv_backward(dev(feat), dev(W), dev(og), dev(pairs), dev(num), inverse, subm)
        ref_i, ref_w = oracle.indice_conv_backward(feat, W, og, pairs, num, inverse, subm)
        np.testing.assert_allclose(ib.cpu().numpy(), ref_i, rtol=1e-4, atol=1e-4)
        # filter gradient sums thousands of products per entry: tolerance relative to its scale
        tol = 2e-5 * max(1.0, float(np.abs(ref_w).max())) * 10
        np.testing.assert_allclose(wb.cpu().numpy(), ref_w, rtol=1e-4, atol=tol)
        ib2, wb2 = ops.indice_conv_backward(dev(feat), dev(W), dev(og), dev(pairs), dev(num), inverse, subm)
        assert torch.equal(ib, ib2) and torch.equal(wb, wb2)  # no atomics: reproducible


def test_conv_modules_autograd_vs_oracle(oracle):
    """loss.backward() through SubMConv3d -> SparseConv3d -> SparseInverseConv3d (functional.py):
    parameter and input gradients against the oracle chain."""
    from pcseqlearning_amd.pcdet.ops import spconv
    torch.manual_seed(0)
    rng = np.random.default_rng(2)
    shape = [8, 30, 30]
    lin = rng.choice(int(np.prod(shape)), 900, replace=False)
    idx = np.concatenate([np.zeros((900, 1), np.int64), np.stack(np.unravel_index(lin, shape), 1)], 1).astype(np.int32)
    feat = rng.normal(size=(900, 6)).astype(np.float32)
    c1 = spconv.SubMConv3d(6, 16, 3, padding=1, bias=True, indice_key="s1").to(DEV)
    c2 = spconv.SparseConv3d(16, 32, 3, stride=2, padding=1, bias=False, indice_key="d2").to(DEV)
    c3 = spconv.SparseInverseConv3d(32, 8, 3, indice_key="d2", bias=False).to(DEV)
    x = dev(feat).requires_grad_(True)
    t = spconv.SparseConvTensor(x, dev(idx), shape, 1)
    y = c3(c2(c1(t)))
    G = rng.normal(size=tuple(y.features.shape)).astype(np.float32)
    (y.features * dev(G)).sum().backward()
    # oracle chain
    k3, p1 = [3, 3, 3], [1, 1, 1]
    _, sp, sn = oracle.get_indice_pairs(idx, shape, k3, [1, 1, 1], p1, [1, 1, 1], True)
    oshape = oracle.conv_out_shape(shape, k3, [2, 2, 2], p1, [1, 1, 1])
    didx, dp, dn = oracle.get_indice_pairs(idx, oshape, k3, [2, 2, 2], p1, [1, 1, 1], False)
    W1, b1 = c1.weight.detach().cpu().numpy(), c1.bias.detach().cpu().numpy()
    W2, W3 = c2.weight.detach().cpu().numpy(), c3.weight.detach().cpu().numpy()
    h1 = oracle.indice_conv(feat, W1, sp, sn, 900, False, True) + b1
    h2 = oracle.indice_conv(h1, W2, dp, dn, didx.shape[0], False, False)
    h3 = oracle.indice_conv(h2, W3, dp, dn, 900, True, False)
    np.testing.assert_allclose(y.features.detach().cpu().numpy(), h3, rtol=1e-4, atol=1e-4)
    g2, gW3 = oracle.indice_conv_backward(h2, W3, G, dp, dn, True, False)
    g1, gW2 = oracle.indice_conv_backward(h1, W2, g2, dp, dn, False, False)
    g0, gW1 = oracle.indice_conv_backward(feat, W1, g1, sp, sn, False, True)
    for got, want in ((c3.weight.grad, gW3), (c2.weight.grad, gW2), (c1.weight.grad, gW1), (x.grad, g0),
                      (c1.bias.grad, g1.sum(0))):
        want = np.asarray(want)
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-4, atol=2e-4 * max(1.0, float(np.abs(want).max())))


@pytest.mark.parametrize("name", ["maxpool_ties", "maxpool_float"])
def test_maxpool_vs_reference_golden(oracle, name):
    """pcs_indice_maxpool_f32 / _backward_f32 against the outputs of the
    reference's own CPU functors (tests/golden/make_golden_maxpool.py). Bit-exact, ties included."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    n_out = g["out_ids"].shape[0]
    out = ops.indice_maxpool(dev(g["features"]), dev(g["pairs"]), dev(g["num"]), n_out)
    np.testing.assert_array_equal(out.cpu().numpy(), g["out"])
    din = ops.indice_maxpool_backward(dev(g["features"]), out, dev(g["out_grad"]), dev(g["pairs"]), dev(g["num"]))
    np.testing.assert_array_equal(din.cpu().numpy(), g["in_grad"])
    # module + autograd: SparseMaxPool3d(3, 2, 1) builds the same rulebook and routes the gradient
    from pcseqlearning_amd.pcdet.ops import spconv
    x = dev(g["features"]).requires_grad_(True)
    t = spconv.SparseConvTensor(x, dev(g["indices"]), [7, 20, 22], 1)
    y = spconv.SparseMaxPool3d(3, stride=2, padding=1)(t)
    np.testing.assert_array_equal(y.indices.cpu().numpy(), g["out_ids"])
    np.testing.assert_array_equal(y.features.detach().cpu().numpy(), g["out"])
    (y.features * dev(g["out_grad"])).sum().backward()
    np.testing.assert_array_equal(x.grad.cpu().numpy(), g["in_grad"])


@pytest.mark.gpu
def test_bn_fold_kernel_matches_the_tensor_ops():
    """pcs_bn_fold_f32 (one launch) against the four tensor ops it replaces: bit for bit."""
    from pcseqlearning_amd import ops
    torch.manual_seed(5)
    for C, affine in ((16, True), (64, True), (128, False), (5, True)):
        bn = torch.nn.BatchNorm1d(C, eps=1e-3, affine=affine).cuda().eval()
        bn.running_mean.normal_()
        bn.running_var.uniform_(0.1, 4.0)
        if affine:
            bn.weight.data.normal_()
            bn.bias.data.normal_()
        inv = torch.rsqrt(bn.running_var.float() + bn.eps)
        scale = inv * bn.weight.float() if affine else inv
        shift = -bn.running_mean.float() * scale
        if affine:
            shift = shift + bn.bias.float()
        s2, h2 = ops.bn_fold(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
        assert torch.equal(s2, scale) and torch.equal(h2, shift)


def test_deconv_rulebook_and_transposed_conv(oracle):
    """Transposed conv (getIndicePairsDeConv, geometry.h:197-245; SparseConvTranspose3d, conv.py:308-331):
    the HIP rulebook bit-exact against the reference's own outputs (tests/golden/deconvbook_*.npz) and
    against the oracle on larger random inputs; the module's features against the oracle's
    gather-GEMM-scatter over the same rulebook."""
    from pcseqlearning_amd.pcdet.ops import spconv
    from tests.golden.make_golden import random_indices
    from tests.golden.make_golden_deconv import DECONV_CASES
    for name, shape, ks, st, pd, dl, opad in DECONV_CASES:
        g = np.load(os.path.join(os.path.dirname(__file__), "golden", f"deconvbook_{name}.npz"))
        out_ids, pairs, num = ops.get_indice_pairs(dev(g["indices"]), 2, list(g["out_shape"]), list(g["spatial_shape"]),
                                                   list(ks), list(st), list(pd), list(dl), False, transpose=True)
        np.testing.assert_array_equal(num.cpu().numpy(), g["num"])
        np.testing.assert_array_equal(out_ids.cpu().numpy(), g["out_ids"])
        np.testing.assert_array_equal(pairs.cpu().numpy(), g["pairs"])
    rng = np.random.default_rng(31)
    shape, ks, st, pd, dl, opad = [21, 80, 72], [3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1], [1, 1, 1]
    idx = random_indices(rng, shape, 20000, batch=2)
    out_shape = oracle.deconv_out_shape(shape, ks, st, pd, dl, opad)
    want = oracle.get_indice_pairs(idx, out_shape, ks, st, pd, dl, False, transpose=True)
    out_ids, pairs, num = ops.get_indice_pairs(dev(idx), 2, out_shape, shape, ks, st, pd, dl, False, transpose=True)
    np.testing.assert_array_equal(num.cpu().numpy(), want[2])
    np.testing.assert_array_equal(out_ids.cpu().numpy(), want[0])
    np.testing.assert_array_equal(pairs.cpu().numpy(), want[1])
    # the module: 16 -> 32 channels
    feat = rng.normal(0, 1, (idx.shape[0], 16)).astype(np.float32)
    conv = spconv.SparseConvTranspose3d(16, 32, 3, stride=2, padding=1, bias=True, indice_key="up").to(DEV)
    conv.output_padding = [1, 1, 1]
    x = spconv.SparseConvTensor(dev(feat), dev(idx), shape, 2)
    with torch.no_grad():
        y = conv(x)
    assert list(y.spatial_shape) == out_shape
    np.testing.assert_array_equal(y.indices.cpu().numpy(), want[0])
    W = conv.weight.detach().cpu().numpy()
    ref = oracle.indice_conv(feat, W, want[1], want[2], want[0].shape[0], False, False) + conv.bias.detach().cpu().numpy()
    np.testing.assert_allclose(y.features.cpu().numpy(), ref, rtol=1e-4, atol=1e-4)


def test_half_precision_conv_and_maxpool(oracle):
    """indice_conv_half / fused_indice_conv_half / indice_maxpool_half (src/all.cc:35-49). The HIP entries keep
    fp16 storage and accumulate in fp32 over all offsets; the oracle restates the reference's order of
    roundings (each offset's GEMM rounded to fp16, fp16 scatter-add): agreement to a few fp16 units of the
    result's scale; against the fp32 conv of the same fp16 values to one fp16 rounding; pooling bit-exact."""
    from tests.golden.make_golden import random_indices
    rng = np.random.default_rng(41)
    shape = [9, 24, 24]
    idx = random_indices(rng, shape, 1500, batch=2)
    V = idx.shape[0]
    for subm, ks, st, pd in ((True, [3, 3, 3], [1, 1, 1], [1, 1, 1]), (False, [3, 3, 3], [2, 2, 2], [1, 1, 1])):
        out_shape = shape if subm else oracle.conv_out_shape(shape, ks, st, pd, [1, 1, 1])
        out_ids, pairs, num = oracle.get_indice_pairs(idx, out_shape, ks, st, pd, [1, 1, 1], subm)
        n_out = V if subm else out_ids.shape[0]
        feat = rng.normal(0, 1, (V, 16)).astype(np.float16)
        W = rng.normal(0, 0.1, (27, 16, 32)).astype(np.float16)
        bias = rng.normal(0, 0.5, 32).astype(np.float16)
        for b in (None, bias):
            got = ops.indice_conv_half(dev(feat), dev(W), dev(pairs), dev(num), n_out, False, subm,
                                       bias=None if b is None else dev(b))
            assert got.dtype == torch.float16
            got = got.cpu().numpy().astype(np.float32)
            want = oracle.indice_conv_half(feat, W, pairs, num, n_out, False, subm, bias=b).astype(np.float32)
            scale = float(np.abs(want).max())
            assert float(np.abs(got - want).max()) <= 8 * 2.0 ** -10 * scale
            exact = oracle.indice_conv(feat.astype(np.float32), W.astype(np.float32), pairs, num, n_out, False, subm)
            if b is not None:
                exact = exact + b.astype(np.float32)
            np.testing.assert_allclose(got, exact, rtol=2.0 ** -10, atol=2.0 ** -11 * scale)
        if not subm:
            pooled = ops.indice_maxpool_half(dev(feat), dev(pairs), dev(num), n_out)
            want = oracle.indice_maxpool(feat.astype(np.float32), pairs, num, n_out)
            assert pooled.dtype == torch.float16
            np.testing.assert_array_equal(pooled.cpu().numpy().astype(np.float32), want)


@pytest.mark.parametrize("cin,cout", [(16, 16), (32, 64), (64, 64), (128, 128), (64, 32), (48, 32), (20, 24)])
def test_half_conv_matrix_core_tiles_vs_fp32_of_the_same_values(oracle, cin, cout):
    """pcs_indice_conv_f16 on the fp16 matrix pipe (v_mfma_f32_16x16x32_f16: channel counts that are multiples of
    16, zero-padded to the K of an MFMA; wider than 64 channels in slices) and on the one-thread-per-element form
    (48 -> slices of 64 + 16-multiple; 20 / 24: not tile shapes): products of fp16 values are exact in fp32, so the
    result is the fp32 conv of the same values rounded once -- within one fp16 rounding of the oracle's fp32 sum."""
    from tests.golden.make_golden import random_indices
    rng = np.random.default_rng(100 + cin + cout)
    shape = [9, 20, 20]
    idx = random_indices(rng, shape, 1200, batch=2)
    V = idx.shape[0]
    for subm, ks, st, pd in ((True, [3, 3, 3], [1, 1, 1], [1, 1, 1]), (False, [3, 3, 3], [2, 2, 2], [1, 1, 1])):
        out_shape = shape if subm else oracle.conv_out_shape(shape, ks, st, pd, [1, 1, 1])
        out_ids, pairs, num = oracle.get_indice_pairs(idx, out_shape, ks, st, pd, [1, 1, 1], subm)
        n_out = V if subm else out_ids.shape[0]
        feat = rng.normal(0, 1, (V, cin)).astype(np.float16)
        W = rng.normal(0, 0.1, (27, cin, cout)).astype(np.float16)
        got = ops.indice_conv_half(dev(feat), dev(W), dev(pairs), dev(num), n_out, False, subm).cpu().numpy()
        exact = oracle.indice_conv(feat.astype(np.float32), W.astype(np.float32), pairs, num, n_out, False, subm)
        scale = float(np.abs(exact).max())
        np.testing.assert_allclose(got.astype(np.float32), exact, rtol=2.0 ** -10, atol=2.0 ** -10 * scale)


@pytest.mark.parametrize("cin,cout", [(16, 16), (32, 32), (64, 64), (128, 128), (64, 32), (48, 32), (5, 16), (8, 32)])
def test_half_conv_forms_agree_bit_for_bit(oracle, cin, cout):
    """Round 5's fp16 conv kernels -- rows gathered straight into the MFMA fragments, weights of several offsets in
    LDS (512- and 256-thread workgroups); 128-row staged tiles with the next offset prefetched -- against round 4's
    64-row kernel (pcs_indice_conv_f16_short_tiles): per output row the offsets, channel slices and K chunks come in
    the same order, so the fp16 results are identical -- on a ragged row count, with and without bias, subM (centre
    first) and strided. Up to 8 input channels (the backbone's first conv): one thread per output row against one per
    output element."""
    from pcseqlearning_amd import lib as L
    from tests.golden.make_golden import random_indices
    rng = np.random.default_rng(500 + cin + cout)
    shape = [9, 28, 28]
    idx = random_indices(rng, shape, 3001, batch=2)
    V = idx.shape[0]
    lib = L.load()
    for subm, ks, st, pd in ((True, [3, 3, 3], [1, 1, 1], [1, 1, 1]), (False, [3, 3, 3], [2, 2, 2], [1, 1, 1])):
        out_shape = shape if subm else oracle.conv_out_shape(shape, ks, st, pd, [1, 1, 1])
        out_ids, pairs, num = oracle.get_indice_pairs(idx, out_shape, ks, st, pd, [1, 1, 1], subm)
        n_out = V if subm else out_ids.shape[0]
        feat = rng.normal(0, 1, (V, cin)).astype(np.float16)
        W = rng.normal(0, 0.1, (27, cin, cout)).astype(np.float16)
        bias = rng.normal(0, 0.5, cout).astype(np.float16)
        for b in (None, bias):
            res = []
            for short in (1, 0, 2, 3):
                L.check(lib.pcs_indice_conv_f16_short_tiles(short), "tiles")
                try:
                    res.append(ops.indice_conv_half(dev(feat), dev(W), dev(pairs), dev(num), n_out, False, subm,
                                                    bias=None if b is None else dev(b)).cpu().numpy())
                finally:
                    L.check(lib.pcs_indice_conv_f16_short_tiles(0), "tiles")
            for other in res[1:]:
                np.testing.assert_array_equal(res[0].view(np.uint16), other.view(np.uint16))


@pytest.mark.parametrize("cin,cout", [(16, 16), (32, 64), (64, 64), (64, 128)])
def test_conv_rows_in_offset_set_order_give_the_same_bits(oracle, cin, cout):
    """pcs_neighbor_table_order + pcs_indice_conv_table_ordered_f32 / _f16: the output rows taken in the order of their
    sets of occupied offsets (stable: ties in row order) -- every row is the same sum in the same order, so the
    outputs equal the table-order call bit for bit; the order is a permutation sorted by the rows' bit sets. The f32
    direct-gather form against the oracle as well (its order of products inside a 16-channel chunk is its own)."""
    from tests.golden.make_golden import random_indices
    rng = np.random.default_rng(900 + cin + cout)
    shape = [9, 28, 28]
    idx = random_indices(rng, shape, 3001, batch=2)
    V = idx.shape[0]
    for subm, ks, st, pd in ((True, [3, 3, 3], [1, 1, 1], [1, 1, 1]), (False, [3, 3, 3], [2, 2, 2], [1, 1, 1])):
        out_shape = shape if subm else oracle.conv_out_shape(shape, ks, st, pd, [1, 1, 1])
        out_ids, pairs, num = oracle.get_indice_pairs(idx, out_shape, ks, st, pd, [1, 1, 1], subm)
        n_out = V if subm else out_ids.shape[0]
        table = ops.build_neighbor_table(dev(pairs), dev(num), n_out)
        order = ops.neighbor_table_order(table, n_out)
        t = table.cpu().numpy()
        o = order.cpu().numpy()
        assert sorted(o.tolist()) == list(range(n_out))
        key = ((t >= 0).astype(np.uint64) << np.arange(27, dtype=np.uint64)).sum(1)
        np.testing.assert_array_equal(o, np.lexsort((key, np.arange(n_out) // 4096)))
        feat = rng.normal(0, 1, (V, cin)).astype(np.float32)
        W = rng.normal(0, 0.1, (27, cin, cout)).astype(np.float32)
        bias = rng.normal(0, 0.5, cout).astype(np.float32)
        from pcseqlearning_amd import lib as L
        L.check(L.load().pcs_conv_use_mfma(3), "mode")  # (the direct-gather form also at this size)
        try:
            a = ops.indice_conv_table(dev(feat), dev(W), table, n_out, subm, bias=dev(bias), relu=True).cpu().numpy()
            b = ops.indice_conv_table(dev(feat), dev(W), table, n_out, subm, bias=dev(bias), relu=True, order=order).cpu().numpy()
        finally:
            L.check(L.load().pcs_conv_use_mfma(1), "mode")
        np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
        ref = np.maximum(oracle.indice_conv(feat, W, pairs, num, n_out, False, subm) + bias, 0)
        np.testing.assert_allclose(a, ref, rtol=1e-4, atol=1e-4)
        h = ops.indice_conv_table_half(dev(feat).half(), dev(W).half(), table, n_out, subm, bias=dev(bias), relu=True)
        ho = ops.indice_conv_table_half(dev(feat).half(), dev(W).half(), table, n_out, subm, bias=dev(bias), relu=True,
                                        order=order)
        np.testing.assert_array_equal(h.cpu().numpy().view(np.uint16), ho.cpu().numpy().view(np.uint16))


@pytest.mark.parametrize("cin,cout", [(5, 16), (3, 32), (8, 16)])
def test_first_conv_one_thread_per_row_is_the_tile_kernel_bit_for_bit(oracle, cin, cout):
    """Up to 8 input channels (VoxelBackBone8x's 5 -> 16): round 5's one-thread-per-output-row kernel against round 4's
    tile kernel (pcs_conv_use_mfma(2) selects the older forms): offsets centre-first, channels ascending, one fmaf
    each in both, so the f32 outputs are identical; and against the oracle."""
    from pcseqlearning_amd import lib as L
    from tests.golden.make_golden import random_indices
    rng = np.random.default_rng(300 + cin + cout)
    shape = [9, 28, 28]
    idx = random_indices(rng, shape, 3001, batch=2)
    V = idx.shape[0]
    lib = L.load()
    for subm, ks, st, pd in ((True, [3, 3, 3], [1, 1, 1], [1, 1, 1]), (False, [3, 3, 3], [2, 2, 2], [1, 1, 1])):
        out_shape = shape if subm else oracle.conv_out_shape(shape, ks, st, pd, [1, 1, 1])
        out_ids, pairs, num = oracle.get_indice_pairs(idx, out_shape, ks, st, pd, [1, 1, 1], subm)
        n_out = V if subm else out_ids.shape[0]
        table = ops.build_neighbor_table(dev(pairs), dev(num), n_out)
        feat = rng.normal(0, 1, (V, cin)).astype(np.float32)
        W = rng.normal(0, 0.1, (27, cin, cout)).astype(np.float32)
        bias = rng.normal(0, 0.5, cout).astype(np.float32)
        res = []
        for mode in (2, 1):
            L.check(lib.pcs_conv_use_mfma(mode), "mode")
            try:
                res.append(ops.indice_conv_table(dev(feat), dev(W), table, n_out, subm, bias=dev(bias), relu=True).cpu().numpy())
            finally:
                L.check(lib.pcs_conv_use_mfma(1), "mode")
        np.testing.assert_array_equal(res[0].view(np.uint32), res[1].view(np.uint32))
        ref = np.maximum(oracle.indice_conv(feat, W, pairs, num, n_out, False, subm) + bias, 0)
        np.testing.assert_allclose(res[1], ref, rtol=1e-4, atol=1e-4)


def test_neighbor_table_order_beyond_one_grid_of_workgroups():
    """600 001 rows (the mask kernel's launch is capped at 2048 workgroups and strides over the rest): the order is
    numpy's stable argsort of the rows' bit sets."""
    rng = np.random.default_rng(77)
    n = 600001
    t = np.where(rng.random((n, 27)) < 0.5, rng.integers(0, n, (n, 27)), -1).astype(np.int32)
    order = ops.neighbor_table_order(dev(t), n).cpu().numpy()
    key = ((t >= 0).astype(np.uint64) << np.arange(27, dtype=np.uint64)).sum(1)
    np.testing.assert_array_equal(order, np.lexsort((key, np.arange(n) // 4096)))


def test_half_precision_backward_entries(oracle):
    """indice_conv_backward_half / indice_maxpool_backward_half (src/all.cc:36-37, 48-50): input and filter gradients
    from fp16 rows with fp32 accumulation, rounded once -- within fp16 rounding of the oracle's fp32 gradients of
    the same values; the pooling's backward adds the selected gradients in fp16 in ascending offset (the
    reference's functor on at::Half), restated here in numpy float16."""
    from tests.golden.make_golden import random_indices
    rng = np.random.default_rng(77)
    shape = [9, 20, 20]
    idx = random_indices(rng, shape, 1000, batch=2)
    V = idx.shape[0]
    for subm, ks, st, pd in ((True, [3, 3, 3], [1, 1, 1], [1, 1, 1]), (False, [3, 3, 3], [2, 2, 2], [1, 1, 1])):
        out_shape = shape if subm else oracle.conv_out_shape(shape, ks, st, pd, [1, 1, 1])
        out_ids, pairs, num = oracle.get_indice_pairs(idx, out_shape, ks, st, pd, [1, 1, 1], subm)
        n_out = V if subm else out_ids.shape[0]
        for cin, cout in ((16, 32), (64, 64), (20, 24)):
            feat = rng.normal(0, 1, (V, cin)).astype(np.float16)
            W = rng.normal(0, 0.1, (27, cin, cout)).astype(np.float16)
            og = rng.normal(0, 1, (n_out, cout)).astype(np.float16)
            g_in, g_w = ops.indice_conv_backward_half(dev(feat), dev(W), dev(og), dev(pairs), dev(num), False, subm)
            assert g_in.dtype == torch.float16 and g_w.dtype == torch.float16
            w_in, w_w = oracle.indice_conv_backward(feat.astype(np.float32), W.astype(np.float32), og.astype(np.float32),
                                                    pairs, num, False, subm)
            for got, want in ((g_in, w_in), (g_w, w_w)):
                want = np.asarray(want, np.float32).reshape(got.shape)
                scale = float(np.abs(want).max())
                np.testing.assert_allclose(got.cpu().numpy().astype(np.float32), want, rtol=2.0 ** -9,
                                           atol=2.0 ** -9 * scale)
        if not subm:
            feat = rng.integers(-3, 4, (V, 16)).astype(np.float16)  # (ties: several inputs equal the pooled value)
            pooled = oracle.indice_maxpool(feat.astype(np.float32), pairs, num, n_out).astype(np.float16)
            og = rng.normal(0, 1, (n_out, 16)).astype(np.float16)
            got = ops.indice_maxpool_backward_half(dev(feat), dev(pooled), dev(og), dev(pairs), dev(num))
            want = np.zeros((V, 16), np.float16)
            for k in range(pairs.shape[0]):  # the reference's loop (maxpool.cc:46-62) on float16
                for p in range(int(num[k])):
                    i, o = int(pairs[k, 0, p]), int(pairs[k, 1, p])
                    hit = feat[i] == pooled[o]
                    want[i] = np.where(hit, (want[i] + og[o]).astype(np.float16), want[i])
            np.testing.assert_array_equal(got.cpu().numpy(), want)


def _rulebook_2d(idx, shape, ks, st, pd, subm):
    """get_indice_pairs_2d restated in plain Python (geometry.h:25-195 for NDim = 2): inputs in order, kernel offsets
    row-major, an output id is its rank of first appearance; subM: outputs = inputs, an offset pairs a voxel with the
    input found at (p + offset - centre)."""
    kv = ks[0] * ks[1]
    pairs = [[] for _ in range(kv)]
    if subm:
        where = {}
        for j, (b, y, x) in enumerate(idx.tolist()):
            where[(b, y, x)] = j
        for j, (b, y, x) in enumerate(idx.tolist()):
            for ky in range(ks[0]):
                for kx in range(ks[1]):
                    # the output that input j reaches through offset (ky, kx): out * 1 - pad + k = in
                    o = (b, y + pd[0] - ky, x + pd[1] - kx)
                    if o in where:
                        pairs[ky * ks[1] + kx].append((j, where[o]))
        return idx, pairs
    out_shape = [(shape[d] + 2 * pd[d] - (ks[d] - 1) - 1) // st[d] + 1 for d in range(2)]
    ids, order = {}, []
    for j, (b, y, x) in enumerate(idx.tolist()):
        for ky in range(ks[0]):
            for kx in range(ks[1]):
                ny, nx = y + pd[0] - ky, x + pd[1] - kx
                if ny % st[0] or nx % st[1]:
                    continue
                oy, ox = ny // st[0], nx // st[1]
                if not (0 <= oy < out_shape[0] and 0 <= ox < out_shape[1]):
                    continue
                key = (b, oy, ox)
                if key not in ids:
                    ids[key] = len(order)
                    order.append(key)
                pairs[ky * ks[1] + kx].append((j, ids[key]))
    return np.array(order, np.int32).reshape(-1, 3), pairs


@pytest.mark.parametrize("subm", [True, False])
def test_rulebook_and_modules_in_two_dimensions(subm):
    """get_indice_pairs_2d / SubMConv2d / SparseConv2d / SparseMaxPool2d (src/all.cc:22-25, conv.py:207-229, 370-393,
    pool.py:74-79 of the reference): a plane goes through the 3-D rulebook as a volume one cell deep. The pairs of
    every offset, as SETS per offset, and the conv against a plain-Python 2-D restatement; the output ids of the
    strided form are the restatement's up to the order the reference leaves to its hash walk (compared as sets, the
    features row by row through the ids)."""
    from pcseqlearning_amd.pcdet.ops import spconv
    rng = np.random.default_rng(12 + int(subm))
    shape = [40, 36]
    cells = rng.permutation(2 * 40 * 36)[:900]
    idx = np.stack([cells // (40 * 36), (cells // 36) % 40, cells % 36], 1).astype(np.int32)
    ks, st, pd = ([3, 3], [1, 1], [1, 1]) if subm else ([3, 3], [2, 2], [1, 1])
    outids, pairs, num = spconv.ops.get_indice_pairs(dev(idx), 2, shape, ks, st, pd, 1, 0, subm)[:3]
    want_ids, want_pairs = _rulebook_2d(idx, shape, ks, st, pd, subm)
    got_ids = outids.cpu().numpy()
    assert got_ids.shape[1] == 3
    assert sorted(map(tuple, got_ids.tolist())) == sorted(map(tuple, want_ids.tolist()))
    p, n = pairs.cpu().numpy(), num.cpu().numpy()
    to_key = {i: tuple(r) for i, r in enumerate(got_ids.tolist())}
    want_key = {i: tuple(r) for i, r in enumerate(want_ids.tolist())}
    for k in range(9):
        got = {(int(a), to_key[int(b)]) for a, b in zip(p[k, 0, :n[k]], p[k, 1, :n[k]])}
        want = {(a, want_key[b]) for a, b in want_pairs[k]}
        assert got == want, k
    # the module: conv + max-pool on the plane against dense torch ops
    feat = rng.normal(0, 1, (idx.shape[0], 4)).astype(np.float32)
    x = spconv.SparseConvTensor(dev(feat), dev(idx), shape, 2)
    conv = (spconv.SubMConv2d(4, 8, 3, padding=1, bias=True) if subm else
            spconv.SparseConv2d(4, 8, 3, stride=2, padding=1, bias=True)).to(DEV)
    with torch.no_grad():
        y = conv(x)
        dense_in = x.dense()                                   # [B, C, H, W]
        w = conv.weight.permute(3, 2, 0, 1).contiguous()       # [Cout, Cin, ky, kx]
        ref = torch.nn.functional.conv2d(dense_in, w, conv.bias, stride=st, padding=pd)
        yi = y.indices.long()
        np.testing.assert_allclose(y.features.cpu().numpy(), ref[yi[:, 0], :, yi[:, 1], yi[:, 2]].cpu().numpy(),
                                   rtol=1e-4, atol=1e-4)
        if not subm:
            pool = spconv.SparseMaxPool2d(3, stride=2, padding=1)
            z = pool(x)
            dz = torch.nn.functional.max_pool2d(dense_in.clamp(min=0), 3, stride=2, padding=1)
            zi = z.indices.long()
            np.testing.assert_allclose(z.features.cpu().numpy(), dz[zi[:, 0], :, zi[:, 1], zi[:, 2]].cpu().numpy(),
                                       rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("kv,n_out,cin,cout", [(1, 1, 16, 16), (3, 127, 32, 32), (8, 129, 64, 64), (9, 1000, 16, 32),
                                               (27, 4097, 64, 128), (32, 300, 32, 64), (27, 100001, 16, 16)])
def test_conv_kernel_forms_on_random_tables(kv, n_out, cin, cout):
    """The conv entries on tables that no rulebook would make -- random partners, rows without any, kernel volumes 1 to
    32, row counts around the tiles' edges (1, 127, 129, 4097, 100 001), with and without a centre offset: round 5's
    direct-gather kernels against round 4's staged ones (fp16: identical bits; f32: the same sums in another order
    inside a 16-channel chunk) and against a float64 restatement."""
    from pcseqlearning_amd import lib as L
    lib = L.load()
    rng = np.random.default_rng(kv * 1000 + n_out % 997 + cin + cout)
    n_in = max(n_out // 2, 3)
    table = rng.integers(0, n_in, (n_out, kv)).astype(np.int32)
    table[rng.random((n_out, kv)) < 0.5] = -1
    table[rng.random(n_out) < 0.1] = -1
    feat = rng.normal(0, 1, (n_in, cin)).astype(np.float32)
    W = rng.normal(0, 0.1, (kv, cin, cout)).astype(np.float32)
    bias = rng.normal(0, 0.5, cout).astype(np.float32)
    for subm in ((True, False) if kv % 2 else (False,)):
        got32 = {}
        for mode in (2, 3):
            L.check(lib.pcs_conv_use_mfma(mode), "mode")
            try:
                got32[mode] = ops.indice_conv_table(dev(feat), dev(W), dev(table), n_out, subm, bias=dev(bias),
                                                    relu=True).cpu().numpy()
            finally:
                L.check(lib.pcs_conv_use_mfma(1), "mode")
        f64 = np.zeros((n_out, cout))
        for k in range(kv):
            sel = table[:, k] >= 0
            f64[sel] += feat[table[sel, k]].astype(np.float64) @ W[k].astype(np.float64)
        ref = np.maximum(f64 + bias, 0)
        np.testing.assert_allclose(got32[2], ref, rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(got32[3], ref, rtol=2e-4, atol=2e-4)
        got16 = []
        for tiles in (1, 0, 2, 3):
            L.check(lib.pcs_indice_conv_f16_short_tiles(tiles), "tiles")
            try:
                got16.append(ops.indice_conv_table_half(dev(feat).half(), dev(W).half(), dev(table), n_out, subm,
                                                        bias=dev(bias), relu=True).cpu().numpy())
            finally:
                L.check(lib.pcs_indice_conv_f16_short_tiles(0), "tiles")
        for other in got16[1:]:
            np.testing.assert_array_equal(got16[0].view(np.uint16), other.view(np.uint16))
        scale = max(float(np.abs(ref).max()), 1.0)
        assert float(np.abs(got16[0].astype(np.float64) - ref).max()) <= 2.0 ** -7 * scale


@pytest.mark.parametrize("kv,n_out,cin,cout", [(27, 4097, 16, 16), (27, 300, 64, 64), (27, 100001, 32, 32), (27, 2000, 128, 128),
                                               (27, 5000, 5, 16), (9, 700, 24, 40)])
def test_conv_residual_epilogue_is_the_separate_add(kv, n_out, cin, cout):
    """pcs_indice_conv_table_residual_f32: relu(((conv + bias) * scale + shift) + residual) in the conv's epilogue is,
    bit for bit, the conv without ReLU followed by the tensor add and ReLU -- on every fp32 kernel form (FMA tiles, staged
    and direct-gather matrix-core kernels, one thread per row, one thread per element), and the same for
    pcs_indice_conv_table_residual_f16 on the fp16 forms."""
    from pcseqlearning_amd import lib as L
    lib = L.load()
    rng = np.random.default_rng(n_out + cin)
    n_in = n_out
    table = rng.integers(0, n_in, (n_out, kv)).astype(np.int32)
    table[rng.random((n_out, kv)) < 0.6] = -1
    table[:, kv // 2] = np.arange(n_out)
    feat, W = rng.normal(0, 1, (n_in, cin)).astype(np.float32), rng.normal(0, 0.1, (kv, cin, cout)).astype(np.float32)
    bias, scale, shift = (rng.normal(0, 0.5, cout).astype(np.float32) for _ in range(3))
    res = rng.normal(0, 1, (n_out, cout)).astype(np.float32)
    for mode in (0, 1, 2, 3):
        L.check(lib.pcs_conv_use_mfma(mode), "mode")
        try:
            plain = ops.indice_conv_table(dev(feat), dev(W), dev(table), n_out, True, bias=dev(bias), scale=dev(scale),
                                          shift=dev(shift), relu=False)
            fused = ops.indice_conv_table(dev(feat), dev(W), dev(table), n_out, True, bias=dev(bias), scale=dev(scale),
                                          shift=dev(shift), relu=True, residual=dev(res))
        finally:
            L.check(lib.pcs_conv_use_mfma(1), "mode")
        assert torch.equal(fused, torch.relu(plain + dev(res))), mode
    # fp16: fp16(conv + bias) + fp16 residual as the tensor library adds two fp16 rows, then the ReLU -- on every form
    # (direct 256 / 512 threads, 128-row staged, round 4's 64-row, one thread per row, one thread per element)
    f16, W16, r16 = dev(feat).half(), dev(W).half(), dev(res).half()
    for tiles in (0, 1, 2, 3):
        L.check(lib.pcs_indice_conv_f16_short_tiles(tiles), "tiles")
        try:
            plain = ops.indice_conv_table_half(f16, W16, dev(table), n_out, True, bias=dev(bias), relu=False)
            fused = ops.indice_conv_table_half(f16, W16, dev(table), n_out, True, bias=dev(bias), relu=True, residual=r16)
        finally:
            L.check(lib.pcs_indice_conv_f16_short_tiles(0), "tiles")
        assert fused.dtype == torch.float16 and torch.equal(fused, torch.relu(plain + r16)), tiles


def test_residual_block_fused_add_vs_tensor_ops():
    """SparseBasicBlock in eval mode: the `+ identity` and the ReLU in the second conv's epilogue (default) against the
    same block with FUSE_RESIDUAL_ADD off -- the same bits, fp32 and fp16."""
    from pcseqlearning_amd.pcdet.models.blocks import spconv_blocks as SB
    from pcseqlearning_amd.pcdet.ops import spconv
    from functools import partial
    rng = np.random.default_rng(77)
    shape = [21, 60, 60]
    from tests.golden.make_golden import random_indices
    idx = random_indices(rng, shape, 9000, batch=2)
    for planes in (16, 64):
        torch.manual_seed(planes)
        block = SB.SparseBasicBlock(planes, planes, norm_fn=partial(torch.nn.BatchNorm1d, eps=1e-3, momentum=0.01),
                                    indice_key="res").to(DEV).eval()
        for bn in (block.bn1, block.bn2):
            bn.running_mean.normal_(0, 0.5)
            bn.running_var.uniform_(0.5, 2.0)
            bn.weight.data.uniform_(0.5, 1.5)
            bn.bias.data.normal_(0, 0.3)
        feat = torch.from_numpy(rng.normal(0, 1, (len(idx), planes)).astype(np.float32)).to(DEV)
        for half in (False, True):
            f = feat.half() if half else feat
            with torch.no_grad():
                outs = []
                for fuse in (True, False):
                    SB.FUSE_RESIDUAL_ADD = fuse
                    try:
                        x = spconv.SparseConvTensor(f, dev(idx), shape, 2)
                        outs.append(block(x).features)
                    finally:
                        SB.FUSE_RESIDUAL_ADD = True
            assert outs[0].dtype == outs[1].dtype and torch.equal(outs[0], outs[1]), (planes, half)
            assert float(outs[0].float().abs().mean()) > 0.05


def test_four_dimensional_conv_modules(oracle):
    """SubMConv4d / SparseConv4d (conv.py:257-279, 422-445 of the reference; get_indice_pairs_4d): the rulebook of
    ops._indice_pairs_nd on the device equals the loop restatement of geometry.h (tests/test_rulebook_nd.py pins that
    one), and the modules' outputs are the sum over its 81 offsets' pairs -- fp32, the conv kernels' any-volume form."""
    from pcseqlearning_amd.pcdet.ops import spconv
    from tests.test_rulebook_nd import random_indices_nd
    rng = np.random.default_rng(91)
    shape = [5, 8, 9, 7]
    idx = random_indices_nd(rng, shape, 1500, 2)
    feat = rng.normal(0, 1, (len(idx), 6)).astype(np.float32)
    torch.manual_seed(3)
    for mod, subm, st, pd in ((spconv.SubMConv4d(6, 16, 3, padding=1, indice_key="s4"), True, [1] * 4, [1] * 4),
                              (spconv.SparseConv4d(6, 16, 3, stride=2, padding=1), False, [2] * 4, [1] * 4)):
        mod = mod.to(DEV).eval()
        x = spconv.SparseConvTensor(dev(feat), dev(idx), shape, 2)
        with torch.no_grad():
            y = mod(x)
        osh = shape if subm else spconv.ops.get_conv_output_size(shape, [3] * 4, st, pd, [1] * 4)
        outids, pairs, num = oracle.get_indice_pairs_nd_py(idx, osh, [3] * 4, st, pd, [1] * 4, subm)
        assert list(y.spatial_shape) == list(osh)
        np.testing.assert_array_equal(y.indices.cpu().numpy(), outids)
        W = mod.weight.detach().cpu().numpy().reshape(81, 6, 16).astype(np.float64)
        ref = np.zeros((len(outids), 16))
        for k in range(81):
            n = int(num[k])
            np.add.at(ref, pairs[k, 1, :n], feat[pairs[k, 0, :n]].astype(np.float64) @ W[k])
        ref = ref + mod.bias.detach().cpu().numpy()
        np.testing.assert_allclose(y.features.cpu().numpy(), ref, rtol=1e-4, atol=1e-5)
    # the same rulebook twice under one indice_key (the cached pairs and table are used the second time)
    mod = spconv.SubMConv4d(16, 16, 3, padding=1, indice_key="s4").to(DEV).eval()
    first = spconv.SubMConv4d(6, 16, 3, padding=1, indice_key="s4").to(DEV).eval()
    with torch.no_grad():
        h = first(spconv.SparseConvTensor(dev(feat), dev(idx), shape, 2))
        assert "s4" in h.indice_dict
        z = mod(h)
    assert z.features.shape == (len(idx), 16) and torch.isfinite(z.features).all()
