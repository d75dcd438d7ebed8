"""VoxelResBackBone8x (residual blocks on the HIP sparse conv, eval-mode
BatchNorm / ReLU folded into the conv epilogues) against a layer-by-layer oracle chain, and
HeightCompression / SparseConvTensor.dense() (pcs_sparse_to_dense) against the reference's
scatter_nd + permute."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from pcseqlearning_amd import ops, synthetic  # noqa: E402
from pcseqlearning_amd.pcdet.models.backbones_2d import HeightCompression  # noqa: E402
from pcseqlearning_amd.pcdet.models.backbones_3d import VoxelBackBone8x, VoxelResBackBone8x  # noqa: E402
from pcseqlearning_amd.pcdet.models.backbones_3d.vfe import DynamicMeanVFE  # noqa: E402
from pcseqlearning_amd.pcdet.ops.spconv.structure import SparseConvTensor  # noqa: E402

DEV = "cuda:0"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def bn_np(x, bn):
    w, b = bn.weight.detach().cpu().numpy(), bn.bias.detach().cpu().numpy()
    m, v = bn.running_mean.cpu().numpy(), bn.running_var.cpu().numpy()
    return ((x - m) / np.sqrt(v + bn.eps) * w + b).astype(np.float32)


def randomise_bn(net):
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.1)


def test_voxel_res_backbone_8x_vs_oracle(oracle):
    torch.manual_seed(1)
    pts = synthetic.gen_voxel_frame(3, 30000)
    vfe = DynamicMeanVFE({}, 5, synthetic.VOXEL_SIZE, synthetic.GRID_SIZE, synthetic.PC_RANGE)
    net = VoxelResBackBone8x({}, 5, synthetic.GRID_SIZE.tolist()).to(DEV).eval()
    randomise_bn(net)
    batch = vfe({"points": dev(pts), "batch_size": 1})
    with torch.no_grad():
        out = net(batch)
    feat, coords, _ = oracle.dynamic_voxelize(pts, synthetic.PC_RANGE, synthetic.VOXEL_SIZE, synthetic.GRID_SIZE)
    cache = {}
    k3, s1, s2, p1 = [3, 3, 3], [1, 1, 1], [2, 2, 2], [1, 1, 1]

    def conv(convm, x, idx, shape, ks, st, pd, subm, key):
        if key in cache:
            out_idx, pairs, num, oshape = cache[key]
        else:
            oshape = shape if subm else oracle.conv_out_shape(shape, ks, st, pd, [1, 1, 1])
            out_idx, pairs, num = oracle.get_indice_pairs(idx, oshape, ks, st, pd, [1, 1, 1], subm)
            cache[key] = (out_idx, pairs, num, oshape)
        y = oracle.indice_conv(x, convm.weight.detach().cpu().numpy(), pairs, num, out_idx.shape[0], False, subm)
        return y, out_idx, oshape

    def basic_block(blk, x, idx, shape, key):
        y, _, _ = conv(blk.conv1, x, idx, shape, k3, s1, p1, True, key)
        y = np.maximum(bn_np(y, blk.bn1), 0)
        y, _, _ = conv(blk.conv2, y, idx, shape, k3, s1, p1, True, key)
        y = bn_np(y, blk.bn2)
        return np.maximum(y + x, 0).astype(np.float32)

    x, idx, shape = feat, coords, [41, 1504, 1504]
    y, idx, shape = conv(net.conv_input[0], x, idx, shape, k3, s1, p1, True, "res1")
    x = np.maximum(bn_np(y, net.conv_input[1]), 0)
    x = basic_block(net.conv1[0], x, idx, shape, "res1")
    x = basic_block(net.conv1[1], x, idx, shape, "res1")
    stages = {"x_conv1": (x, idx)}
    for name, stage, pad, key in (("x_conv2", net.conv2, p1, "res2"), ("x_conv3", net.conv3, p1, "res3"),
                                  ("x_conv4", net.conv4, [0, 1, 1], "res4")):
        y, idx, shape = conv(stage[0][0], x, idx, shape, k3, s2, pad, False, name + "_down")
        x = np.maximum(bn_np(y, stage[0][1]), 0)
        x = basic_block(stage[1], x, idx, shape, key)
        x = basic_block(stage[2], x, idx, shape, key)
        stages[name] = (x, idx)
    y, idx, shape = conv(net.conv_out[0], x, idx, shape, [3, 1, 1], [2, 1, 1], [0, 0, 0], False, "down2")
    x = np.maximum(bn_np(y, net.conv_out[1]), 0)
    for name, (ox, oidx) in stages.items():
        t = out["multi_scale_3d_features"][name]
        np.testing.assert_array_equal(t.indices.cpu().numpy(), oidx)
        np.testing.assert_allclose(t.features.cpu().numpy(), ox, rtol=2e-4, atol=2e-4)
    enc = out["encoded_spconv_tensor"]
    assert enc.spatial_shape == [2, 188, 188] and shape == [2, 188, 188]
    np.testing.assert_array_equal(enc.indices.cpu().numpy(), idx)
    np.testing.assert_allclose(enc.features.cpu().numpy(), x, rtol=2e-4, atol=2e-4)
    assert out["multi_scale_3d_strides"] == {"x_conv1": 1, "x_conv2": 2, "x_conv3": 4, "x_conv4": 8}

    # ---- HeightCompression on the encoded tensor: scatter_nd + permute + view of the reference
    hc = HeightCompression({"NUM_BEV_FEATURES": 256})
    with torch.no_grad():
        bev = hc(dict(out))["spatial_features"]
    assert tuple(bev.shape) == (1, 128 * 2, 188, 188) and hc.num_bev_features == 256
    want = np.zeros((1, 2, 188, 188, 128), dtype=np.float32)
    ii = enc.indices.cpu().numpy()
    want[ii[:, 0], ii[:, 1], ii[:, 2], ii[:, 3]] = enc.features.cpu().numpy()
    want = want.transpose(0, 4, 1, 2, 3).reshape(1, 256, 188, 188)
    np.testing.assert_array_equal(bev.cpu().numpy(), want)


def test_folded_batchnorm_matches_module_by_module():
    """The conv epilogue form (eval mode, no_grad) against the same network run module by module
    (train-mode graph walk with frozen statistics is not the same thing: use grad-enabled eval)."""
    torch.manual_seed(2)
    pts = synthetic.gen_voxel_frame(4, 20000)
    vfe = DynamicMeanVFE({}, 5, synthetic.VOXEL_SIZE, synthetic.GRID_SIZE, synthetic.PC_RANGE)
    for cls in (VoxelBackBone8x, VoxelResBackBone8x):
        net = cls({}, 5, synthetic.GRID_SIZE.tolist()).to(DEV).eval()
        randomise_bn(net)
        batch = vfe({"points": dev(pts), "batch_size": 1})
        with torch.no_grad():
            fused = net(dict(batch))["encoded_spconv_tensor"]
        with torch.enable_grad():  # the fusion is gated on no_grad: this walks conv, bn, relu separately
            plain = net(dict(batch))["encoded_spconv_tensor"]
        assert torch.equal(fused.indices, plain.indices)
        np.testing.assert_allclose(fused.features.cpu().numpy(), plain.features.detach().cpu().numpy(),
                                   rtol=1e-4, atol=1e-4)


def test_folded_batchnorm_follows_the_modules_state():
    """Folding is decided per BatchNorm module and recomputed from its current buffers: statistics
    edited through `.data` are seen, and a BatchNorm left in train mode inside an eval container is
    NOT folded (it normalises with batch statistics and updates its running ones, as PyTorch does)."""
    import torch.nn as nn
    from pcseqlearning_amd.pcdet.ops import spconv
    torch.manual_seed(3)
    pts = synthetic.gen_voxel_frame(6, 8000)
    vfe = DynamicMeanVFE({}, 5, synthetic.VOXEL_SIZE, synthetic.GRID_SIZE, synthetic.PC_RANGE)
    batch = vfe({"points": dev(pts), "batch_size": 1})
    x = spconv.SparseConvTensor(batch["voxel_features"], batch["voxel_coords"].int(),
                                synthetic.GRID_SIZE.tolist()[::-1], 1)
    seq = spconv.SparseSequential(spconv.SubMConv3d(5, 16, 3, padding=1, bias=False, indice_key="k"),
                                  nn.BatchNorm1d(16), nn.ReLU()).to(DEV).eval()
    bn = seq[1]

    def module_by_module():
        y = seq[0](x)
        return seq[2](bn(y.features))

    with torch.no_grad():
        bn.running_var.data.copy_(torch.rand(16, device=DEV) + 0.5)   # in-place through .data: no version bump
        bn.running_mean.data.normal_()
        a = seq(x).features
        np.testing.assert_allclose(a.cpu().numpy(), module_by_module().cpu().numpy(), rtol=1e-5, atol=1e-5)
        bn.running_var.data.mul_(4.0)
        b = seq(x).features
        assert not torch.allclose(a, b)
        np.testing.assert_allclose(b.cpu().numpy(), module_by_module().cpu().numpy(), rtol=1e-5, atol=1e-5)
        # BatchNorm back in train mode inside the eval container: batch statistics, running stats move
        bn.train()
        before = bn.running_mean.clone()
        c = seq(x).features
        assert not torch.equal(before, bn.running_mean)
        y = seq[0](x).features
        want = torch.relu((y - y.mean(0)) * torch.rsqrt(y.var(0, unbiased=False) + bn.eps) * bn.weight + bn.bias)
        np.testing.assert_allclose(c.cpu().numpy(), want.cpu().numpy(), rtol=1e-4, atol=1e-4)


def test_sparse_to_dense_vs_scatter():
    rng = np.random.default_rng(0)
    B, D, H, W, C = 2, 3, 17, 23, 20
    lin = rng.choice(B * D * H * W, 900, replace=False)
    idx = np.stack(np.unravel_index(lin, (B, D, H, W)), axis=1).astype(np.int32)
    feat = rng.normal(size=(900, C)).astype(np.float32)
    got = ops.sparse_to_dense(dev(feat), dev(idx), B, [D, H, W]).cpu().numpy()
    want = np.zeros((B, D, H, W, C), dtype=np.float32)
    want[idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]] = feat
    np.testing.assert_array_equal(got, want.transpose(0, 4, 1, 2, 3))
    t = SparseConvTensor(dev(feat), dev(idx), [D, H, W], B)
    np.testing.assert_array_equal(t.dense().cpu().numpy(), want.transpose(0, 4, 1, 2, 3))
    np.testing.assert_array_equal(t.dense(channels_first=False).cpu().numpy(), want)
    empty = ops.sparse_to_dense(torch.empty(0, C, device=DEV), torch.empty(0, 4, dtype=torch.int32, device=DEV), 1, [D, H, W])
    assert tuple(empty.shape) == (1, C, D, H, W) and float(empty.abs().sum()) == 0.0


def test_unet_up_block_modules_vs_oracle(oracle):
    """The decoder step of the reference's sparse UNet (spconv_unet.py:164-189: lateral
    SparseBasicBlock, concat with the bottom features, subM conv, channel reduction, inverse conv on
    the strided conv's rulebook) built from the mirrored modules, against an oracle chain. (UNetV2
    itself cannot be imported in the reference: its `post_processors` package is not in the tree.)"""
    from functools import partial
    from pcseqlearning_amd.pcdet.models.blocks import SparseBasicBlock, post_act_block
    from pcseqlearning_amd.pcdet.utils.spconv_utils import replace_feature, spconv
    torch.manual_seed(3)
    rng = np.random.default_rng(3)
    shape = [9, 40, 40]
    lin = rng.choice(int(np.prod(shape)), 1500, replace=False)
    idx = np.concatenate([np.zeros((1500, 1), np.int64), np.stack(np.unravel_index(lin, shape), 1)], 1).astype(np.int32)
    idx = idx[np.lexsort((idx[:, 3], idx[:, 2], idx[:, 1], idx[:, 0]))]
    feat = rng.normal(size=(1500, 8)).astype(np.float32)
    norm_fn = partial(torch.nn.BatchNorm1d, eps=1e-3, momentum=0.01)
    down = post_act_block(8, 16, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key="spconv2", conv_type="spconv").to(DEV).eval()
    conv_t = SparseBasicBlock(16, 16, indice_key="subm2", norm_fn=norm_fn).to(DEV).eval()
    conv_m = post_act_block(32, 16, 3, norm_fn=norm_fn, padding=1, indice_key="subm2").to(DEV).eval()
    conv_inv = post_act_block(16, 8, 3, norm_fn=norm_fn, indice_key="spconv2", conv_type="inverseconv").to(DEV).eval()
    for m in (down, conv_t, conv_m, conv_inv):
        randomise_bn(m)
    with torch.no_grad():
        x0 = spconv.SparseConvTensor(dev(feat), dev(idx), shape, 1)
        x_lat = down(x0)
        x_trans = conv_t(x_lat)
        x = replace_feature(x_trans, torch.cat((x_lat.features, x_trans.features), dim=1))
        x_m = conv_m(x)
        red = x.features.view(x.features.shape[0], x_m.features.shape[1], -1).sum(dim=2)
        x = replace_feature(x, x_m.features + red)
        up = conv_inv(x)
    # oracle chain
    k3, p1 = [3, 3, 3], [1, 1, 1]
    oshape = oracle.conv_out_shape(shape, k3, [2, 2, 2], p1, [1, 1, 1])
    d_idx, d_pairs, d_num = oracle.get_indice_pairs(idx, oshape, k3, [2, 2, 2], p1, [1, 1, 1], False)
    lat = np.maximum(bn_np(oracle.indice_conv(feat, down[0].weight.detach().cpu().numpy(), d_pairs, d_num,
                                              d_idx.shape[0], False, False), down[1]), 0)
    _, s_pairs, s_num = oracle.get_indice_pairs(d_idx, oshape, k3, [1, 1, 1], p1, [1, 1, 1], True)
    n2 = d_idx.shape[0]

    def subm(convm, v):
        return oracle.indice_conv(v, convm.weight.detach().cpu().numpy(), s_pairs, s_num, n2, False, True)

    t = np.maximum(bn_np(subm(conv_t.conv1, lat), conv_t.bn1), 0)
    t = np.maximum(bn_np(subm(conv_t.conv2, t), conv_t.bn2) + lat, 0)
    cat = np.concatenate([lat, t], axis=1)
    m = np.maximum(bn_np(subm(conv_m[0], cat), conv_m[1]), 0)
    mixed = (m + cat.reshape(n2, 16, -1).sum(axis=2)).astype(np.float32)
    inv = oracle.indice_conv(mixed, conv_inv[0].weight.detach().cpu().numpy(), d_pairs, d_num, idx.shape[0], True, False)
    want = np.maximum(bn_np(inv, conv_inv[1]), 0)
    np.testing.assert_array_equal(x_lat.indices.cpu().numpy(), d_idx)
    np.testing.assert_array_equal(up.indices.cpu().numpy(), idx)   # back on the input's voxels
    assert up.spatial_shape == shape
    np.testing.assert_allclose(up.features.cpu().numpy(), want, rtol=2e-4, atol=2e-4)


class _OracleSparseNet:
    """Walks this package's spconv modules with the CPU oracle's rulebook builder and conv (numpy):
    rulebooks cached by indice key like SparseConvTensor.indice_dict."""

    def __init__(self, oracle):
        self.o, self.books = oracle, {}

    def conv(self, m, idx, feat, shape):
        o = self.o
        w = m.weight.detach().cpu().numpy()
        k, st, pd, dl = (list(v) for v in (m.kernel_size, m.stride, m.padding, m.dilation))
        if m.inverse:
            in_idx, in_shape, pairs, num = self.books[m.indice_key]
            return in_idx, o.indice_conv(feat, w, pairs, num, in_idx.shape[0], True, False), in_shape
        if m.subm:
            if m.indice_key not in self.books:
                _, pairs, num = o.get_indice_pairs(idx, shape, k, [1, 1, 1], pd, dl, True)
                self.books[m.indice_key] = (idx, shape, pairs, num)
            _, _, pairs, num = self.books[m.indice_key]
            return idx, o.indice_conv(feat, w, pairs, num, idx.shape[0], False, True), shape
        oshape = o.conv_out_shape(shape, k, st, pd, dl)
        oidx, pairs, num = o.get_indice_pairs(idx, oshape, k, st, pd, dl, False)
        self.books[m.indice_key] = (idx, shape, pairs, num)
        return oidx, o.indice_conv(feat, w, pairs, num, oidx.shape[0], False, False), oshape

    def block(self, seq, idx, feat, shape):      # post_act_block: conv, BatchNorm1d, ReLU
        idx, y, shape = self.conv(seq[0], idx, feat, shape)
        return idx, np.maximum(bn_np(y, seq[1]), 0), shape

    def stage(self, seq, idx, feat, shape):
        for blk in seq:
            idx, feat, shape = self.block(blk, idx, feat, shape)
        return idx, feat, shape

    def basic(self, b, idx, feat, shape):        # SparseBasicBlock
        _, y, _ = self.conv(b.conv1, idx, feat, shape)
        y = np.maximum(bn_np(y, b.bn1), 0)
        _, y, _ = self.conv(b.conv2, idx, y, shape)
        return np.maximum(bn_np(y, b.bn2) + feat, 0)


@pytest.mark.parametrize("deep", [False, True])
def test_unet_v2_vs_oracle(oracle, deep):
    """UNetV2 (spconv_unet.py:13-271) end to end on a small voxel set against an oracle walk of the same
    modules: encoder 4 (or 6) stages, every decoder step back to the input's voxels; indices bit-exact,
    features 5e-4 (dozens of chained fp32 convs); output keys and centres as the reference computes them."""
    from pcseqlearning_amd.pcdet.models.backbones_3d import UNetV2
    torch.manual_seed(5)
    rng = np.random.default_rng(5)
    grid = [48, 40, 40 if deep else 16]      # x, y, z (the fifth stage needs the reference's 41 z-cells: 41-21-11-5-2)
    shape = [grid[2] + 1, grid[1], grid[0]]
    n = 2500
    lin = rng.choice(int(np.prod(shape[:1]) * shape[1] * shape[2]), n, replace=False)
    zyx = np.stack(np.unravel_index(lin, shape), 1)
    b = rng.integers(0, 2, (n, 1))
    idx = np.concatenate([b, zyx], 1).astype(np.int32)
    idx = idx[np.lexsort((idx[:, 3], idx[:, 2], idx[:, 1], idx[:, 0]))]
    feat = rng.normal(size=(n, 4)).astype(np.float32)
    rc = dict(grid_size=grid, voxel_size=[0.1, 0.1, 0.15], point_cloud_range=[0, -2, -1, 4.8, 2, 1.4],
              input_channels=4, scale=0.25)
    if deep:
        rc["point_cloud_range"] = [0, -2, -1, 4.8, 2, 5.0]
    cfg = {"CONV5": True, "CONV6": True} if deep else {}
    net = UNetV2(cfg, rc).to(DEV).eval()
    randomise_bn(net)
    bcoords = dev(idx[:, [0, 3, 2, 1]].astype(np.int64))       # (b, x, y, z) as the reference's voxel_bcoords
    with torch.no_grad():
        out = net(dict(voxel_feat=dev(feat), voxel_bcoords=bcoords, batch_size=2))
    # ---- oracle walk
    O = _OracleSparseNet(oracle)
    enc = {}
    cur = (idx, feat, shape)
    for s in (1, 2, 3, 4) + ((5, 6) if deep else ()):
        cur = O.stage(getattr(net, f"conv{s}"), *cur)
        enc[s] = cur

    def up_step(s, lateral, bottom_feat):
        l_idx, l_feat, l_shape = lateral
        t = O.basic(getattr(net, f"conv_up_t{s}"), l_idx, l_feat, l_shape)
        cat = np.concatenate([bottom_feat, t], axis=1)
        m_seq = getattr(net, f"conv_up_m{s}")
        _, m, _ = O.block(m_seq, l_idx, cat, l_shape)
        mixed = (m + cat.reshape(cat.shape[0], m.shape[1], -1).sum(axis=2)).astype(np.float32)
        return O.block(getattr(net, f"inv_conv{s}"), l_idx, mixed, l_shape)

    if deep:
        up = up_step(6, enc[6], enc[6][1])
        up = up_step(5, enc[5], up[1])
    else:
        up = enc[4]
    for s in (4, 3, 2, 1):
        up = up_step(s, enc[s], up[1])
    np.testing.assert_array_equal(up[0], idx)                    # back on the input's voxels
    np.testing.assert_allclose(out["unet_voxel_feat"].cpu().numpy(), up[1], rtol=5e-4, atol=5e-4)
    for i in range(5 if deep else 4):
        e_idx, e_feat, _ = enc[i + 1]
        np.testing.assert_allclose(out[f"spconv_unet_feat{5 - i}"].cpu().numpy(), e_feat, rtol=5e-4, atol=5e-4)
        times = np.array([1, 2, 4, 8, [8, 8, 16]][i], np.float32) if i < 4 else np.array([8, 8, 16], np.float32)
        vs = np.array(rc["voxel_size"], np.float32)
        want = e_idx[:, [3, 2, 1]].astype(np.float32) * (vs * times) + np.array(rc["point_cloud_range"][:3], np.float32) + vs * 0.5
        got = out[f"spconv_unet_bcenter{5 - i}"].cpu().numpy()
        np.testing.assert_array_equal(got[:, 0], e_idx[:, 0].astype(np.float32))
        np.testing.assert_allclose(got[:, 1:], want, rtol=0, atol=1e-5)
    centers = (idx[:, [3, 2, 1]].astype(np.float32) + 0.5) * np.array(rc["voxel_size"], np.float32) + \
        np.array(rc["point_cloud_range"][:3], np.float32)
    np.testing.assert_allclose(out["unet_voxel_bxyz"].cpu().numpy()[:, 1:], centers, rtol=0, atol=1e-5)
    assert out["unet_voxel_feat"].shape == (n, 8) and net.num_point_features == 8
