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
