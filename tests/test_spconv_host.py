"""Host-side pieces of the sparse-conv mirror that need no GPU: the BatchNorm folding used by the
conv epilogue, output-size helpers, the backbones' module trees (state-dict keys are the
reference's) and the table-driven stage construction."""
import numpy as np
import torch

from pcseqlearning_amd.pcdet.models.backbones_3d import VoxelBackBone8x, VoxelResBackBone8x
from pcseqlearning_amd.pcdet.ops.spconv import ops as sp_ops
from pcseqlearning_amd.pcdet.ops.spconv.modules import fold_batchnorm


def test_fold_batchnorm_matches_eval_batchnorm():
    torch.manual_seed(0)
    bn = torch.nn.BatchNorm1d(12, eps=1e-3, momentum=0.01).eval()
    bn.running_mean.normal_(0, 0.3)
    bn.running_var.uniform_(0.4, 1.6)
    bn.weight.data.uniform_(0.5, 1.5)
    bn.bias.data.normal_(0, 0.2)
    x = torch.randn(50, 12)
    scale, shift = fold_batchnorm(bn)
    np.testing.assert_allclose((x * scale + shift).numpy(), bn(x).detach().numpy(), rtol=1e-5, atol=1e-6)
    bn.running_mean.add_(1.0)                        # a statistic changes: the fold follows (nothing is cached)
    s2, t2 = fold_batchnorm(bn)
    np.testing.assert_allclose((x * s2 + t2).numpy(), bn(x).detach().numpy(), rtol=1e-5, atol=1e-6)
    assert not torch.equal(t2, shift)


def test_output_size_helpers():
    assert sp_ops.get_conv_output_size([41, 1504, 1504], [3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1]) == [21, 752, 752]
    assert sp_ops.get_conv_output_size([5, 188, 188], [3, 1, 1], [2, 1, 1], [0, 0, 0], [1, 1, 1]) == [2, 188, 188]
    assert sp_ops.get_deconv_output_size([21, 752, 752], [3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1], [0, 0, 0]) == [41, 1503, 1503]


def test_backbone_module_trees():
    a = VoxelBackBone8x({}, 5, [1504, 1504, 40])
    b = VoxelResBackBone8x({}, 5, [1504, 1504, 40])
    assert a.sparse_shape == b.sparse_shape == [41, 1504, 1504]
    ka, kb = set(a.state_dict()), set(b.state_dict())
    # the reference's parameter names (spconv_backbone.py:47-271)
    for k in ("conv_input.0.weight", "conv1.0.0.weight", "conv2.0.0.weight", "conv2.2.1.running_var",
              "conv4.0.0.weight", "conv_out.0.weight", "conv_out.1.bias"):
        assert k in ka, k
    for k in ("conv_input.0.weight", "conv1.0.conv1.weight", "conv1.1.bn2.weight", "conv2.0.0.weight",
              "conv2.1.conv2.weight", "conv4.2.bn1.running_mean", "conv_out.0.weight"):
        assert k in kb, k
    assert tuple(a.conv4[0][0].weight.shape) == (3, 3, 3, 64, 64) and a.conv4[0][0].padding == [0, 1, 1]
    assert tuple(b.conv4[0][0].weight.shape) == (3, 3, 3, 64, 128)
    assert tuple(a.conv_out[0].weight.shape) == (3, 1, 1, 64, 128) and a.conv_out[0].stride == [2, 1, 1]
    assert len(a.conv1) == 1 and len(a.conv2) == 3 and len(b.conv1) == 2 and len(b.conv3) == 3
    assert a.backbone_channels == {"x_conv1": 16, "x_conv2": 32, "x_conv3": 64, "x_conv4": 64}
    assert b.backbone_channels == {"x_conv1": 16, "x_conv2": 32, "x_conv3": 64, "x_conv4": 128}
    # rulebook keys: strided convs spconv2..4 + the stage's own subM key
    assert a.conv2[0][0].indice_key == "spconv2" and a.conv2[1][0].indice_key == "subm2"
    assert b.conv3[0][0].indice_key == "spconv3" and b.conv3[2].conv1.indice_key == "res3"
    assert a.conv_input[0].indice_key == "subm1" and b.conv_input[0].indice_key == "res1"


def test_unet_v2_module_tree():
    """UNetV2's parameter names and shapes as the reference lays them out (spconv_unet.py:20-124)."""
    import pytest
    from pcseqlearning_amd.pcdet.models.backbones_3d import UNetV2
    rc = dict(grid_size=[1504, 1504, 40], voxel_size=[0.1, 0.1, 0.15], point_cloud_range=[-75.2, -75.2, -2, 75.2, 75.2, 4],
              input_channels=5)
    net = UNetV2({}, dict(rc))
    assert net.sparse_shape == [41, 1504, 1504] and net.num_point_features == 32 and net.conv5 is None
    sd = net.state_dict()
    assert sd["conv1.0.0.weight"].shape[-2:] == (5, 32) and sd["conv4.0.0.weight"].shape[-2:] == (128, 256)
    assert sd["conv_up_m4.0.weight"].shape[-2:] == (512, 256) and sd["inv_conv4.0.weight"].shape[-2:] == (256, 128)
    assert sd["inv_conv2.0.weight"].shape[-2:] == (64, 32) and sd["inv_conv1.0.weight"].shape[-2:] == (32, 32)
    assert {"conv_up_t1.conv1.weight", "conv_up_t3.bn2.running_var", "conv3.2.1.weight"} <= set(sd)
    assert not any(k.startswith(("conv5", "conv6", "conv_up_t5", "inv_conv6")) for k in sd)
    deep = UNetV2({"CONV5": True, "CONV6": True}, dict(rc))
    assert list(deep.conv5[0][0].stride) == [2, 1, 1] and list(deep.conv5[0][0].padding) == [0, 1, 1]
    assert {"conv_up_m6.0.weight", "inv_conv5.0.weight", "conv6.2.0.weight"} <= set(deep.state_dict())
    assert net.conv4[0][0].indice_key == "spconv4" and net.inv_conv4[0].indice_key == "spconv4" and net.inv_conv4[0].inverse
    assert net.inv_conv1[0].subm and net.inv_conv1[0].indice_key == "subm1"
    with pytest.raises(NotImplementedError):
        UNetV2({"GLOBAL": {"x": 1}}, dict(rc))
