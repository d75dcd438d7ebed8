"""VoxelBackBone8x (pcdet/models/backbones_3d/spconv_backbone.py:47-158): conv_input (subm1) ->
conv1 -> conv2 (stride 2) -> conv3 (stride 2) -> conv4 (stride 2, pad (0,1,1)) -> conv_out
(k (3,1,1), s (2,1,1)); 12 sparse convs over 8 rulebooks; channels 16/32/64/64/128;
BatchNorm1d(eps 1e-3, momentum 0.01) + ReLU after every conv."""
from functools import partial

import torch.nn as nn

from ...utils.spconv_utils import spconv
from ..blocks import SparseBasicBlock, post_act_block


class VoxelBackBone8x(nn.Module):
    def __init__(self, model_cfg, input_channels, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        self.sparse_shape = [int(g) for g in list(grid_size)[::-1]]
        self.sparse_shape[0] += 1
        self.conv_input = spconv.SparseSequential(
            spconv.SubMConv3d(input_channels, 16, 3, padding=1, bias=False, indice_key="subm1"),
            norm_fn(16), nn.ReLU())
        block = post_act_block
        self.conv1 = spconv.SparseSequential(
            block(16, 16, 3, norm_fn=norm_fn, padding=1, indice_key="subm1"))
        self.conv2 = spconv.SparseSequential(
            block(16, 32, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key="spconv2", conv_type="spconv"),
            block(32, 32, 3, norm_fn=norm_fn, padding=1, indice_key="subm2"),
            block(32, 32, 3, norm_fn=norm_fn, padding=1, indice_key="subm2"))
        self.conv3 = spconv.SparseSequential(
            block(32, 64, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key="spconv3", conv_type="spconv"),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key="subm3"),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key="subm3"))
        self.conv4 = spconv.SparseSequential(
            block(64, 64, 3, norm_fn=norm_fn, stride=2, padding=(0, 1, 1), indice_key="spconv4", conv_type="spconv"),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key="subm4"),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key="subm4"))
        last_pad = self.model_cfg.get("last_pad", 0) if hasattr(self.model_cfg, "get") else 0
        self.conv_out = spconv.SparseSequential(
            spconv.SparseConv3d(64, 128, (3, 1, 1), stride=(2, 1, 1), padding=last_pad, bias=False,
                                indice_key="spconv_down2"),
            norm_fn(128), nn.ReLU())
        self.num_point_features = 128
        self.backbone_channels = {"x_conv1": 16, "x_conv2": 32, "x_conv3": 64, "x_conv4": 64}

    def forward(self, batch_dict):
        x = spconv.SparseConvTensor(features=batch_dict["voxel_features"],
                                    indices=batch_dict["voxel_coords"].int(),
                                    spatial_shape=self.sparse_shape, batch_size=batch_dict["batch_size"])
        x = self.conv_input(x)
        x_conv1 = self.conv1(x)
        x_conv2 = self.conv2(x_conv1)
        x_conv3 = self.conv3(x_conv2)
        x_conv4 = self.conv4(x_conv3)
        out = self.conv_out(x_conv4)
        batch_dict.update({
            "encoded_spconv_tensor": out,
            "encoded_spconv_tensor_stride": 8,
            "multi_scale_3d_features": {"x_conv1": x_conv1, "x_conv2": x_conv2, "x_conv3": x_conv3,
                                        "x_conv4": x_conv4},
            "multi_scale_3d_strides": {"x_conv1": 1, "x_conv2": 2, "x_conv3": 4, "x_conv4": 8},
        })
        return batch_dict


class VoxelResBackBone8x(nn.Module):
    """pcdet/models/backbones_3d/spconv_backbone.py:161-271: the residual variant,
    two SparseBasicBlocks per stage on the stage's subM rulebook ('res1'..'res4'), channels
    16/32/64/128, same strided convs and outputs as VoxelBackBone8x."""

    def __init__(self, model_cfg, input_channels, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        self.sparse_shape = [int(g) for g in list(grid_size)[::-1]]
        self.sparse_shape[0] += 1
        self.conv_input = spconv.SparseSequential(
            spconv.SubMConv3d(input_channels, 16, 3, padding=1, bias=False, indice_key="res1"),
            norm_fn(16), nn.ReLU())
        block = post_act_block
        self.conv1 = spconv.SparseSequential(
            SparseBasicBlock(16, 16, norm_fn=norm_fn, indice_key="res1"),
            SparseBasicBlock(16, 16, norm_fn=norm_fn, indice_key="res1"))
        self.conv2 = spconv.SparseSequential(
            block(16, 32, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key="spconv2", conv_type="spconv"),
            SparseBasicBlock(32, 32, norm_fn=norm_fn, indice_key="res2"),
            SparseBasicBlock(32, 32, norm_fn=norm_fn, indice_key="res2"))
        self.conv3 = spconv.SparseSequential(
            block(32, 64, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key="spconv3", conv_type="spconv"),
            SparseBasicBlock(64, 64, norm_fn=norm_fn, indice_key="res3"),
            SparseBasicBlock(64, 64, norm_fn=norm_fn, indice_key="res3"))
        self.conv4 = spconv.SparseSequential(
            block(64, 128, 3, norm_fn=norm_fn, stride=2, padding=(0, 1, 1), indice_key="spconv4", conv_type="spconv"),
            SparseBasicBlock(128, 128, norm_fn=norm_fn, indice_key="res4"),
            SparseBasicBlock(128, 128, norm_fn=norm_fn, indice_key="res4"))
        last_pad = self.model_cfg.get("last_pad", 0) if hasattr(self.model_cfg, "get") else 0
        self.conv_out = spconv.SparseSequential(
            spconv.SparseConv3d(128, 128, (3, 1, 1), stride=(2, 1, 1), padding=last_pad, bias=False,
                                indice_key="spconv_down2"),
            norm_fn(128), nn.ReLU())
        self.num_point_features = 128
        self.backbone_channels = {"x_conv1": 16, "x_conv2": 32, "x_conv3": 64, "x_conv4": 128}

    def forward(self, batch_dict):
        x = spconv.SparseConvTensor(features=batch_dict["voxel_features"],
                                    indices=batch_dict["voxel_coords"].int(),
                                    spatial_shape=self.sparse_shape, batch_size=batch_dict["batch_size"])
        x = self.conv_input(x)
        x_conv1 = self.conv1(x)
        x_conv2 = self.conv2(x_conv1)
        x_conv3 = self.conv3(x_conv2)
        x_conv4 = self.conv4(x_conv3)
        out = self.conv_out(x_conv4)
        batch_dict.update({
            "encoded_spconv_tensor": out,
            "encoded_spconv_tensor_stride": 8,
            "multi_scale_3d_features": {"x_conv1": x_conv1, "x_conv2": x_conv2, "x_conv3": x_conv3,
                                        "x_conv4": x_conv4},
            "multi_scale_3d_strides": {"x_conv1": 1, "x_conv2": 2, "x_conv3": 4, "x_conv4": 8},
        })
        return batch_dict
