"""The two 8x sparse-conv backbones of the reference (pcdet/models/backbones_3d/spconv_backbone.py):

VoxelBackBone8x (:47-158)      conv_input (subM, key subm1) -> conv1 -> conv2 -> conv3 -> conv4 -> conv_out;
                               every stage after the first = strided 3x3x3 conv + two subM convs on the
                               stage's rulebook; channels 16 / 32 / 64 / 64, conv_out (3,1,1)/(2,1,1) -> 128
VoxelResBackBone8x (:161-271)  same skeleton with two SparseBasicBlocks per stage (keys res1..res4),
                               channels 16 / 32 / 64 / 128

Both are built from one stage table; module names and Sequential indices (hence state-dict keys) are the
reference's. BatchNorm1d(eps 1e-3, momentum 0.01) + ReLU follow every conv.
"""
from functools import partial

from torch import nn

from ...utils.spconv_utils import spconv
from ..blocks import SparseBasicBlock, post_act_block

# (name, out channels, padding of the stage's strided conv); stage 1 has no strided conv
_STAGES = (("conv1", 0, None), ("conv2", 1, 1), ("conv3", 2, 1), ("conv4", 3, (0, 1, 1)))


class _Backbone8x(nn.Module):
    channels = None     # per stage
    residual = False

    def __init__(self, model_cfg, input_channels, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        zyx = [int(g) for g in list(grid_size)[::-1]]
        self.sparse_shape = [zyx[0] + 1] + zyx[1:]
        ch = self.channels
        key1 = "res1" if self.residual else "subm1"
        self.conv_input = spconv.SparseSequential(
            spconv.SubMConv3d(input_channels, ch[0], 3, padding=1, bias=False, indice_key=key1),
            norm_fn(ch[0]), nn.ReLU())
        for name, si, pad in _STAGES:
            width, n = ch[si], si + 1
            key = f"res{n}" if self.residual else f"subm{n}"
            layers = []
            if pad is not None:
                layers.append(post_act_block(ch[si - 1], width, 3, norm_fn=norm_fn, stride=2, padding=pad,
                                             indice_key=f"spconv{n}", conv_type="spconv"))
            repeats = 2 if (self.residual or pad is not None) else 1
            for _ in range(repeats):
                layers.append(SparseBasicBlock(width, width, norm_fn=norm_fn, indice_key=key) if self.residual
                              else post_act_block(width, width, 3, norm_fn=norm_fn, padding=1, indice_key=key))
            setattr(self, name, spconv.SparseSequential(*layers))
        last_pad = self.model_cfg.get("last_pad", 0) if hasattr(self.model_cfg, "get") else 0
        self.conv_out = spconv.SparseSequential(
            spconv.SparseConv3d(ch[3], 128, (3, 1, 1), stride=(2, 1, 1), padding=last_pad, bias=False,
                                indice_key="spconv_down2"),
            norm_fn(128), nn.ReLU())
        self.num_point_features = 128
        self.backbone_channels = {f"x_conv{i + 1}": c for i, c in enumerate(ch)}

    def forward(self, batch_dict):
        x = spconv.SparseConvTensor(features=batch_dict["voxel_features"], indices=batch_dict["voxel_coords"].int(),
                                    spatial_shape=self.sparse_shape, batch_size=batch_dict["batch_size"])
        x = self.conv_input(x)
        feats = {}
        for name, si, _ in _STAGES:
            x = getattr(self, name)(x)
            feats[f"x_conv{si + 1}"] = x
        batch_dict["encoded_spconv_tensor"] = self.conv_out(x)
        batch_dict["encoded_spconv_tensor_stride"] = 8
        batch_dict["multi_scale_3d_features"] = feats
        batch_dict["multi_scale_3d_strides"] = {f"x_conv{i + 1}": 2 ** i for i in range(4)}
        return batch_dict


class VoxelBackBone8x(_Backbone8x):
    channels = (16, 32, 64, 64)


class VoxelResBackBone8x(_Backbone8x):
    channels = (16, 32, 64, 128)
    residual = True
