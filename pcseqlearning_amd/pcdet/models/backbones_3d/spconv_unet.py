"""`UNetV2`, the sparse-conv UNet of the reference (pcdet/models/backbones_3d/spconv_unet.py:13-271):
four encoder stages of three conv-BN-ReLU blocks (optional fifth / sixth), and per scale a decoder step
(`UR_block_forward`: residual block on the lateral features, concatenation with the features coming up,
subM conv, channel reduction, inverse conv on the encoder's strided rulebook) back to the input voxels.

Same constructor `(model_cfg, runtime_cfg)`, attribute / state_dict names (`conv1..conv6`,
`conv_up_t*`, `conv_up_m*`, `inv_conv*`), rulebook keys and output dictionary keys as the reference, on
this package's spconv modules (HIP rulebook + conv kernels; BatchNorm / ReLU folded into the conv
epilogues in eval mode). Built from a stage table instead of the reference's written-out layers.
Not mirrored: the `GLOBAL` dense branch (`BaseBEVBackbone`) and `POST_PROCESSING_CFG` (the reference's
`.post_processors` package is absent from its own tree, so it cannot construct them either): both are
refused in the constructor.
"""
from functools import partial

import torch
from torch import nn

from ...utils.spconv_utils import replace_feature, spconv
from ..blocks import SparseBasicBlock, post_act_block


def voxel_corners(voxel_coords, downsample_times, voxel_size, point_cloud_range):
    """common_utils.get_voxel_corners (:209-225): lower corner of voxel (z, y, x) in metres, xyz order."""
    xyz = voxel_coords[:, [2, 1, 0]].float()
    size = torch.tensor(voxel_size, device=xyz.device).float() * downsample_times
    return xyz * size + torch.tensor(point_cloud_range[0:3], device=xyz.device).float()


def voxel_centers(voxel_coords, downsample_times, voxel_size, point_cloud_range):
    """common_utils.get_voxel_centers (:191-207)."""
    xyz = voxel_coords[:, [2, 1, 0]].float()
    size = torch.tensor(voxel_size, device=xyz.device).float() * downsample_times
    return (xyz + 0.5) * size + torch.tensor(point_cloud_range[0:3], device=xyz.device).float()


class UNetV2(nn.Module):
    # encoder stage n: (in width, out width, stride, padding of its strided conv) in units of d1;
    # stage 1 has no strided conv
    ENCODER = {2: (1, 2, 2, 1), 3: (2, 4, 2, 1), 4: (4, 8, 2, (0, 1, 1)),
               5: (8, 8, (2, 1, 1), (0, 1, 1)), 6: (8, 8, 1, 1)}
    # decoder step n: (lateral width, width handed upwards) in units of d1
    DECODER = {6: (8, 8), 5: (8, 8), 4: (8, 4), 3: (4, 2), 2: (2, 1), 1: (1, 1)}
    DOWNSAMPLE = (1, 2, 4, 8, [8, 8, 16])

    def __init__(self, model_cfg, runtime_cfg, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        if model_cfg.get("GLOBAL", None) is not None or model_cfg.get("POST_PROCESSING_CFG", None):
            raise NotImplementedError("UNetV2: GLOBAL / POST_PROCESSING_CFG are not mirrored")
        grid_size = list(runtime_cfg["grid_size"])
        self.sparse_shape = [grid_size[2] + 1, grid_size[1], grid_size[0]]
        self.voxel_size = runtime_cfg["voxel_size"]
        self.point_cloud_range = runtime_cfg["point_cloud_range"]
        self.scale = runtime_cfg.get("scale", 1.0)
        d1 = int(32 * self.scale)
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        block = partial(post_act_block, norm_fn=norm_fn)

        def subm_pair(width, key):
            return [block(width, width, 3, padding=1, indice_key=key) for _ in range(2)]

        self.conv1 = spconv.SparseSequential(block(runtime_cfg["input_channels"], d1, 3, padding=1, indice_key="subm1"),
                                             *subm_pair(d1, "subm1"))
        self.stages = [1]
        for n in (2, 3, 4, 5, 6):
            if n >= 5 and not model_cfg.get(f"CONV{n}", False):
                setattr(self, f"conv{n}", None)
                continue
            cin, cout, stride, pad = self.ENCODER[n]
            setattr(self, f"conv{n}", spconv.SparseSequential(
                block(cin * d1, cout * d1, 3, stride=stride, padding=pad, indice_key=f"spconv{n}", conv_type="spconv"),
                *subm_pair(cout * d1, f"subm{n}")))
            self.stages.append(n)
        self.global_conv = None
        for n in sorted(self.stages, reverse=True):
            lat, up = self.DECODER[n]
            setattr(self, f"conv_up_t{n}", SparseBasicBlock(lat * d1, lat * d1, indice_key=f"subm{n}", norm_fn=norm_fn))
            # (the reference gives conv_up_m4..6 padding=1 and leaves it out for 1..3: a subM conv ignores it)
            setattr(self, f"conv_up_m{n}", block(2 * lat * d1, lat * d1, 3, padding=1, indice_key=f"subm{n}"))
            if n == 1:  # the last step stays on the input's voxels: a subM conv, not an inverse conv
                self.inv_conv1 = block(d1, d1, 3, padding=1, indice_key="subm1")
            else:
                setattr(self, f"inv_conv{n}", block(lat * d1, up * d1, 3, indice_key=f"spconv{n}", conv_type="inverseconv"))
        self.num_point_features = d1
        runtime_cfg["input_key"] = "unet_voxel"
        runtime_cfg["num_point_features"] = self.num_point_features
        self.post_processor = None

    def UR_block_forward(self, x_lateral, x_bottom, conv_t, conv_m, conv_inv):
        x_trans = conv_t(x_lateral)
        x = replace_feature(x_trans, torch.cat((x_bottom.features, x_trans.features), dim=1))
        x_m = conv_m(x)
        x = self.channel_reduction(x, x_m.features.shape[1])
        return conv_inv(replace_feature(x, x_m.features + x.features))

    @staticmethod
    def channel_reduction(x, out_channels):
        n, in_channels = x.features.shape
        assert in_channels % out_channels == 0 and in_channels >= out_channels
        return replace_feature(x, x.features.view(n, out_channels, -1).sum(dim=2))

    def _up(self, n, lateral, bottom):
        return self.UR_block_forward(lateral, bottom, getattr(self, f"conv_up_t{n}"), getattr(self, f"conv_up_m{n}"),
                                     getattr(self, f"inv_conv{n}"))

    def forward(self, batch_dict):
        coords = batch_dict["voxel_bcoords"][:, [0, 3, 2, 1]]
        x = spconv.SparseConvTensor(features=batch_dict["voxel_feat"], indices=coords.int(),
                                    spatial_shape=self.sparse_shape, batch_size=batch_dict["batch_size"])
        enc = {}
        for n in (1, 2, 3, 4):
            x = getattr(self, f"conv{n}")(x)
            enc[n] = x
        if self.conv5 is not None:
            enc[5] = self.conv5(enc[4])
            if self.conv6 is not None:
                x_conv6 = self.conv6(enc[5])
                x_up6 = self._up(6, x_conv6, x_conv6)
            else:
                x_up6 = enc[5]
            up = self._up(5, enc[5], x_up6)
        else:
            up = enc[4]
        voxel_size = torch.tensor(self.voxel_size, device=coords.device).float()
        for i in range(5 if self.conv5 is not None else 4):
            t = enc[i + 1]
            times = torch.tensor(self.DOWNSAMPLE[i]).to(t.features)
            corners = voxel_corners(t.indices[:, 1:], times, self.voxel_size, self.point_cloud_range) + voxel_size * 0.5
            batch_dict[f"spconv_unet_bcenter{5 - i}"] = torch.cat([t.indices[:, 0:1], corners], dim=-1)
            batch_dict[f"spconv_unet_feat{5 - i}"] = t.features
        for n in (4, 3, 2, 1):
            up = self._up(n, enc[n], up)
        batch_dict["unet_voxel_feat"] = up.features
        centers = voxel_centers(up.indices[:, 1:], 1, self.voxel_size, self.point_cloud_range)
        batch_dict["unet_voxel_bxyz"] = torch.cat((up.indices[:, 0:1].float(), centers), dim=1)
        return batch_dict
