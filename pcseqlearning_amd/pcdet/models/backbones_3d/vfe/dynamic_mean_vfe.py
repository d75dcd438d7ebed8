"""DynamicMeanVFE (pcdet/models/backbones_3d/vfe/dynamic_mean_vfe.py:14-76) as one HIP call:
floor-quantise, in-range mask, x-major merge key, sorted unique, ordered per-voxel mean,
coords emitted as (b, z, y, x)."""
import torch
import torch.nn as nn

from ..... import ops


class DynamicMeanVFE(nn.Module):
    def __init__(self, model_cfg, num_point_features, voxel_size, grid_size, point_cloud_range, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_point_features = num_point_features
        self.voxel_size = [float(v) for v in voxel_size]
        self.grid_size = [int(g) for g in grid_size]
        self.point_cloud_range = [float(v) for v in point_cloud_range]

    def get_output_feature_dim(self):
        return self.num_point_features

    @torch.no_grad()
    def forward(self, batch_dict, **kwargs):
        points = batch_dict["points"]  # (batch_idx, x, y, z, features...)
        feat, coords, _ = ops.dynamic_voxelize(points, self.point_cloud_range, self.voxel_size, self.grid_size)
        batch_dict["voxel_features"] = feat.contiguous()
        batch_dict["voxel_coords"] = coords.contiguous()
        return batch_dict
