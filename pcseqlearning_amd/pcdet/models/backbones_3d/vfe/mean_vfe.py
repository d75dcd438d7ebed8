"""MeanVFE (pcdet/models/backbones_3d/vfe/mean_vfe.py:14-53): mean of the (<= max_points) points
stored per hard voxel: sum over the point axis / clamp(count, 1)."""
import torch
import torch.nn as nn


class MeanVFE(nn.Module):
    def __init__(self, model_cfg, num_point_features, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_point_features = num_point_features

    def get_output_feature_dim(self):
        return self.num_point_features

    def forward(self, batch_dict, **kwargs):
        voxels, num = batch_dict["voxels"], batch_dict["voxel_num_points"]
        mean = voxels.sum(dim=1) / torch.clamp_min(num.view(-1, 1), min=1.0).type_as(voxels)
        batch_dict["voxel_features"] = mean.contiguous()
        return batch_dict
