"""HeightCompression (reference: pcdet/models/backbones_2d/map_to_bev/height_compression.py:4-27; SURVEY 8f
row f3): the encoded sparse tensor as a dense bird's-eye-view map, the height axis folded into the
channels: [N, C, D, H, W] -> [N, C*D, H, W]. `dense()` is ONE HIP fill + scatter straight into the
channels-first layout (`pcs_sparse_to_dense`), so the fold is a view."""
from torch import nn


class HeightCompression(nn.Module):
    def __init__(self, model_cfg, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_bev_features = model_cfg["NUM_BEV_FEATURES"] if "NUM_BEV_FEATURES" in model_cfg else None

    def forward(self, batch_dict):
        volume = batch_dict["encoded_spconv_tensor"].dense()
        batch_dict.update(spatial_features=volume.flatten(1, 2),
                          spatial_features_stride=batch_dict["encoded_spconv_tensor_stride"])
        return batch_dict
