"""HeightCompression (pcdet/models/backbones_2d/map_to_bev/height_compression.py:4-27): the encoded
sparse tensor as a dense BEV map [N, C*D, H, W]. `dense()` is one HIP fill +
scatter into the channels-first layout (`pcs_sparse_to_dense`), so the view below is free."""
import torch.nn as nn


class HeightCompression(nn.Module):
    def __init__(self, model_cfg, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_bev_features = self.model_cfg["NUM_BEV_FEATURES"] if "NUM_BEV_FEATURES" in self.model_cfg else None

    def forward(self, batch_dict):
        spatial_features = batch_dict["encoded_spconv_tensor"].dense()
        N, C, D, H, W = spatial_features.shape
        batch_dict["spatial_features"] = spatial_features.view(N, C * D, H, W)
        batch_dict["spatial_features_stride"] = batch_dict["encoded_spconv_tensor_stride"]
        return batch_dict
