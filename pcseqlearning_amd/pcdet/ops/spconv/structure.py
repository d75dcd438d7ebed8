"""SparseConvTensor (pcdet/ops/spconv/structure.py:21-69): features [V,C], indices [V,4] int32
(batch, z, y, x), spatial_shape, batch_size and the rulebook cache `indice_dict`."""
import numpy as np
import torch


class SparseConvTensor(object):
    def __init__(self, features, indices, spatial_shape, batch_size, grid=None):
        self.features = features
        self.indices = indices if indices.dtype == torch.int32 else indices.int()
        self.spatial_shape = list(spatial_shape)
        self.batch_size = batch_size
        self.indice_dict = {}
        self.grid = grid  # accepted for API compatibility; the HIP rulebook needs no dense grid

    @property
    def spatial_size(self):
        return np.prod(self.spatial_shape)

    def find_indice_pair(self, key):
        if key is None:
            return None
        return self.indice_dict.get(key, None)

    def replace_feature(self, new_features):
        self.features = new_features
        return self

    def dense(self, channels_first=True):
        if (channels_first and self.features.is_cuda and len(self.spatial_shape) == 3
                and self.features.dtype == torch.float32 and not self.features.requires_grad):
            # one fill + one scatter straight into [B,C,D,H,W] (pcs_sparse_to_dense)
            from .... import ops as _ops
            return _ops.sparse_to_dense(self.features, self.indices, self.batch_size, self.spatial_shape)
        shape = [self.batch_size] + list(self.spatial_shape) + [self.features.shape[1]]
        out = torch.zeros(*shape, dtype=self.features.dtype, device=self.features.device)
        idx = self.indices.long()
        out[tuple(idx[:, d] for d in range(idx.shape[1]))] = self.features  # (2-D and 3-D alike)
        if not channels_first:
            return out
        nd = len(self.spatial_shape)
        return out.permute(0, nd + 1, *range(1, nd + 1)).contiguous()

    @property
    def sparity(self):
        return self.indices.shape[0] / np.prod(self.spatial_shape) / self.batch_size
