"""GridSampling3D (pcdet/models/model_utils/grid_sampling.py:7-49) on the HIP voxel-key /
sorted-unique / ordered-segment-mean kernels."""
import torch
from torch import nn

from .... import ops


class GridSampling3D(nn.Module):
    def __init__(self, grid_size):
        super().__init__()
        self._grid_size = grid_size
        if isinstance(grid_size, (list, tuple)):
            size = [1.0] + [float(g) for g in grid_size]
        else:
            size = [1.0] + [float(grid_size)] * 3
        assert len(size) == 4, "Expecting 4D grid size."
        self.register_buffer("grid_size", torch.tensor(size).float())
        self._size_host = size

    def segments(self, points):
        """Voxel structure of `points` [N,4]: ops.Segments (order, seg_start, inv, num)."""
        ids = ops.grid_cluster_ids(points.contiguous(), self._size_host, 0.5)
        return ops.unique_i64(ids, want_uniq=False)

    def forward(self, points, return_inverse=False):
        seg = self.segments(points)
        sampled = ops.segment_reduce(points, seg, mean=True)
        if return_inverse:
            return sampled, seg.inv
        return sampled

    def extra_repr(self):
        return "grid size {}".format(self._grid_size)
