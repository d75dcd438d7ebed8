"""Module fronts of pcdet/ops/torch_hash (torch_hash_modules.py:10-93 `RadiusGraph`, :96-129 `ChamferDistance` of the
reference): the same constructors, call signatures and return values, on the four drop-in table ops of
libpcs_hip.so (hash_insert_gpu / radius_graph_gpu on the reference's own table layout, include/pcs.h).

`RadiusGraph(max_num_points, ndim)(ref [N, 1 + ndim], query [M, 1 + ndim], radius, num_neighbors, sort_by_dist)`
returns edges [2, E] (row 0: index into ref, row 1: index into query), grouped by query; `radius` a float or a
per-query tensor; the first column is the batch / frame index and separates the clouds (cell size 1 - 1e-3).
Where the reference's walk order depends on an atomicCAS race the order here is canonical: (dist2, ref index) when
sorted, ref index when not (the drop-ins' rule)."""
import torch
from torch import nn

from .... import ops


class RadiusGraph(nn.Module):
    def __init__(self, max_num_points=400000, ndim=3):
        super().__init__()
        self.ndim = ndim
        self.max_num_points = max_num_points  # (kept for repr / ChamferDistance: the table is sized per call)
        self.register_buffer("qmin", torch.tensor([0] + [-1] * ndim, dtype=torch.int32), persistent=False)
        self.register_buffer("qmax", torch.tensor([0] + [1] * ndim, dtype=torch.int32), persistent=False)

    @torch.no_grad()
    def forward(self, ref, query, radius, num_neighbors, sort_by_dist=False):
        if ref.shape[1] != self.ndim + 1 or query.shape[1] != self.ndim + 1:
            raise ValueError(f"points must have {self.ndim + 1} dimensions")
        dev = ref.device
        ref = ref.float().contiguous()
        query = query.float().contiguous().to(dev)
        if not torch.is_tensor(radius):
            radius = query.new_zeros(query.shape[0]) + float(radius)
        radius = radius.float().contiguous()
        # the reference's cell grid: cell = the largest radius (1 - 1e-3 along the batch column), origin two cells
        # below the clouds' minimum, coordinates by round-half-even + 1 (torch_hash_modules.py:53-62)
        r_max = float(radius.max().item())
        cell = torch.tensor([1 - 1e-3] + [r_max] * self.ndim, device=dev)
        both = torch.cat([ref, query], dim=0)
        lo = both.min(0)[0] - cell * 2
        hi = both.max(0)[0] + cell * 2
        c_ref = torch.round((ref - lo) / cell).long() + 1
        c_query = torch.round((query - lo) / cell).long() + 1
        dims = torch.round((hi - lo) / cell).long() + 3
        slots = max(2 * ref.shape[0], 1)
        keys = torch.full((slots,), -1, dtype=torch.int64, device=dev)
        values = torch.zeros(slots, self.ndim + 1, dtype=torch.float32, device=dev)
        reverse = torch.zeros(slots, dtype=torch.int64, device=dev)
        ops.hash_insert_gpu(keys, values, reverse, dims, c_ref, ref)
        edges = ops.radius_graph_gpu(keys, values, reverse, dims, c_query, query, self.qmin.to(dev), self.qmax.to(dev),
                                     radius, int(num_neighbors), bool(sort_by_dist))
        return edges.T

    def extra_repr(self):
        return f"ndim={self.ndim}"


class ChamferDistance(nn.Module):
    """CD = mean |src - nn_target(src)|^2 + mean |target - nn_src(target)|^2 over the pairs closer than `radius`."""

    def __init__(self, max_num_points=400000, ndim=3, radius_graph=None):
        super().__init__()
        if radius_graph is not None:
            self.radius_graph = radius_graph
            self.ndim = radius_graph.ndim
            self.max_num_points = radius_graph.max_num_points
        else:
            self.radius_graph = RadiusGraph(max_num_points, ndim=ndim)
            self.ndim = ndim
            self.max_num_points = max_num_points

    def forward(self, src_bxyz, target_bxyz, radius):
        f_src, f_tgt = self.radius_graph(src_bxyz, target_bxyz, radius, 1, sort_by_dist=True)
        b_tgt, b_src = self.radius_graph(target_bxyz, src_bxyz, radius, 1, sort_by_dist=True)
        d_fwd = (src_bxyz[f_src] - target_bxyz[f_tgt]).square().sum(-1).mean()
        d_bwd = (src_bxyz[b_src] - target_bxyz[b_tgt]).square().sum(-1).mean()
        return d_fwd + d_bwd

    def __repr__(self):
        return f"ChamferDistance(ndim={self.ndim}, max_npoints={self.max_num_points})"
