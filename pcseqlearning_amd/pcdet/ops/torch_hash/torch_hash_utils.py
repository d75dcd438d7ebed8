"""`HashTable`, the stateful front of pcdet/ops/torch_hash (torch_hash_utils.py:9-363 of the reference): one table of
`size` slots kept on the device, filled by `hash_into_gpu` (or by `find_corres` / `voxel_graph`, which insert and query
in one call) and queried by the `*_step2` methods. Same constructor, method names, arguments and return layouts, on the
drop-in table ops of libpcs_hip.so (`hash_insert_gpu`, `correspondence`, `radius_graph_gpu`, `points_in_radius_gpu`:
the reference's own table layout and walk).

Points are [n, D] rows whose LAST column is the temporal one (`temporal_offset` restricts a query to the cells that
many frames away); cells are `voxel_size` wide from `pc_range[:D]`, coordinates by round-half-even + 1 and `dims` =
extent / voxel_size + 3, exactly as the reference forms them. `voxel_graph*` call an op (`voxel_graph_gpu`) that the
reference's extension does not export (torch_hash_api.cpp:9-15): here they are the radius graph with
`max_num_neighbors` neighbours per query in table-walk order, returned in the layout the reference's method builds
([2, E]: row 0 query index, row 1 reference index)."""
import torch

from .... import ops


class HashTable:
    def __init__(self, size=2 ** 20, dtype=torch.float32, device="cuda:0"):
        self.size = int(size)
        self.device = torch.device(device)
        self.keys = torch.full((self.size,), -1, dtype=torch.int64, device=self.device)
        self.values = torch.zeros(self.size, 4, dtype=dtype, device=self.device)
        self.reverse_indices = torch.zeros(self.size, dtype=torch.int64, device=self.device)
        self.qmin = torch.full((4,), -1, dtype=torch.int32, device=self.device)
        self.qmax = torch.full((4,), 1, dtype=torch.int32, device=self.device)
        self.corres_pool_size = 10000
        self.corres = torch.full((self.corres_pool_size,), -1, dtype=torch.int64, device=self.device)

    # ------------------------------------------------------------------ helpers
    def clear(self):
        self.keys.fill_(-1)

    def _cells(self, points):
        return torch.round((points - self.pc_range[: self.ndim]) / self.voxel_size).long() + 1

    def _set_geometry(self, voxel_size, pc_range, ndim):
        self.ndim = int(ndim)
        self.voxel_size = voxel_size.to(self.device).clone()
        self.pc_range = pc_range.to(self.device).clone()
        self.dims = ((self.pc_range[ndim:] - self.pc_range[:ndim]) / self.voxel_size).long() + 3
        if self.values.shape[1] != ndim:
            self.values = torch.zeros(self.size, ndim, dtype=self.values.dtype, device=self.device)
        if self.qmin.shape[0] != ndim:
            self.qmin = torch.full((ndim,), -1, dtype=torch.int32, device=self.device)
            self.qmax = torch.full((ndim,), 1, dtype=torch.int32, device=self.device)

    def _need_state(self):
        if not hasattr(self, "voxel_size"):
            raise ValueError("make sure you call hash_into_gpu first")

    def _window(self, temporal_offset):
        self.qmin[-1] = int(temporal_offset)
        self.qmax[-1] = int(temporal_offset)

    def _pool(self, n):
        """corres[:n] = -1, the pool grown to 2 n when it is too small"""
        if n > self.corres_pool_size:
            self.corres_pool_size = 2 * n
            self.corres = torch.full((self.corres_pool_size,), -1, dtype=torch.int64, device=self.device)
        else:
            self.corres[:n] = -1
        return self.corres[:n]

    def _insert(self, ref_points):
        ref_points = ref_points.to(self.device).to(self.values.dtype).contiguous()
        self.clear()
        self.num_points = ref_points.shape[0]
        ops.hash_insert_gpu(self.keys, self.values, self.reverse_indices, self.dims, self._cells(ref_points), ref_points)
        return ref_points

    # ------------------------------------------------------------------ the reference's methods
    @torch.no_grad()
    def hash_into_gpu(self, ref_points, voxel_size, pc_range):
        """Fill the table with ref_points [N, D]; the geometry stays for the *_step2 queries."""
        self._set_geometry(voxel_size, pc_range, ref_points.shape[-1])
        self._insert(ref_points)

    @torch.no_grad()
    def find_corres_step2(self, query_points, temporal_offset):
        """[2, E]: row 0 = indices of the queries that found a point, row 1 = the hashed point each found (the nearest
        one among the cells walked, torch_hash_kernel.cu:444-485)."""
        self._need_state()
        query_points = query_points.to(self.device).to(self.values.dtype).contiguous()
        self._window(temporal_offset)
        corres = self._pool(query_points.shape[0])
        ops.correspondence(self.keys, self.values, self.reverse_indices, self.dims, self._cells(query_points),
                           query_points, self.qmin, self.qmax, corres)
        hit = (corres != -1).nonzero(as_tuple=True)[0]
        return torch.stack([hit, corres[hit]], dim=0)

    @torch.no_grad()
    def find_corres(self, ref_points, query_points, voxel_size, pc_range, temporal_offset):
        """hash_into_gpu(ref) + find_corres_step2(query) in one call; both clouds are clamped into pc_range first
        (as the reference does)."""
        ndim = ref_points.shape[-1]
        self._set_geometry(voxel_size, pc_range, ndim)
        lo, hi = self.pc_range[:ndim], self.pc_range[ndim:]
        ref = torch.max(torch.min(ref_points.to(self.device).to(self.values.dtype), hi), lo)
        qry = torch.max(torch.min(query_points.to(self.device).to(self.values.dtype), hi), lo)
        self._insert(ref)
        return self.find_corres_step2(qry, temporal_offset)

    @torch.no_grad()
    def points_in_radius_step2(self, query_points, temporal_offset, radius=0.5):
        """Indices of the hashed points closer than `radius` to some query (torch_hash_kernel.cu:563-605)."""
        self._need_state()
        query_points = query_points.to(self.device).to(self.values.dtype).contiguous()
        self._window(temporal_offset)
        if getattr(self, "visited", None) is None or self.visited.shape[0] != self.num_points:
            self.visited = torch.zeros(self.num_points, dtype=torch.int64, device=self.device)
        else:
            self.visited.zero_()
        ops.points_in_radius_gpu(self.keys, self.values, self.reverse_indices, self.dims, self._cells(query_points),
                                 query_points, self.qmin, self.qmax, float(radius), self.visited)
        return self.visited.nonzero(as_tuple=True)[0]

    @torch.no_grad()
    def voxel_graph_step2(self, query_points, temporal_offset, radius=0.5, max_num_neighbors=32):
        """[2, E]: (query index, hashed point index) of up to max_num_neighbors points within `radius` of every query."""
        self._need_state()
        query_points = query_points.to(self.device).to(self.values.dtype).contiguous()
        self._window(temporal_offset)
        r = query_points.new_zeros(query_points.shape[0]) + float(radius)
        edges = ops.radius_graph_gpu(self.keys, self.values, self.reverse_indices, self.dims, self._cells(query_points),
                                     query_points, self.qmin, self.qmax, r, int(max_num_neighbors), False)
        return torch.stack([edges[:, 1], edges[:, 0]], dim=0)

    @torch.no_grad()
    def voxel_graph(self, ref_points, query_points, voxel_size, pc_range, temporal_offset, radius=0.5,
                    max_num_neighbors=32):
        assert ref_points.shape[0] + query_points.shape[0] < self.size * 2
        self._set_geometry(voxel_size, pc_range, ref_points.shape[-1])
        self._insert(ref_points)
        return self.voxel_graph_step2(query_points, temporal_offset, radius, max_num_neighbors)
