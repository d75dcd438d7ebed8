"""RadiusGraph / connected_components of pcdet/models/model_utils/graph_utils.py (:36-53,
:131-212) on the HIP radius-graph engine. Same registry (`GRAPHS`, `build_graph`), same mutable
`.radius / .qmin / .qmax` that callers poke (registration_utils.py:131-137,
cluster_tracking.py:356-358), same `(e_ref, e_query, None)` return."""
import torch
from torch import nn

from .... import ops
from ....lib import PcsError


def connected_components(edges, num_nodes=None, host_count=True):
    """edges [2,E] -> (num_components, component [N]); label = rank of the component by its
    smallest node id, which is what scipy.sparse.csgraph returns in the reference. host_count=False: the count is
    a one-element device tensor (no host synchronisation)."""
    if num_nodes is None:
        num_nodes = int(edges.max().item()) + 1
    num, comp = ops.connected_components(edges[0].contiguous(), edges[1].contiguous(), num_nodes, host_count)
    return num, comp.to(edges.dtype)


class RadiusGraph(nn.Module):
    def __init__(self, runtime_cfg=None, model_cfg=None):
        super().__init__()
        self.model_cfg = model_cfg
        self.radius = model_cfg.get("RADIUS", None)
        if model_cfg.get("DYNAMIC_RADIUS", False):
            raise NotImplementedError("DYNAMIC_RADIUS (torch_cluster.knn) is not on the hot path")
        if "EDGE_DIRECTION" in model_cfg:
            raise NotImplementedError("EDGE_DIRECTION filtering is not on the hot path")
        self.max_num_neighbors = model_cfg.get("MAX_NUM_NEIGHBORS", 32)
        self.sort_by_dist = model_cfg.get("SORT_BY_DIST", False)
        self.util_ratio = 0.5
        # host-resident on purpose: the engine takes the frame-offset range as launch arguments, so
        # the callers' `graph.qmin[0] = df` writes cost no device round trip
        self.qmin = torch.tensor([0, -1, -1, -1]).int()
        self.qmax = torch.tensor([0, 1, 1, 1]).int()
        self.relative_key = model_cfg.get("RELATIVE_KEY", "bxyz")
        # fine-cell size of the expanding-ring search; None -> heuristic from the radius
        self.cell_size = model_cfg.get("CELL_SIZE", None)

    def build_graph(self, ref, query):
        ref_bxyz = ref[self.relative_key]
        query_bxyz = query[self.relative_key]
        assert ref_bxyz.shape[-1] == 4
        qmin, qmax = self.qmin.tolist(), self.qmax.tolist()
        if qmin[1:] != [-1, -1, -1] or qmax[1:] != [1, 1, 1]:
            raise PcsError("RadiusGraph: only the default spatial query range (-1..1) is supported "
                           "by the engine; use ops.hash_insert_gpu/radius_graph_gpu for others")
        radius = float(torch.tensor(float(self.radius), dtype=torch.float32))  # fp32 as in build_graph
        cell = self.cell_size
        if cell is None:
            # spatial radius without the frame-offset term
            df = max(abs(qmin[0]), abs(qmax[0]))
            r_xyz = max(radius * radius - df * df, 1e-6) ** 0.5
            # 1-NN: the two-level grid wants (largest admissible spatial distance) / 4
            cell = r_xyz / 2 if self.max_num_neighbors > 1 else (r_xyz * r_xyz + 3e-5 * radius * radius + 3e-6) ** 0.5 / 4 * 1.0005
        e_ref, e_query = ops.radius_graph(ref_bxyz, query_bxyz, radius, self.max_num_neighbors,
                                          self.sort_by_dist, qmin[0], qmax[0], cell_size=cell)
        return e_ref, e_query, None

    def forward(self, ref, query):
        return self.build_graph(ref, query)

    def extra_repr(self):
        return f"radius={self.radius}, max_ngbrs={self.max_num_neighbors}, sort={self.sort_by_dist}"


GRAPHS = {
    "RadiusGraph": RadiusGraph,
}


def build_graph(graph_cfg, runtime_cfg=None):
    graph = GRAPHS[graph_cfg["TYPE"]]
    return graph(model_cfg=graph_cfg, runtime_cfg=runtime_cfg)
