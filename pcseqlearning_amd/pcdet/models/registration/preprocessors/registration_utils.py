"""Host side of pcdet/models/registration/preprocessors/registration_utils.py: the same function
names and argument meaning, computed by the HIP kernels (ordered segment reductions, medians, the
fused ICP). Segment reductions here sum in ascending point index, so results are reproducible
run to run (the reference's scatter atomics / unstable argsort are not)."""
import torch

from ..... import ops
from ....utils.common_utils import EasyDict, filter_dict


class IndexSegments:
    """Sorted-segment view of an index tensor (component ids, voxel ids), built once and shared by
    every reduction over that index (the reference re-sorts or re-scatters per call)."""

    def __init__(self, index, max_index=None):
        self.index = index
        self.max_index = max_index
        self.seg = ops.unique_i64(index.long().contiguous(), want_inv=False)
        self.ids = self.seg.uniq

    def scatter_dense(self, values, max_index, fill=0):
        out = values.new_full((max_index,) + tuple(values.shape[1:]), fill)
        if self.seg.num:
            out[self.ids] = values
        return out


def _segments(index, max_index):
    return index if isinstance(index, IndexSegments) else IndexSegments(index, max_index)


def robust_mean(data, index, max_index):
    """sum / count where the count is positive, 0 elsewhere (registration_utils.py:12-21)."""
    sg = _segments(index, max_index)
    flat = data.reshape(data.shape[0], -1)
    if data.dtype == torch.float64:
        # rare (fp64 inputs): carried as two fp32 halves would lose bits -> plain ordered loop on
        # the device via index_add over the sorted order (deterministic for sorted input)
        order = sg.seg.order.long()
        seg_of = torch.repeat_interleave(torch.arange(sg.seg.num, device=data.device),
                                         (sg.seg.seg_start[1:] - sg.seg.seg_start[:-1]).long())
        sums = flat.new_zeros(sg.seg.num, flat.shape[1]).index_add_(0, seg_of, flat[order])
        cnt = (sg.seg.seg_start[1:] - sg.seg.seg_start[:-1]).to(data.dtype)[:, None]
        vals = sums / cnt
    else:
        vals = ops.segment_reduce(flat, sg.seg, mean=True, wave=True).to(data.dtype)
    return sg.scatter_dense(vals, max_index).reshape((max_index,) + tuple(data.shape[1:]))


def efficient_robust_mean(data, index, max_index, sort_index=None):
    return robust_mean(data, index, max_index)


def efficient_robust_sum(data, index, max_index, sort_index=None, block=False):
    """block=True: one workgroup per segment (few, very large segments)."""
    sg = _segments(index, max_index)
    flat = data.reshape(data.shape[0], -1)
    vals = ops.segment_reduce(flat, sg.seg, mean=False, wave=True, block=block).to(data.dtype)
    return sg.scatter_dense(vals, max_index).reshape((max_index,) + tuple(data.shape[1:]))


def segment_max(data, index, max_index):
    """scatter(data, index, reduce='max', dim_size=max_index); empty segments -> 0 (torch_scatter)."""
    sg = _segments(index, max_index)
    vals = ops.segment_minmax(data, sg.seg, True)
    return sg.scatter_dense(vals, max_index)


def segment_min(data, index, max_index):
    sg = _segments(index, max_index)
    vals = ops.segment_minmax(data, sg.seg, False)
    return sg.scatter_dense(vals, max_index)


def truncated_robust_mean(data, index, max_index, trunc_dist=0.3):
    sg = _segments(index, max_index)
    mean = robust_mean(data, sg, max_index)
    idx = sg.index.long()
    clamped = torch.minimum(torch.maximum(data, mean[idx] - trunc_dist), mean[idx] + trunc_dist)
    return robust_mean(clamped, sg, max_index)


def robust_median(data, index, max_index):
    """Per-index median of integers: sorted element offset + deg // 2; empty -> -1e10
    (registration_utils.py:60-81)."""
    sg = _segments(index, max_index)
    vals = ops.segment_median(data.long(), sg.seg).to(data.dtype)
    return sg.scatter_dense(vals, max_index, fill=int(-1e10))


def register_to_next_frame(graph, moving, ref, num_components, angle_regularizer=10, max_iter=20,
                           stopping_delta=5e-2):
    """Same contract as the reference (registration_utils.py:83-206): `moving` / `ref` dicts with
    fxyz, frame, stationary (and moving.component); returns (moving, T [C,4,4] f64,
    l1_component_error [C] f64, comp_edge_ratio [C] f32). The whole ICP loop runs in ONE call of
    the fused HIP path; `graph` supplies radius / neighbour settings and keeps its state."""
    temp_radius = graph.radius
    frame_offset = int((ref.frame[0] - moving.frame[0]).long().item())
    component_deg = torch.bincount(moving.component.long(), minlength=num_components)
    ns_mask = ~moving.stationary
    mov_fxyz = moving.fxyz[ns_mask].contiguous()
    mov_comp = moving.component[ns_mask]
    ref_ns = ref.fxyz[~ref.stationary]
    cell = 0.0  # fused ICP picks radius / 4 (two-level grid, pcs_grid2.hpp)
    T, l1_component_error, comp_edge_ratio, _ = ops.register_to_next_frame(
        mov_fxyz, mov_comp, ref_ns, ref.fxyz, component_deg, num_components, float(temp_radius),
        frame_offset, float(angle_regularizer), int(max_iter), float(stopping_delta), float(cell))
    # the reference leaves the graph looking at the forward frame offset when it returns
    graph.radius = temp_radius
    graph.qmin[0] = frame_offset
    graph.qmax[0] = frame_offset
    moving.fxyz[ns_mask, 1:] = mov_fxyz[:, 1:]
    return moving, T, l1_component_error, comp_edge_ratio
