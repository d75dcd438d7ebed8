"""VoxelAggregation (pcdet/ops/voxel/voxel_modules.py:7-126) on the HIP path.

Same constructor `(model_cfg, runtime_cfg)`, same `forward(point_wise_mean_dict,
point_wise_median_dict=None) -> (voxel_wise_dict, point_wise_dict, num_voxels,
out_of_boundary_mask)` with the reference's keys. The 4-D (batch, x, y, z) cells are quantised with
the reference's fp32 `floor((p - min) / size)`, linearised with dim 0 most significant
(:58-60), made unique by the sorted-unique kernel (`pcs_unique_i64`: voxel index = rank of the
linear id, as torch.unique gives), attributes averaged by the ordered segment sums
(`pcs_segment_reduce_f32`), integer attributes reduced by the per-voxel upper median
(`pcs_segment_median_i64`; voxel_modules.py:81-93). Integer outputs equal the reference's bit for bit;
the means differ from torch_scatter's unordered fp32 sums by rounding only.
"""
import torch
from torch import nn

from .... import ops
from ...utils import common_utils


class VoxelAggregation(nn.Module):
    def __init__(self, model_cfg, runtime_cfg):
        super().__init__()
        self.model_cfg = model_cfg
        voxel_size = [1] + list(model_cfg.get("VOXEL_SIZE", None))
        self.register_buffer("voxel_size", torch.tensor(voxel_size).float(), persistent=False)
        point_cloud_range = model_cfg.get("POINT_CLOUD_RANGE", None)
        if point_cloud_range is not None:
            point_cloud_range = [0] + list(point_cloud_range[:3]) + [1] + list(point_cloud_range[3:])
            self.register_buffer("point_cloud_range", torch.tensor(point_cloud_range).float(), persistent=False)
        else:
            self.point_cloud_range = None

    def forward(self, point_wise_mean_dict, point_wise_median_dict=None):
        point_bxyz = point_wise_mean_dict["point_bxyz"]
        batch_size = point_bxyz[:, 0].max().long().item() + 1
        if self.point_cloud_range is None:
            pc_range_min = point_bxyz.min(0)[0]
            voxel_coords = torch.floor((point_bxyz - pc_range_min) / self.voxel_size).long()
            voxel_coords = voxel_coords - voxel_coords.min(0)[0]
            dims = voxel_coords.max(0)[0] + 1
            out_of_boundary_mask = torch.zeros(voxel_coords.shape[0], dtype=torch.bool, device=voxel_coords.device)
        else:
            pc_range_min = self.point_cloud_range[:4].clone()
            pc_range_max = self.point_cloud_range[4:].clone()
            pc_range_min[0] = 0
            pc_range_max[0] = batch_size - 1e-5
            voxel_coords = torch.floor((point_bxyz - pc_range_min) / self.voxel_size).long()
            dims = torch.ceil((pc_range_max - pc_range_min) / self.voxel_size).long() + 1
            out_of_boundary_mask = (voxel_coords >= dims)[:, 1:4].any(-1) | (voxel_coords < 0)[:, 1:4].any(-1)
            keep = ~out_of_boundary_mask
            voxel_coords = voxel_coords[keep]
            point_bxyz = point_bxyz[keep]
            point_wise_mean_dict = common_utils.filter_dict(point_wise_mean_dict, keep)
        point_coords = voxel_coords
        assert (voxel_coords >= 0).all(), f"VoxelGraph: min={voxel_coords.min(0)[0]}"
        assert (voxel_coords < dims).all(), f"VoxelGraph: min={voxel_coords.max(0)[0]}, dims={dims}"

        dims_h = [int(v) for v in dims.tolist()]
        merge_coords = torch.zeros_like(voxel_coords[:, 0])
        for i in range(4):
            merge_coords = merge_coords * dims_h[i] + voxel_coords[:, i]
        seg = ops.unique_i64(merge_coords.contiguous())  # sorted unique + voxel index of every point
        voxel_index, num_voxels = seg.inv, seg.num
        unq = seg.uniq.clone()
        unq_coords = unq.new_zeros(num_voxels, 4)
        for i in range(3, -1, -1):
            unq_coords[:, i] = unq % dims_h[i]
            unq = torch.div(unq, dims_h[i], rounding_mode="trunc").long()
        voxel_center = (unq_coords * self.voxel_size + pc_range_min)[:, 1:4]
        voxel_center += self.voxel_size[1:4] / 2

        voxel_wise_mean_dict = {}
        for key, val in point_wise_mean_dict.items():
            voxel_key = "voxel_" + key.split("point_")[-1]
            red = ops.segment_reduce(val, seg, mean=True)
            # torch_scatter's integer mean is a floor division; the segment kernel sums in fp32
            voxel_wise_mean_dict[voxel_key] = red if val.is_floating_point() else torch.floor(red).to(val.dtype)
        voxel_bxyz = voxel_wise_mean_dict["voxel_bxyz"]

        ret_median_dict = {}
        if point_wise_median_dict is not None:
            for key, val in point_wise_median_dict.items():
                val = val[~out_of_boundary_mask]
                max_val, min_val = val.max(), val.min()
                if int(min_val.item()) >= 0:
                    # the reference sorts (val - min) + voxel_index * max and reads element offset + deg // 2:
                    # with min >= 0 the voxels' key ranges do not interleave, i.e. the per-voxel upper median
                    # of (val - min) -- it does not add min back, and neither does this
                    ret_median_dict["voxel_" + key] = (ops.segment_median(val, seg) - min_val).to(val.dtype)
                else:
                    # negative values make the reference's keys of neighbouring voxels interleave; its
                    # literal arithmetic (voxel_modules.py:84-93), on the device
                    degree = (seg.seg_start[1:] - seg.seg_start[:-1]).long()
                    offset = degree.cumsum(dim=0) - degree
                    median_offset = offset + torch.div(degree, 2, rounding_mode="floor")
                    tval = (val - min_val) + voxel_index * max_val
                    sorted_vals = torch.sort(tval)[0]
                    ret_median_dict["voxel_" + key] = \
                        sorted_vals[median_offset] - torch.arange(num_voxels).to(val) * max_val

        point_xyz = point_bxyz[:, 1:4].contiguous()
        voxel_xyz = voxel_bxyz[:, 1:4].contiguous()
        voxel_wise_dict = dict(
            voxel_batch_index=voxel_bxyz[:, 0].round().long(),
            voxel_xyz=voxel_xyz,
            voxel_center=voxel_center,
            voxel_bcenter=torch.cat([voxel_bxyz[:, 0:1], voxel_center], dim=-1),
            voxel_bcoords=unq_coords,
        )
        voxel_wise_dict.update(voxel_wise_mean_dict)
        voxel_wise_dict.update(ret_median_dict)
        # containment invariant of the reference (voxel_modules.py:113-117)
        diff_min = point_xyz - voxel_center[voxel_index] + self.voxel_size[1:4] / 2
        diff_max = point_xyz - voxel_center[voxel_index] - self.voxel_size[1:4] / 2
        assert (diff_min > -1e-4).all() and (diff_max < 1e-4).all(), "VoxelAggregation: a point lies outside its voxel"

        point_wise_dict = dict(voxel_index=voxel_index, point_bcoords=point_coords)
        return voxel_wise_dict, point_wise_dict, num_voxels, out_of_boundary_mask

    def extra_repr(self):
        return f"voxel_size={list(self.voxel_size.detach().cpu().numpy())[1:]}"
