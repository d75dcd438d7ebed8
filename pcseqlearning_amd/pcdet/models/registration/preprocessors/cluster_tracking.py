"""ClusterTracking (pcdet/models/registration/preprocessors/cluster_tracking.py) on the HIP path.

Same registry name, constructor `(model_cfg, runtime_cfg)`, `forward(seq_dict) -> seq_dict`, same
on-disk outputs (`<DIR>/<seq>/<frame:03d>_<key>.pth`, `<DIR>/<seq>/all.pth`). The control flow
restates SURVEY.md A.6 / cluster_tracking.py:430-787; every point-level computation goes through
pcseqlearning_amd.ops (grid sampling, medians, fused ICP, radius graph, ordered segment reductions,
smooth_velo, points_in_boxes). The polyscope visualiser hooks of the reference are not mirrored.
"""
import os
import time

import numpy as np
import torch
import torch.nn as nn

from ..... import ops
from .....lib import PcsError
from ....utils.common_utils import EasyDict, filter_dict, run_deferred, thread_stream
from ....utils.timer import Timer
from ...model_utils import graph_utils
from ...model_utils.grid_sampling import GridSampling3D
from .eval_utils import component_box_iou_frames, count_into, scatter_max_
from .registration_utils import (IndexSegments, efficient_robust_sum, register_to_next_frame,
                                 robust_mean, robust_median, segment_max, segment_min)


def run_ids(counts, device):
    """torch.repeat_interleave(arange(len(counts)), counts) for host counts: [sum(counts)] int64, run i holds i. By
    bucketing a ramp against the runs' end offsets: repeat_interleave's own kernel takes 190 us per million outputs
    (four of those per sequence in the extraction), a ramp and a bucketize 10 us each."""
    ends, total = [], 0
    for c in counts:
        total += int(c)
        ends.append(total)
    if total == 0:
        return torch.zeros(0, dtype=torch.int64, device=device)
    return torch.bucketize(torch.arange(total, device=device), torch.tensor(ends, device=device), right=True)


HOST_FALLBACKS = 0  # sequences that left the native tracker for the op-by-op host loop (bench.py reports it)
_FALLBACK_LOCK = __import__("threading").Lock()


def note_host_fallback(sequence_id, err):
    global HOST_FALLBACKS
    import sys
    with _FALLBACK_LOCK:
        HOST_FALLBACKS += 1
    print(f"[pcs] ClusterTracking: sequence {sequence_id}: native tracker returned {err}; "
          f"tracking this sequence through the op-by-op host loop (slow path)", file=sys.stderr, flush=True)


def sample_frame(grid_sampler, frame):
    """Voxel-subsample a frame: mean of fxyz / stationary, median of component / frame ids
    (cluster_tracking.py:39-51)."""
    seg = grid_sampler.segments(frame.fxyz)
    new_frame = EasyDict()
    new_frame["fxyz"] = ops.segment_reduce(frame.fxyz, seg, mean=True)
    new_frame["stationary"] = ops.segment_reduce(frame.stationary.float(), seg, mean=True) > 0.5
    for key in ("component", "frame"):
        new_frame[key] = ops.segment_median(frame[key].long().reshape(-1), seg)
    return new_frame


def rotation_angle(R):
    diag = R.reshape(*R.shape[:-2], 9)[..., [0, 4, 8]]
    return ((diag.sum(-1) - 1) / 2).clamp(min=-1, max=1).arccos()


def pose_inv(T):
    R, t = T[..., :3, :3], T[..., :3, 3]
    Tinv = T.clone()
    Tinv[..., :3, :3] = R.transpose(-1, -2)
    Tinv[..., :3, 3:] = -R.transpose(-1, -2) @ t.unsqueeze(-1)
    return Tinv


def dist_compensate(comp_deg):
    """Size-dependent slack on the registration-error stop rule (cluster_tracking.py:80-87)."""
    thresholds = [0, 10, 40, 100, 200, 400, 10000000]
    slack = [1, 0.5, 0.3, 0.2, 0.1, 0.0]
    out = torch.zeros_like(comp_deg).float()
    for lo, hi, val in zip(thresholds[:-1], thresholds[1:], slack):
        out[(comp_deg >= lo) & (comp_deg < hi)] = val
    return out


def _component_stats(frame_points, sg=None):
    xyz = frame_points.fxyz[:, 1:]
    num_components = int(frame_points.component.max().long().item()) + 1
    if sg is None:
        sg = IndexSegments(frame_points.component, num_components)
    deg = efficient_robust_sum(torch.ones_like(xyz[:, 0]), sg, num_components)
    valid = (deg > 0.5).reshape(-1)
    center = efficient_robust_sum(xyz, sg, num_components)
    # (a select, not a masked assignment: the same quotients where the mask holds, no host synchronisation)
    center = torch.where(valid[:, None], center / deg.reshape(-1, 1), center)
    return num_components, sg, deg, valid, center


def component_center(frame_points, sg=None):
    return _component_stats(frame_points, sg)[4]


def component_diameter(frame_points, sg=None):
    """2 x the largest distance of a component's points to its centre (cluster_tracking.py:89-107)."""
    num_components, sg, _, _, center = _component_stats(frame_points, sg)
    dist = (frame_points.fxyz[:, 1:] - center[frame_points.component]).norm(p=2, dim=-1)
    return segment_max(dist, sg, num_components) * 2


def filter_components(frame_points, max_diameter=12.5, sg=None):
    num_components = int(frame_points.component.max().long().item()) + 1
    if sg is None:
        sg = IndexSegments(frame_points.component, num_components)
    deg = efficient_robust_sum(torch.ones_like(frame_points.fxyz[:, 0]), sg, num_components)
    valid = (deg > 0.5).reshape(-1)
    if max_diameter > 0:
        valid = valid & (component_diameter(frame_points, sg) < max_diameter)
    return valid


def smooth_velo(comp_velos, comp_center_diffs, frame_id, next_frame_id, weight0=1, weight=10,
                num_itr=300, stopping=1e-3):
    """cluster_tracking.py:162-199 as one persistent HIP kernel; in place, like the reference's
    Parameter (which aliases its input)."""
    return ops.smooth_velo(comp_velos, comp_center_diffs, frame_id, next_frame_id, float(weight0),
                           float(weight), num_itr, stopping)


_DEFERRED_POOL = []


def _deferred_pool():
    """Process-wide helper threads for the work parked beside the native tracker (persistent: the ops
    they call keep per-thread pinned pages)."""
    if not _DEFERRED_POOL:
        from concurrent.futures import ThreadPoolExecutor
        _DEFERRED_POOL.append(ThreadPoolExecutor(max_workers=16, thread_name_prefix="pcs-deferred"))
    return _DEFERRED_POOL[0]


class ClusterTracking(nn.Module):
    def __init__(self, model_cfg, runtime_cfg):
        super().__init__()
        self.model_cfg = model_cfg
        self.forward_dict = EasyDict()
        registration_cfg = model_cfg["REGISTRATION"]
        self.stopping_delta = registration_cfg["STOPPING_DELTA"]
        graph_cfg = registration_cfg["GRAPH"]
        self.radius_list = graph_cfg["RADIUS"]
        self.voxel_size_list = registration_cfg["VOXEL_SIZE"]
        for i, (radius, voxel_size) in enumerate(zip(self.radius_list, self.voxel_size_list)):
            cfg = dict(graph_cfg)
            cfg["RADIUS"] = radius
            graph = graph_utils.build_graph(cfg, {})
            # fine-cell size of the fused ICP's neighbour search: a couple of sampling voxels
            graph.icp_cell_size = 2.0 * float(max(voxel_size[:2]))
            self.add_module(f"registration_graph_{i}", graph)
            self.add_module(f"sampler_{i}", GridSampling3D(list(voxel_size)))
        self.nn_graph = graph_utils.build_graph(model_cfg["NN_GRAPH"], runtime_cfg=runtime_cfg)
        self.visualize = False  # the reference's polyscope hooks are out of scope
        params = model_cfg.get("TRACKING_PARAMS", {})
        self.reg_error_coeff = params.get("REGISTRATION_ERROR_COEFFICIENT", 0.13)
        self.track_interval = params.get("TRACK_INTERVAL", 10)
        self.angle_threshold = params.get("ANGLE_THRESHOLD", 45)
        self.min_move_frame = params.get("MIN_MOVE_FRAME", 6)
        self.component_keys = model_cfg["COMPONENT_KEYS"]
        self.verbose = bool((runtime_cfg or {}).get("verbose", False))
        self.num_registration_calls = 0
        # NATIVE_TRACKER: run track_frame's loop in pcs_track_frame (default) or op by op
        self.native = bool(model_cfg.get("NATIVE_TRACKER", True))
        self._track_params = None

    # ------------------------------------------------------------------ boxes
    def format_boxes(self, seq_dict, num_frames):
        return EasyDict(dict(
            attr=seq_dict["gt_box_attr"].reshape(-1, 7),
            cls_label=seq_dict["gt_box_cls_label"].reshape(-1),
            frame=seq_dict["gt_box_frame"].reshape(-1),
            trace_id=seq_dict["gt_box_track_label"].reshape(-1),
            velo=seq_dict["gt_box_velo"].reshape(-1),
            moving=seq_dict["moving"].reshape(-1),
        ))

    # ------------------------------------------------------------------ full-resolution harvest
    def sequence_views(self, all_points, seq_boxes):
        """What every tracking unit of a sequence needs of the full-resolution points and the GT boxes,
        built once per sequence: the points grouped by frame (stable, so the order within a frame is
        the reference's masked order), the row range of every frame, the boxes' row ranges per frame,
        and for every point the first box of its frame that contains it (its GT box)."""
        dev = all_points.fxyz.device
        v = EasyDict()
        fr = all_points.frame.reshape(-1).long()
        num_frames = int(fr.max().item()) + 1
        perm = torch.argsort(fr, stable=True)
        v.perm = perm
        v.frame = fr[perm]
        v.fxyz = all_points.fxyz[perm].contiguous()
        for key in ("full_segmentation_label", "full_instance_label"):
            v[key] = all_points[key][perm]
        v.frame_start = torch.searchsorted(v.frame, torch.arange(num_frames + 1, device=dev))
        v.frame_start_host = v.frame_start.tolist()
        assert bool((seq_boxes.frame[1:] >= seq_boxes.frame[:-1]).all()), "boxes must be grouped by frame"
        box_cnt = count_into(seq_boxes.frame.long(), num_frames)
        v.box_cnt = box_cnt
        v.box_off = torch.zeros(box_cnt.numel() + 1, dtype=torch.int64, device=dev)
        v.box_off[1:] = box_cnt.cumsum(0)
        v.box_off32 = v.box_off.int()
        v.bmax = int(box_cnt.max().item()) if box_cnt.numel() else 0
        v.num_frames = num_frames
        _, first = ops.points_in_frame_boxes(v.fxyz[:, 1:].contiguous(), v.frame, seq_boxes.attr,
                                             v.box_off32, v.bmax, want_member=False)
        v.gt_local = first.long()
        inb = v.gt_local >= 0
        # (rows outside every box carry weight 0; their index is clamped into the table: a frame without boxes
        #  behind the last box would point one past it)
        n_box = seq_boxes.attr.shape[0]
        v.gt_cnt_all = count_into((v.box_off[v.frame.clamp(max=box_cnt.numel() - 1)]
                                   + v.gt_local.clamp(min=0)).clamp(max=max(n_box - 1, 0)), n_box, inb)
        return v

    def extract_traces_and_update_boxes(self, all_points, extracted, seq_boxes, views=None):
        """Map tracked voxels back to the full-resolution points of each frame and update the
        per-box best IoU (cluster_tracking.py:287-428): `extract_traces_of_units` for one unit."""
        full = self.extract_traces_of_units(all_points, [extracted], seq_boxes, views)[0]
        return full, seq_boxes

    def extract_traces_of_units(self, all_points, extracted_list, seq_boxes, views=None):
        """cluster_tracking.py:287-428 for several tracking units at once (the reference runs it per
        unit and, inside, per frame). The units' tracked voxels are concatenated with their component
        ids offset, so that a (unit, component) pair is one id: all frames of all units go through ONE
        radius-graph call (edges never cross frames; a query belongs to one unit), the per-(frame,
        component) centres / diameters are segment reductions on a combined key, the GT bookkeeping is
        the histogram form of eval_utils, and best_iou -- a running maximum over units in the
        reference -- is one scatter-max. Returns the reference's `full_extracted` per unit, rows in the
        reference's order (frame ascending, then tracked voxel); seq_boxes.best_iou is updated in place.
        Units without any tracked voxel come back unchanged (the reference skips the call for them)."""
        if views is None:
            views = self.sequence_views(all_points, seq_boxes)
        out = list(extracted_list)
        todo = [i for i, ex in enumerate(extracted_list) if ex.fxyz.shape[0] > 0]
        if not todo:
            return out
        exs = [extracted_list[i] for i in todo]
        transforms = [ex.pop("transforms") for ex in exs]
        rows = [ex.fxyz.shape[0] for ex in exs]
        # one host read for all units: components per unit and the frame range
        stats = torch.stack([torch.stack([ex.component.max().long(), ex.fxyz[:, 0].min().round().long(),
                                          ex.fxyz[:, 0].max().round().long()]) for ex in exs]).tolist()
        fxyz = torch.cat([ex.fxyz for ex in exs]) if len(exs) > 1 else exs[0].fxyz
        comp_local = torch.cat([ex.component.long() for ex in exs]) if len(exs) > 1 else exs[0].component.long()
        moving = torch.cat([ex.moving for ex in exs])
        full = self._extract_units_core(views, seq_boxes, fxyz, comp_local, moving, rows, stats,
                                        [ex.component.dtype for ex in exs], transforms)
        for i, f in zip(todo, full):
            out[i] = f
        return out

    def _extract_units_core(self, v, seq_boxes, fxyz, comp_local, moving, rows, stats, dtypes, transforms, unit=None):
        """The body of extract_traces_of_units on the units' rows already concatenated: fxyz [n, 4], comp_local [n]
        int64 (each unit's own component ids), moving [n] bool, rows / stats / dtypes / transforms per unit
        (stats = (largest component id, first frame, last frame) of the unit's rows, host integers; a frame range wider
        than the rows' own gives the same results: frames without voxels have no edges and empty groups). `rows` (host
        counts) or `unit` ([n] int64, the unit of every row; rows grouped by unit) says whose rows are whose. What is computed per unit in the reference -- labels, original indices, frame-local indices -- is gathered
        ONCE for all units; the per-unit results are views of those arrays (a dozen small launches per unit less)."""
        n_units = len(stats)
        ncomp = [s[0] + 1 for s in stats]
        f_lo, f_hi = min(s[1] for s in stats), max(s[2] for s in stats)
        coff = [0]
        for c in ncomp:
            coff.append(coff[-1] + c)
        num_components = coff[-1]
        dev = fxyz.device
        if unit is None:
            unit = run_ids(rows, dev)
        coff_t = torch.tensor(coff[:-1], device=dev)
        comp = comp_local + coff_t[unit] if n_units > 1 else comp_local
        component_hit = comp.new_zeros(num_components)
        ext_frame = fxyz[:, 0].round().long()
        # 1-NN of every full-resolution point among the tracked voxels OF ONE UNIT (the graph's "ref" set
        # is the voxels, its queries the full-resolution points of the unit's frame range): the units
        # are kept apart by shifting the frame coordinate of unit u by u * stride on both sides
        stride = v.num_frames + 2
        q_lo = [v.frame_start_host[s[1]] for s in stats]
        q_hi = [v.frame_start_host[s[2] + 1] for s in stats]
        q_off = [0]
        for a, b in zip(q_lo, q_hi):
            q_off.append(q_off[-1] + (b - a))
        if n_units == 1:
            vox, queries = fxyz, v.fxyz[q_lo[0]:q_hi[0]]
        else:
            vox = fxyz.clone()
            vox[:, 0] += (unit * stride).to(vox.dtype)
            queries = torch.cat([v.fxyz[a:b] for a, b in zip(q_lo, q_hi)])
            q_unit = run_ids([b - a for a, b in zip(q_lo, q_hi)], dev)
            queries[:, 0] += (q_unit * stride).to(queries.dtype)

        self.nn_graph.radius *= 1.732
        e_ext, e_ref, _ = self.nn_graph(EasyDict(fxyz=vox), EasyDict(fxyz=queries))
        self.nn_graph.radius /= 1.732
        # row in the frame-grouped full-resolution arrays
        to_row = torch.tensor([a - o for a, o in zip(q_lo, q_off[:-1])], device=dev)
        g_ref = e_ref + to_row[unit[e_ext]]

        # centre and xy-diameter of every (unit component, frame) of the tracked voxels. ONE sorted-segment structure,
        # component-major: the groups of a component are its frames, so the first and the last frame of every component
        # (component_size, a second sort by component and two segment reductions in the reference's form) are the
        # first and the last occupied column of the [component, frame] table of groups
        num_f = f_hi - f_lo + 1
        gkey = comp * num_f + (ext_frame - f_lo)
        num_groups = num_f * num_components
        gsg = IndexSegments(gkey, num_groups)
        occupied = torch.zeros(num_groups, dtype=torch.bool, device=dev)
        occupied[gsg.ids] = True
        occupied = occupied.reshape(num_components, num_f)
        col = torch.arange(num_f, device=dev)
        first_f = torch.where(occupied, col, num_f).amin(1)
        last_f = torch.where(occupied, col, -1).amax(1)
        # (a component id without rows: first = last = frame 0 in the reference's scatter with fill 0, size 1)
        component_size = torch.where(last_f >= 0, last_f - first_f + 1, torch.ones_like(last_f))
        center = robust_mean(fxyz[:, 1:3], gsg, num_groups)
        spread = (fxyz[:, 1:3] - center[gkey]).norm(p=2, dim=-1)
        diameter = segment_max(spread, gsg, num_groups)
        ekey = gkey[e_ext]
        ref_xyz = v.fxyz[g_ref]
        dz = fxyz[e_ext, -1] - ref_xyz[:, -1]
        keep = dz < 0.5
        dist = (ref_xyz[:, 1:3] - center[ekey]).norm(p=2, dim=-1)
        keep &= dist < diameter[ekey] + 0.05
        keep &= dz > -0.05
        kidx = keep.nonzero(as_tuple=True)[0]
        # rows in the reference's order: unit by unit, frame by frame ascending, within a frame by full-resolution
        # point. The edges ARRIVE in that order -- they are grouped by ascending query, the queries are the units'
        # frame-grouped points one unit after the other, and an edge's voxel has its query's unit and frame -- so the
        # stable sort by (unit, frame) that used to follow was the identity (a radix sort of a million keys)
        e_ext, g_ref = e_ext[kidx], g_ref[kidx]
        out_frame = v.frame[g_ref]
        out_comp = comp[e_ext]
        out_unit = unit[e_ext]
        full_fxyz = ref_xyz[kidx]
        # GT bookkeeping on the frames that have boxes
        if v.bmax > 0 and full_fxyz.shape[0] > 0:
            sel = (v.box_cnt[out_frame] > 0).nonzero(as_tuple=True)[0]
            if sel.numel():
                s_fr = out_frame[sel]
                # (the hits of every (frame, component) group per box, summed inside the box test: no [n, bmax] matrix)
                sel_xyz = full_fxyz[sel, 1:].contiguous()
                gframe, gcomp, _, has_box, box, iou = component_box_iou_frames(
                    out_comp[sel], s_fr,
                    (v.bmax, lambda inv, G: ops.frame_box_counts(sel_xyz, s_fr, seq_boxes.attr, v.box_off32, v.bmax, inv, G)),
                    v.gt_local[g_ref[sel]], v.box_off, v.gt_cnt_all, span=num_components)
                iou32 = iou.to(seq_boxes.best_iou.dtype)
                hit = has_box & (iou32 > 0.7)
                # (weights and selects instead of masked indexing: adding 0 / a maximum with 0 changes nothing, the
                #  IoUs are >= 0, and no mask has to be compacted on the host's clock)
                component_hit.index_add_(0, gcomp, hit.to(component_hit.dtype))
                n_box = seq_boxes.best_iou.shape[0]
                scatter_max_(seq_boxes.best_iou, (v.box_off[gframe] + box).clamp(max=max(n_box - 1, 0)),
                             torch.where(has_box, iou32, torch.zeros_like(iou32)))
        # (the rows are sorted by unit: the units' row ranges are the places where the unit number changes)
        bounds = torch.searchsorted(out_unit, torch.arange(n_units + 1, device=dev)).tolist()
        counts = [bounds[u + 1] - bounds[u] for u in range(n_units)]
        # everything that is a gather or an elementwise step per unit in the reference, once for all units
        same_dtype = all(d == dtypes[0] for d in dtypes)
        comp_rows = out_comp - coff_t[out_unit] if n_units > 1 else out_comp
        frame_rows = g_ref - v.frame_start[out_frame]
        if same_dtype:
            comp_rows, frame_rows = comp_rows.to(dtypes[0]), frame_rows.to(dtypes[0])
            hit_all = component_hit.to(dtypes[0])
        seg_rows = v.full_segmentation_label[g_ref]
        inst_rows = v.full_instance_label[g_ref]
        orig_rows = v.perm[g_ref].reshape(-1, 1)
        moving_rows = moving[e_ext]
        res = []
        start = 0
        for u in range(n_units):
            sl = slice(start, start + counts[u])
            start += counts[u]
            full = EasyDict(dict(
                fxyz=full_fxyz[sl],
                component=comp_rows[sl] if same_dtype else comp_rows[sl].to(dtypes[u]),
                segmentation_label=seg_rows[sl],
                instance_label=inst_rows[sl],
                original_indices=orig_rows[sl],
                frame_indices=frame_rows[sl] if same_dtype else frame_rows[sl].to(dtypes[u]),
                moving=moving_rows[sl],
            ))
            hit_u = hit_all[coff[u]:coff[u + 1]] if same_dtype else component_hit[coff[u]:coff[u + 1]].to(dtypes[u])
            full.component_hit = hit_u
            full.component_size = component_size[coff[u]:coff[u + 1]]
            full.transforms = transforms[u]
            res.append(full)
        return res

    # ------------------------------------------------------------------ tracking of one anchor
    def track_frame(self, seq_points, frame, seq_boxes):
        """Track the components of anchor frame `frame` through `seq_points` within
        +-TRACK_INTERVAL frames (cluster_tracking.py:430-787).

        When the sequence rows are grouped by frame (`seq_points.frame_offsets`, set by forward)
        the whole loop runs in one native call (pcs_track_frame, GIL released, one HIP stream);
        `track_frame_host` is the same loop op by op on the same kernels (bit-identical ids) and
        is the fallback when the rows are not grouped or a search grid overflows."""
        offsets = seq_points.get("frame_offsets", None) if self.native else None
        if offsets is not None and seq_points.fxyz.is_cuda:
            try:
                return self._track_frame_native(seq_points, frame, offsets)
            except PcsError as err:
                if "PCS_ERR_RANGE" not in str(err):
                    raise
        return self.track_frame_host(seq_points, frame, seq_boxes)

    @staticmethod
    def frame_offsets_of(frame_ids):
        """(offsets, perm): host list [num_frames + 1] of row ranges of the rows grouped by
        ascending frame id, and the stable permutation that groups them (None if they already are)."""
        fr = frame_ids.reshape(-1).long()
        if fr.numel() == 0:
            return None, None
        perm = None
        if bool((fr[1:] < fr[:-1]).any()):
            perm = torch.argsort(fr, stable=True)
            fr = fr[perm]
        num_frames = int(fr[-1].item()) + 1
        bounds = torch.searchsorted(fr, torch.arange(num_frames + 1, device=fr.device))
        return bounds.tolist(), perm

    def _native_params(self):
        if self._track_params is None:
            self._track_params = ops.track_params(
                self.radius_list, self.voxel_size_list, self.stopping_delta,
                self.model_cfg["ANGLE_REGULARIZER"], self.nn_graph.radius, self.reg_error_coeff,
                self.angle_threshold, self.min_move_frame, max_iter=80)
        return self._track_params

    def _native_window(self, offsets, frame_id):
        present = [f for f in range(len(offsets) - 1) if offsets[f + 1] > offsets[f]]
        return (max(present[0], frame_id - self.track_interval),
                min(present[-1], frame_id + self.track_interval))

    def _extracted_from_native(self, seq_points, frame_id, out, perm, comp_dtype):
        """The reference's `extracted` dictionary from the native tracker's outputs."""
        idx, comp, moving = out["ex_index"], out["ex_component"].to(comp_dtype), out["moving"]
        if perm is not None:
            idx = perm[idx]
        extracted = EasyDict(dict(
            fxyz=seq_points.fxyz[idx],
            component=comp,
            segmentation_label=seq_points.segmentation_label[idx],
            frame_indices=out["ex_local"],
            original_indices=idx,
            moving=moving[comp],
            valid_comp_mask=moving[comp],
            gt_box_label=torch.zeros_like(comp),
        ))
        valid = out["valid_comp"] & ((out["comp_max_frame_id"] >= frame_id + self.min_move_frame) |
                                     (out["comp_min_frame_id"] <= frame_id - self.min_move_frame))
        extracted = EasyDict(filter_dict(extracted, valid[extracted.component]))
        extracted.transforms = out["transforms"]
        seq_points.extracted.index_fill_(0, extracted.original_indices.reshape(-1), True)  # (an indexed assignment synchronises)
        self.last_track_debug = dict(ticks=out.get("_ticks", 0), reg_errors=out["reg_errors"],
                                     comp_edge_ratios=out["comp_edge_ratios"],
                                     comp_velos=out["comp_velos"],
                                     comp_min_frame_id=out["comp_min_frame_id"],
                                     comp_max_frame_id=out["comp_max_frame_id"])
        return extracted

    def _extract_units_native(self, seq_fxyz, units, tracked, perm, views, seq_boxes, windows):
        """extract_traces_of_units straight from the native tracker's outputs of ALL units of a sequence: what
        `_extracted_from_native` + the head of `extract_traces_of_units` do unit by unit -- the reference's final
        validity filter (tracked >= MIN_MOVE_FRAME frames, cluster_tracking.py:753 of the reference), the gathers of
        the tracked voxels, the per-unit component ranges -- done once on the concatenated rows (two host reads per
        sequence instead of three per unit; the intermediate per-unit dictionaries, whose other fields nothing
        downstream reads, are not built). `windows`: the units' (first, last) tracked frame, host integers. Same
        results as the unit-by-unit path, value for value."""
        dev = seq_fxyz.device
        outs = [t[0] for t in tracked]
        n_units = len(outs)
        n_ex = [o["ex_index"].shape[0] for o in outs]
        n_c = [o["moving"].shape[0] for o in outs]
        if sum(n_ex) == 0:
            return [None] * n_units
        idx = torch.cat([o["ex_index"] for o in outs])
        if perm is not None:
            idx = perm[idx]
        comp = torch.cat([o["ex_component"] for o in outs])
        unit0 = run_ids(n_ex, dev)
        cbase = [0]
        for c in n_c:
            cbase.append(cbase[-1] + c)
        comp_g = comp + torch.tensor(cbase[:-1], device=dev)[unit0]  # row of the units' concatenated [C] arrays
        moving_c = torch.cat([o["moving"] for o in outs])
        unit_c = run_ids(n_c, dev)  # the unit of every row of the concatenated [C] arrays
        anchor = torch.tensor([f for _, f in units], device=dev)[unit_c]
        valid_c = torch.cat([o["valid_comp"] for o in outs]) & (
            (torch.cat([o["comp_max_frame_id"] for o in outs]) >= anchor + self.min_move_frame) |
            (torch.cat([o["comp_min_frame_id"] for o in outs]) <= anchor - self.min_move_frame))
        keep = valid_c[comp_g].nonzero(as_tuple=True)[0]
        idx, comp, comp_g, unit_k = idx[keep], comp[keep], comp_g[keep], unit0[keep]
        fxyz = seq_fxyz[idx]
        moving = moving_c[comp_g]
        # per unit: the largest component id among its rows (one host read; a unit without rows: -1) -- from the
        # components that have a row, not from the rows (a reduction of half a million rows into nine words is a
        # queue of atomics on nine addresses); the frame range is the unit's tracking window, which the host knows
        present = torch.zeros(cbase[-1], dtype=torch.bool, device=dev)
        present[comp_g] = True
        local1 = torch.arange(1, cbase[-1] + 1, device=dev) - torch.tensor(cbase[:-1], device=dev)[unit_c]
        top = torch.segment_reduce(torch.where(present, local1, torch.zeros_like(local1)).float(), "max",
                                   lengths=torch.tensor(n_c, device=dev), unsafe=True).long().tolist()
        todo = [u for u in range(n_units) if top[u] > 0]
        res = [None] * n_units
        if not todo:
            return res
        if len(todo) < n_units:  # (renumber the units that have rows: the rows are already grouped in unit order)
            remap = torch.full((n_units,), -1, dtype=torch.int64, device=dev)
            remap[torch.tensor(todo, device=dev)] = torch.arange(len(todo), device=dev)
            unit_k = remap[unit_k]
        full = self._extract_units_core(
            views, seq_boxes, fxyz, comp, moving, None,
            [(top[u] - 1, windows[u][0], windows[u][1]) for u in todo], [tracked[u][1] for u in todo],
            [outs[u]["transforms"] for u in todo], unit=unit_k)
        for u, f in zip(todo, full):
            res[u] = f
        return res

    def _track_frame_native(self, seq_points, frame, offsets):
        frame_id = int(frame.frame[0].long().item())
        n_a = offsets[frame_id + 1] - offsets[frame_id]
        assert frame.fxyz.shape[0] == n_a, "anchor rows must be the frame's rows of seq_points"
        perm = seq_points.get("frame_perm", None)
        s_fxyz, s_comp, s_stat = seq_points.fxyz, seq_points.component, seq_points.stationary
        if perm is not None:  # rows not grouped by frame: track on the grouped copy, map rows back
            s_fxyz, s_comp, s_stat = s_fxyz[perm].contiguous(), s_comp[perm], s_stat[perm]
        num_components = int(frame.component.max().long().item()) + 1
        comp_base = int((s_comp[offsets[frame_id]] - frame.component[0]).long().item())
        min_frame_id, max_frame_id = self._native_window(offsets, frame_id)
        if self.verbose:
            print(f"Tracking frame-{frame_id:03d} with {num_components:03d} components (native)")
        out, (_, reg_calls, _) = ops.track_frame(
            self._native_params(), s_fxyz, s_comp, s_stat, offsets,
            frame_id, min_frame_id, max_frame_id, comp_base, num_components)
        self.num_registration_calls += reg_calls
        return self._extracted_from_native(seq_points, frame_id, out, perm, frame.component.dtype)

    def track_frame_host(self, seq_points, frame, seq_boxes):
        """track_frame op by op (one framework call per step of cluster_tracking.py:430-787): the same
        loop in four pieces (host_init / host_start_dir / host_step / host_finish) so that a test can
        run one frame step from a given state (tests/test_gpu_track_steps.py)."""
        st = self.host_init(seq_points, frame)
        for track_dir in (-1, 1):  # backward, then forward
            self.host_start_dir(st, seq_points, track_dir)
            while self.host_can_step(st):
                self.host_step(st, seq_points)
        frame.fxyz = seq_points.fxyz[st.frame_mask]
        return self.host_finish(st, seq_points)

    def host_init(self, seq_points, frame):
        st = EasyDict(frame=frame)
        C = st.num_components = int(frame.component.max().long().item()) + 1
        frame_id = st.frame_id = int(frame.frame[0].long().item())
        st.frame_mask = (seq_points.frame == frame_id).reshape(-1)
        st.min_frame_id = max(int(seq_points.frame.min().long().item()), frame_id - self.track_interval)
        st.max_frame_id = min(int(seq_points.frame.max().long().item()), frame_id + self.track_interval)
        st.comp_sg = IndexSegments(frame.component, C)
        st.comp_deg = efficient_robust_sum(torch.ones_like(frame.component), st.comp_sg, C)
        if self.verbose:
            print(f"Tracking frame-{frame_id:03d} with {C:03d} components")
        st.comp_diameter = component_diameter(frame, st.comp_sg)

        F = st.max_frame_id + 1
        st.transforms = torch.diag_embed(
            frame.fxyz.new_ones(C, st.max_frame_id - st.min_frame_id + 1, 4).double())
        st.reg_errors = frame.fxyz.new_zeros(C, F)
        st.comp_edge_ratios = frame.fxyz.new_zeros(C, F)
        st.comp_min_frame_id = frame.component.new_zeros(C) + frame_id
        st.comp_max_frame_id = frame.component.new_zeros(C) + frame_id
        st.comp_velos = frame.fxyz.new_zeros(C, F, 3)
        st.comp_centers = frame.fxyz.new_zeros(C, F, 3)
        st.comp_centers[:, frame_id] = component_center(frame, st.comp_sg)
        st.comp_center_diffs = frame.fxyz.new_zeros(C, F, 3)

        st.valid_comp_mask = filter_components(frame, sg=st.comp_sg)
        vpm = st.valid_comp_mask[frame.component]
        st.ex_fxyz = [frame.fxyz[vpm]]
        st.ex_component = [frame.component[vpm]]
        st.ex_seg = [frame.segmentation_label[vpm]]
        st.ex_indices = [vpm.nonzero().reshape(-1)]
        st.ex_original = [st.frame_mask.nonzero()[vpm].reshape(-1)]
        st.last_velo = None
        st.moving = None
        st.smooth_iters = None  # tests only: a fixed AdamW step count for smooth_velo
        return st

    def host_start_dir(self, st, seq_points, track_dir):
        st.track_dir = track_dir
        st.next_frame_id = st.frame_id + track_dir
        st.frame.fxyz = seq_points.fxyz[st.frame_mask]
        st.stopped = ~st.valid_comp_mask
        st.moving = st.valid_comp_mask.clone()
        st.last_xyz = st.frame.fxyz[:, 1:].clone()
        if track_dir == 1 and st.frame_id > 0:
            st.last_velo = st.comp_velos[:, st.frame_id]

    def host_can_step(self, st):
        return st.min_frame_id <= st.next_frame_id <= st.max_frame_id and bool((~st.stopped).any())

    def host_step(self, st, seq_points):
        """One frame step; returns the harvest's edges (f_this, f_next) before the stop filter."""
        frame, frame_id, track_dir, next_frame_id = st.frame, st.frame_id, st.track_dir, st.next_frame_id
        num_components, comp_sg = st.num_components, st.comp_sg
        transforms, comp_velos, comp_centers = st.transforms, st.comp_velos, st.comp_centers
        comp_center_diffs, comp_diameter = st.comp_center_diffs, st.comp_diameter
        stopped, moving, last_velo = st.stopped, st.moving, st.last_velo
        if self.verbose:
            print(f"Tracking {frame_id} --> {next_frame_id}")
        next_mask = (seq_points.frame == next_frame_id).reshape(-1)
        next_frame = EasyDict(filter_dict(seq_points, next_mask, ignore_keys=("frame_offsets", "frame_perm")))
        col = next_frame_id - st.min_frame_id
        frame.require_corres = ((~stopped)[frame.component]).reshape(-1)
        transforms[:, col] = transforms[:, col - track_dir]
        if last_velo is not None:
            # constant-velocity prediction for the components still being tracked
            trans = last_velo.clone()
            trans[stopped] = 0
            frame.fxyz[:, 1:] += trans[frame.component] * track_dir
            transforms[:, col, :3, 3] += trans.double() * track_dir
        for i in range(len(self.radius_list)):
            sampler = getattr(self, f"sampler_{i}")
            graph = getattr(self, f"registration_graph_{i}")
            sub_frame = sample_frame(sampler, frame)
            sub_next = sample_frame(sampler, next_frame)
            self.num_registration_calls += 1
            with Timer(f"REGISTRATION at Level {i}"):
                sub_frame, T, _l1, _ratio = register_to_next_frame(
                    graph, sub_frame, sub_next, num_components,
                    self.model_cfg["ANGLE_REGULARIZER"], max_iter=80,
                    stopping_delta=self.stopping_delta[i])
            if i == 0:
                comp_edge_ratio = _ratio
            if i == len(self.radius_list) - 1:
                l1_reg_error = _l1
            Tc = T[frame.component]
            frame.fxyz[:, 1:] = (Tc[:, :3, :3] @ frame.fxyz[:, 1:, None].double()).squeeze(-1).float() \
                + Tc[:, :3, 3].float()
            transforms[:, col] = T @ transforms[:, col]

        comp_centers[:, next_frame_id] = component_center(frame, comp_sg)
        point_velo = (frame.fxyz[:, 1:] - st.last_xyz) * track_dir
        comp_velo = robust_mean(point_velo, comp_sg, num_components)
        comp_velo[:, 2] = 0
        comp_velos[:, next_frame_id] = comp_velo
        comp_center_diffs[:, next_frame_id] = \
            (comp_centers[:, next_frame_id] - comp_centers[:, next_frame_id - track_dir]) * track_dir
        if st.smooth_iters is None:
            smooth_velo(comp_velos, comp_center_diffs, frame_id + track_dir, next_frame_id)
        else:
            smooth_velo(comp_velos, comp_center_diffs, frame_id + track_dir, next_frame_id,
                        num_itr=st.smooth_iters, stopping=-1e30)
        delta_velo = comp_velos[:, next_frame_id] - comp_velo
        comp_velo = comp_velos[:, next_frame_id]  # live view, like the reference's
        frame.fxyz[:, 1:] += delta_velo[frame.component] * track_dir
        transforms[:, col, :3, 3] += delta_velo * track_dir
        st.last_xyz = frame.fxyz[:, 1:].clone()

        # ---- stop rules (cluster_tracking.py:676-691)
        stopped = stopped | (l1_reg_error > self.reg_error_coeff * comp_diameter *
                             (1 + dist_compensate(st.comp_deg)))
        stopped = stopped | (comp_edge_ratio < 0.5)
        if (next_frame_id - frame_id) * track_dir == self.min_move_frame:
            moved = (comp_centers[:, next_frame_id] - comp_centers[:, frame_id]).norm(p=2, dim=-1)
            moving = moving & (moved > 0.08 * comp_diameter)
        if last_velo is not None:
            dev_velo_norm = (comp_velo - last_velo).norm(p=2, dim=-1)
            stopped = stopped | (dev_velo_norm > 0.24 * comp_diameter)
            prev_velo = comp_velos[:, next_frame_id - track_dir]
            norm = (comp_velo.norm(p=2, dim=-1) * prev_velo.norm(p=2, dim=-1)).clamp(min=1e-6)
            angle = ((comp_velo * prev_velo).sum(-1) / norm).clamp(min=-1, max=1).arccos() \
                / np.pi * 180.0
            stopped = stopped | (angle > self.angle_threshold) & \
                (comp_velos[:, next_frame_id, :2].norm(p=2, dim=-1) > 0.01)
        st.last_velo = comp_velo
        if next_frame_id == frame_id - 1:
            comp_velos[:, frame_id] = comp_velo
        if track_dir == -1:
            st.comp_min_frame_id[~stopped] = next_frame_id
        else:
            st.comp_max_frame_id[~stopped] = next_frame_id

        # ---- harvest the target frame's points next to the moved anchor points
        frame.fxyz[:, 0] = next_frame_id
        f_this, f_next, _ = self.nn_graph(frame, next_frame)
        ok = (~stopped)[frame.component[f_this]]
        st.ex_fxyz.append(next_frame.fxyz[f_next[ok]])
        st.ex_component.append(frame.component[f_this[ok]])
        st.ex_seg.append(next_frame.segmentation_label[f_next[ok]])
        st.ex_indices.append(f_next[ok].reshape(-1))
        st.ex_original.append(next_mask.nonzero()[f_next[ok]].reshape(-1))
        frame.fxyz[:, 0] = frame_id

        st.reg_errors[:, next_frame_id] = l1_reg_error
        st.comp_edge_ratios[:, next_frame_id] = comp_edge_ratio
        st.l1_reg_error, st.comp_edge_ratio = l1_reg_error, comp_edge_ratio
        st.stopped, st.moving = stopped, moving
        st.next_frame_id = next_frame_id + track_dir
        return f_this, f_next

    def host_finish(self, st, seq_points):
        frame_id = st.frame_id
        ex_component = torch.cat(st.ex_component, dim=0)
        extracted = EasyDict(dict(
            fxyz=torch.cat(st.ex_fxyz, dim=0),
            component=ex_component,
            segmentation_label=torch.cat(st.ex_seg, dim=0),
            frame_indices=torch.cat(st.ex_indices, dim=0),
            original_indices=torch.cat(st.ex_original, dim=0),
            moving=st.moving[ex_component],
            valid_comp_mask=st.moving[ex_component],
            gt_box_label=torch.zeros_like(ex_component),
        ))
        valid_comp_mask = st.valid_comp_mask & ((st.comp_max_frame_id >= frame_id + self.min_move_frame) |
                                                (st.comp_min_frame_id <= frame_id - self.min_move_frame))
        extracted = EasyDict(filter_dict(extracted, valid_comp_mask[extracted.component]))
        extracted.transforms = st.transforms
        seq_points.extracted.index_fill_(0, extracted.original_indices.reshape(-1), True)  # (an indexed assignment synchronises)
        self.last_track_debug = dict(reg_errors=st.reg_errors, comp_edge_ratios=st.comp_edge_ratios,
                                     comp_velos=st.comp_velos, comp_min_frame_id=st.comp_min_frame_id,
                                     comp_max_frame_id=st.comp_max_frame_id)
        return extracted

    def _track_units_native(self, seq_dict, key_points, units, offsets, perm):
        """All (component key, anchor frame) units through ops.track_sequence. While the native call
        runs (GIL released) a helper thread does the parked host-side work (ClusterProposal's
        evaluation) on a side stream."""
        import threading
        descs, dtypes = [], []
        grouped = {}
        ranges = []
        for comp_key, frame_id in units:
            sp = key_points[comp_key]
            if comp_key not in grouped:
                s_comp, s_stat = sp.component, sp.stationary
                if perm is not None:
                    s_comp, s_stat = s_comp[perm], s_stat[perm]
                grouped[comp_key] = (s_comp.long().contiguous(), s_stat.to(torch.uint8).contiguous())
            s_comp, s_stat = grouped[comp_key]
            lo, hi = offsets[frame_id], offsets[frame_id + 1]
            ranges.append(torch.stack(torch.aminmax(s_comp[lo:hi])))
        # the component id ranges of all units in ONE host read (they were two reads per unit)
        ranges = torch.stack(ranges).tolist()
        for (comp_key, frame_id), (cmin, cmax) in zip(units, ranges):
            s_comp, s_stat = grouped[comp_key]
            min_frame_id, max_frame_id = self._native_window(offsets, frame_id)
            descs.append(dict(component=s_comp, stationary=s_stat, anchor_frame=frame_id,
                              min_frame_id=min_frame_id, max_frame_id=max_frame_id, comp_base=int(cmin),
                              num_components=int(cmax) - int(cmin) + 1))
            dtypes.append(key_points[comp_key].component.dtype)
        s_fxyz = key_points[units[0][0]].fxyz
        if perm is not None:
            s_fxyz = s_fxyz[perm].contiguous()
        helper = None
        if seq_dict.get("_deferred"):
            device = s_fxyz.device
            main = torch.cuda.current_stream(device)
            side = thread_stream(device, 1)  # the calling thread's second stream, the same at every call
            side.wait_stream(main)
            grad = torch.is_grad_enabled()
            err = []

            def side_job():
                try:
                    with torch.cuda.device(device), torch.cuda.stream(side), torch.set_grad_enabled(grad):
                        run_deferred(seq_dict)
                except BaseException as e:  # re-raised on the main thread
                    err.append(e)

            helper = _deferred_pool().submit(side_job)
        # UNITS_PER_CALL (default: all): how many units share a native call; 1 = one unit after the
        # other (same results bit for bit -- a unit never depends on what runs beside it)
        per_call = int(self.model_cfg.get("UNITS_PER_CALL", 0)) or len(descs)
        per_call = max(1, min(per_call, ops.MAX_TRACK_UNITS))
        try:
            res = []
            for b in range(0, len(descs), per_call):
                res += ops.track_sequence(self._native_params(), s_fxyz, offsets, descs[b:b + per_call])
        finally:
            if helper is not None:
                helper.result()
                main.wait_stream(side)
                if err:
                    raise err[0]
        self.num_registration_calls += sum(r[1][1] for r in res)
        self.last_ticks = max(r[0]["_ticks"] for r in res)
        return [(r[0], dt) for r, dt in zip(res, dtypes)]

    # ------------------------------------------------------------------ whole sequence
    def forward(self, seq_dict):
        seq_points = EasyDict(
            fxyz=seq_dict["point_fxyz"],
            frame=seq_dict["point_sweep"],
        )
        if "point_gt_box_id" in seq_dict:  # absent while ClusterProposal's evaluation is deferred
            seq_points.gt_box_id = seq_dict["point_gt_box_id"]
        for key in ("instance_label", "segmentation_label"):
            if key in seq_dict:
                seq_points[key] = seq_dict[key]
        all_points = EasyDict(
            fxyz=seq_dict["full_point_fxyz"],
            frame=seq_dict["full_point_sweep"],
            height=seq_dict["full_point_height"],
        )
        for key in ("full_instance_label", "full_segmentation_label"):
            if key in seq_dict:
                all_points[key] = seq_dict[key]
        all_points = EasyDict(filter_dict(all_points, seq_dict["full_point_height"] > 0.0))

        num_frames = int(seq_points.frame.max().long().item()) + 1
        self.num_registration_calls = 0
        sequence_id = seq_dict["frame_id"][0][:-4]
        outfolder = f"{self.model_cfg['DIR']}/{sequence_id}"
        outpath = f"{outfolder}/all.pth"
        save = self.model_cfg.get("SAVE", True)
        if save and os.path.exists(outpath):
            print(f"{outpath} already exists. skipping...")
            return seq_dict
        if save:
            os.makedirs(outfolder, exist_ok=True)

        seq_boxes = self.format_boxes(seq_dict, num_frames)
        if seq_boxes.attr.shape[0] == 0:
            return seq_dict
        seq_boxes.best_iou = torch.zeros_like(seq_boxes.attr[:, 0])

        # The 3 keys x anchors {0, 8, 16, ...} are independent units (the reference runs them one
        # after the other). Each unit is a chain of small, latency-bound launches, so the native
        # tracker advances ALL of them in lock-step from one host thread on one stream
        # (ops.track_sequence -> pcs_track_sequence): every launch serves all units that are in the
        # same phase. best_iou is a running maximum, so per-unit copies merged by max equal the
        # serial result.
        frame_offsets, frame_perm = self.frame_offsets_of(seq_points.frame) if self.native else (None, None)
        key_points = {}
        for comp_key in self.component_keys:
            sp = EasyDict(dict(seq_points))
            sp.frame_offsets, sp.frame_perm = frame_offsets, frame_perm
            sp.component = seq_dict[f"point_{comp_key}"]
            diam = component_diameter(sp)
            sp.component_diameter = diam[sp.component]
            sp.stationary = sp.component_diameter > 12.5
            key_points[comp_key] = sp
        units = []
        if frame_offsets is not None and bool((seq_points.fxyz[:, 0] == seq_points.frame.reshape(-1).to(seq_points.fxyz)).all()):
            # (the frame column IS the frame id: the row ranges already on the host say which frames have points)
            present = [f for f in range(len(frame_offsets) - 1) if frame_offsets[f + 1] > frame_offsets[f]]
        else:
            present = torch.unique(seq_points.fxyz[:, 0]).long().tolist()
        for comp_key in self.component_keys:
            for frame_id in range(0, num_frames, self.track_interval):
                if frame_id in present:
                    units.append((comp_key, frame_id))

        views = self.sequence_views(all_points, seq_boxes)

        def unit_points(comp_key):
            sp = EasyDict(dict(key_points[comp_key]))
            sp.extracted = torch.zeros_like(sp.fxyz[:, 0]).bool()
            return sp

        def run_unit_host(comp_key, frame_id):
            sp = unit_points(comp_key)
            frame_mask = (sp.fxyz[:, 0] == frame_id).reshape(-1)
            frame_points = EasyDict(filter_dict(sp, frame_mask, ignore_keys=("frame_offsets", "frame_perm")))
            frame_points.component = frame_points.component - frame_points.component.min()
            with Timer(f"Tracking Frame {frame_id}"):
                return self.track_frame_host(sp, frame_points, None)

        tracked = None
        self.last_num_workers = 1
        if frame_offsets is not None and seq_points.fxyz.is_cuda:
            try:
                with Timer("Track all units (native, lock-step)"):
                    tracked = self._track_units_native(seq_dict, key_points, units, frame_offsets, frame_perm)
                self.last_num_workers = len(units)
            except PcsError as err:
                if "PCS_ERR_RANGE" not in str(err):
                    raise
                # a search grid / fixed-point range overflowed inside the native tracker: the op-by-op
                # host loop gives the same results, two orders of magnitude slower -- never silently
                note_host_fallback(sequence_id, err)
        run_deferred(seq_dict)  # (already done beside the native call when that ran)
        # best_iou is a running maximum over the units: one pass over all of them
        if tracked is not None and self.model_cfg.get("EXTRACT_UNITS_TOGETHER", True):
            # the native tracker's outputs of all units go through the final filter and the extraction together
            with Timer("Extract Traces from all points"):
                full_list = self._extract_units_native(
                    seq_points.fxyz, units, tracked, frame_perm, views, seq_boxes,
                    [self._native_window(frame_offsets, frame_id) for _, frame_id in units])
            for i, (comp_key, frame_id) in enumerate(units):
                if full_list[i] is None:  # (no tracked voxel survived the filter: the unit's own, empty dictionary)
                    out, comp_dtype = tracked[i]
                    full_list[i] = self._extracted_from_native(unit_points(comp_key), frame_id, out, frame_perm,
                                                               comp_dtype)
        else:
            extracted_list = []
            for i, (comp_key, frame_id) in enumerate(units):
                if tracked is not None:
                    out, comp_dtype = tracked[i]
                    extracted_list.append(self._extracted_from_native(unit_points(comp_key), frame_id, out,
                                                                      frame_perm, comp_dtype))
                else:
                    extracted_list.append(run_unit_host(comp_key, frame_id))
            with Timer("Extract Traces from all points"):
                full_list = self.extract_traces_of_units(all_points, extracted_list, seq_boxes, views)
        results = {(frame_id, comp_key): ex for (comp_key, frame_id), ex in zip(units, full_list)}
        for (frame_id, comp_key), extracted in results.items():
            if save:
                # (the per-unit arrays are views of the sequence's arrays: a file takes its own rows only)
                torch.save({k: (t.clone() if torch.is_tensor(t) and t._base is not None else t)
                            for k, t in dict(extracted).items()}, f"{outfolder}/{frame_id:03d}_{comp_key}.pth")
            extracted.pop("transforms", None)
        if save:
            torch.save(dict(seq_boxes), outpath)
        seq_dict["cluster_tracking_results"] = results
        seq_dict["gt_box_best_iou_tracking"] = seq_boxes.best_iou
        return seq_dict

    def get_output_feature_dim(self):
        return 0

    def extra_repr(self):
        return (f"reg_error_coeff={self.reg_error_coeff}, min_move_frame={self.min_move_frame}, "
                f"angle_threshold={self.angle_threshold}, track_interval={self.track_interval}")
