"""ClusterProposal (pcdet/models/registration/preprocessors/cluster_proposal.py): per radius key
and per 10-frame chunk, a 32-nearest radius graph of the chunk on itself and its connected
components (ids offset by the running total), then the GT-box evaluation that tags every point
with its box / trace. Graph + components run on the HIP engine with no host round trip of the
edge list (the reference ships up to 32 N edges through scipy on the host)."""
from collections import defaultdict

import torch
import torch.nn as nn

from ..... import ops
from ....utils import timer
from ....utils.common_utils import EasyDict, filter_dict
from ....utils.timer import Timer
from ...model_utils import graph_utils
from .eval_utils import component_box_iou, component_box_iou_frames, count_into, scatter_max_


class ClusterProposal(nn.Module):
    def __init__(self, model_cfg, runtime_cfg):
        super().__init__()
        self.model_cfg = model_cfg
        self.forward_dict = EasyDict()
        self.fake_params = nn.Parameter(torch.zeros(1, dtype=torch.float32), requires_grad=True)
        self.component_keys = model_cfg["COMPONENT_KEYS"]
        for i, key in enumerate(self.component_keys):
            cfg = dict(model_cfg["GRAPH"])
            cfg["RADIUS"] = cfg["RADIUS"][i]
            self.add_module(f"graph_{key}", graph_utils.build_graph(cfg, runtime_cfg=runtime_cfg))
        self.chunk = 10
        self.verbose = bool((runtime_cfg or {}).get("verbose", False))

    def propose_cluster(self, seq_dict):
        pcseq = EasyDict(fxyz=seq_dict["point_fxyz"], frame=seq_dict["point_sweep"])
        num_points = pcseq.fxyz.shape[0]
        frames = pcseq.frame.reshape(-1)
        num_frames = int(frames.max().long().item()) + 1
        graphs = [getattr(self, f"graph_{k}") for k in self.component_keys]
        radii = [float(torch.tensor(float(g.radius), dtype=torch.float32)) for g in graphs]
        shared = (len({(g.max_num_neighbors, g.sort_by_dist, g.relative_key) for g in graphs}) == 1
                  and graphs[0].sort_by_dist and not self.model_cfg.get("SEPARATE_GRAPHS", False))
        components = {k: torch.zeros(num_points, dtype=torch.int64, device=pcseq.fxyz.device)
                      for k in self.component_keys}
        # (the running component offsets stay on the device: the counts are only ever added to labels)
        totals = {k: torch.zeros(1, dtype=torch.int64, device=pcseq.fxyz.device) for k in self.component_keys}
        for f0 in range(0, num_frames, self.chunk):
            mask = (frames >= f0) & (frames < f0 + self.chunk)
            idx = mask.nonzero(as_tuple=True)[0]
            if idx.numel() == 0:
                continue
            pcframe = EasyDict(filter_dict(pcseq, idx))
            n_chunk = pcframe.fxyz.shape[0]
            if shared:
                # The k nearest within r, listed nearest first, are the k nearest within R >= r cut at
                # r: ONE search at the largest radius serves every key (the reference builds one graph per
                # key), and the components of every cut come straight from the neighbour table
                # (ops.knn_components: no edge list, no masked gathers). The cut uses the engine's own
                # test, dist2 <= fl32(r) * fl32(r) in fp32.
                with Timer(f"Propose Cluster: graph r={max(radii):g} + components"):
                    g = graphs[0]
                    pts = pcframe[g.relative_key]
                    comp_all, num_all, n_edges = ops.knn_components(pts, radii, g.max_num_neighbors)
                    timer.count("proposal graph points", 2 * n_chunk)
                    timer.count("proposal graph edges", n_edges)
                    timer.count("proposal cc edges", n_edges * len(radii))
                    timer.count("proposal cc nodes", n_chunk * len(radii))
                for i, key in enumerate(self.component_keys):
                    components[key][idx] = comp_all[i] + totals[key]
                    totals[key] = totals[key] + num_all[i:i + 1]
                continue
            for key, graph, radius in zip(self.component_keys, graphs, radii):
                with Timer(f"Propose Cluster for {key}"):
                    e0, e1, _ = graph(pcframe, pcframe)
                    num, comp = graph_utils.connected_components(torch.stack([e0, e1], dim=0), n_chunk,
                                                                 host_count=False)
                    timer.count("proposal cc edges", e0.shape[0])
                    timer.count("proposal cc nodes", n_chunk)
                    components[key][idx] = comp.to(torch.int64) + totals[key]
                    totals[key] = totals[key] + num.reshape(1)
        for key in self.component_keys:
            seq_dict[f"point_{key}"] = components[key]
            if self.verbose:
                print(f"Cluster Proposal {key}: num_components={int(totals[key].item())}")
        return seq_dict

    def assign_instances_to_boxes(self, point_instance_label, bp_mask):
        """Majority box of every instance: (dict instance -> box id, per-point box id)."""
        instance2box = defaultdict(lambda: -1)
        point_box_id = torch.zeros_like(point_instance_label) - 1
        for inst in point_instance_label.unique().tolist():
            imask = (point_instance_label == inst).reshape(-1)
            bi = bp_mask[:, imask]
            if not bool(bi.any()):
                continue
            box_id = bi.long().sum(-1).argmax()
            instance2box[inst] = box_id
            point_box_id[imask] = box_id
        return instance2box, point_box_id

    def format_boxes(self, seq_dict, num_frames):
        return EasyDict(dict(
            attr=seq_dict["gt_box_attr"].reshape(-1, 7),
            cls_label=seq_dict["gt_box_cls_label"].reshape(-1),
            frame=seq_dict["gt_box_frame"].reshape(-1),
            trace_id=seq_dict["gt_box_track_label"].reshape(-1),
        ))

    def evaluate_proposal(self, seq_dict):
        """Point-wise IoU of the proposed components against the GT boxes / traces
        (cluster_proposal.py:142-285); also emits point_gt_box_id etc. consumed downstream."""
        num_frames = int(seq_dict["point_sweep"].max().long().item()) + 1
        seq_boxes = self.format_boxes(seq_dict, num_frames)
        if seq_boxes.attr.shape[0] == 0:
            n = seq_dict[f"point_{self.component_keys[0]}"].numel()
            for key in ("gt_box_id", "gt_trace_id", "pred_trace_id", "pred_box_id"):
                seq_dict[f"point_{key}"] = seq_dict["segmentation_label"].new_zeros(n) - 1
            return seq_dict
        seq_boxes.best_iou = torch.zeros_like(seq_boxes.attr[:, 0])
        num_traces = int(seq_boxes.trace_id.max().long().item()) + 1
        traces = EasyDict(dict(
            best_iou=seq_boxes.attr.new_zeros(num_traces),
            min_frame=seq_boxes.trace_id.new_zeros(num_traces),
            max_frame=seq_boxes.trace_id.new_zeros(num_traces),
        ))
        tid = seq_boxes.trace_id.long()
        fr = seq_boxes.frame.to(traces.min_frame.dtype)
        present = torch.zeros(num_traces, dtype=torch.bool, device=tid.device)
        present[tid] = True
        lo = torch.full_like(traces.min_frame, torch.iinfo(traces.min_frame.dtype).max)
        traces.min_frame = torch.where(present, lo.scatter_reduce(0, tid, fr, reduce="amin"), traces.min_frame)
        traces.max_frame = torch.where(present, torch.zeros_like(lo).scatter_reduce(0, tid, fr, reduce="amax",
                                                                                  include_self=False),
                                       traces.max_frame)

        # All frames at once (the reference loops over frames and, inside, over components): boxes are
        # grouped by frame, a point is tested against the boxes of its own frame
        # (ops.points_in_frame_boxes), and the per-(frame, component) statistics are histograms
        # (eval_utils.component_box_iou_frames). Box ids are indices within the frame, as in the
        # reference. best_iou is a running maximum, so the per-key snapshots are cumulative maxima.
        fxyz = seq_dict["point_fxyz"]
        proto = seq_dict[f"point_{self.component_keys[0]}"]
        dev = fxyz.device
        assert bool((seq_boxes.frame[1:] >= seq_boxes.frame[:-1]).all()), "boxes must be grouped by frame"
        box_cnt = count_into(seq_boxes.frame.long(), num_frames)
        box_off = torch.zeros(num_frames + 1, dtype=torch.int64, device=dev)
        box_off[1:] = box_cnt.cumsum(0)
        bmax = int(box_cnt.max().item())
        gt_box_id = torch.zeros_like(proto) - 1
        gt_trace_id = torch.zeros_like(proto) - 1
        pred = {k: (torch.zeros_like(proto) - 1, torch.zeros_like(proto) - 1) for k in self.component_keys}
        key_best = {k: torch.zeros_like(seq_boxes.best_iou) for k in self.component_keys}
        fr = fxyz[:, 0].round().long()
        sel = (box_cnt[fr.clamp(0, num_frames - 1)] > 0) & (fr >= 0) & (fr < num_frames)
        idx = sel.nonzero(as_tuple=True)[0]            # points of frames that have boxes
        s_fr = fr[idx]
        member, first = ops.points_in_frame_boxes(fxyz[idx, 1:].contiguous(), s_fr, seq_boxes.attr,
                                                  box_off.int(), bmax)
        gt_local = first.to(proto.dtype)
        in_box = gt_local >= 0
        g_global = box_off[s_fr] + gt_local.clamp(min=0)
        gt_box_id[idx] = gt_local
        gt_trace_id[idx] = torch.where(in_box, seq_boxes.trace_id[g_global].to(proto.dtype), gt_local)
        gt_cnt_all = count_into(g_global, seq_boxes.attr.shape[0], in_box)
        for comp_key in self.component_keys:
            comp = seq_dict[f"point_{comp_key}"][idx]
            gframe, _, inv, has_box, box, iou = component_box_iou_frames(comp, s_fr, member, gt_local,
                                                                        box_off, gt_cnt_all)
            p_box = torch.where(has_box[inv], box[inv], torch.full_like(inv, -1))
            gbox_pt = box_off[s_fr] + p_box.clamp(min=0)
            p_trace = torch.where(p_box >= 0, seq_boxes.trace_id[gbox_pt].to(p_box.dtype), p_box)
            pred[comp_key][0][idx] = p_box.to(proto.dtype)
            pred[comp_key][1][idx] = p_trace.to(proto.dtype)
            gbox = (box_off[gframe] + box)[has_box]
            scatter_max_(key_best[comp_key], gbox, iou[has_box].to(key_best[comp_key].dtype))
            scatter_max_(traces.best_iou, seq_boxes.trace_id[gbox].long(),
                         iou[has_box].to(traces.best_iou.dtype))
        for comp_key in self.component_keys:
            seq_boxes.best_iou = torch.maximum(seq_boxes.best_iou, key_best[comp_key])
            seq_boxes[f"best_iou_after_{comp_key}"] = seq_boxes.best_iou.clone()
        last = self.component_keys[-1]
        seq_points = EasyDict(gt_box_id=gt_box_id, gt_trace_id=gt_trace_id,
                              pred_box_id=pred[last][0], pred_trace_id=pred[last][1])
        seq_dict["gt_box_best_iou"] = seq_boxes.best_iou
        seq_dict["gt_trace_best_iou"] = traces.best_iou
        for key in ("gt_box_id", "gt_trace_id", "pred_trace_id", "pred_box_id"):
            seq_dict[f"point_{key}"] = seq_points[key]
        return seq_dict

    EVAL_KEYS = ("point_gt_box_id", "point_gt_trace_id", "point_pred_trace_id", "point_pred_box_id",
                 "gt_box_best_iou", "gt_trace_best_iou")

    def forward(self, seq_dict):
        seq_dict = self.propose_cluster(seq_dict)
        if not self.model_cfg.get("DEFER_EVALUATION", True):
            with Timer("Evaluate Proposal"):
                return self.evaluate_proposal(seq_dict)
        # No later stage reads the evaluation's outputs: parked for run_deferred (ClusterTracking runs
        # it on the main thread while its units track in native code; SimpleReg runs what is left).
        snapshot = EasyDict(dict(seq_dict))

        def evaluate():
            with Timer("Evaluate Proposal"):
                out = self.evaluate_proposal(snapshot)
            return {k: out[k] for k in self.EVAL_KEYS if k in out}

        seq_dict.setdefault("_deferred", []).append(evaluate)
        return seq_dict

    def get_output_feature_dim(self):
        return 0
