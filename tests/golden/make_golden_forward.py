"""Golden vectors for the stages around the tracker, produced by
IMPORTING the reference's own cluster_proposal.py / cluster_tracking.py in the authoring container
(shims as in make_golden_track.py: the CUDA-only leaves -- radius graph, grid sampling, points-in-boxes
-- are the CPU oracle's restatements, pinned separately) and running, on CPU tensors of a small
synthetic sequence with GT boxes:

  ClusterProposal.evaluate_proposal            -> point_gt_box_id / gt_trace_id / pred_box_id /
                                                  pred_trace_id, gt_box_best_iou, gt_trace_best_iou
  ClusterTracking.extract_traces_and_update_boxes
        on the reference's own track_frame(anchor 8) extraction -> the full-resolution harvest and the
        per-box IoUs (inputs AND outputs stored: the stage is deterministic given its input)
  ClusterTracking.forward                      -> everything it torch.saves (`<frame:03d>_<key>`,
                                                  `all`), captured instead of written

Inputs are built here in numpy (8 cm subsample, ground dropped by label, components from the oracle's
proposal, box formatting per simple_reg.py:35-101) and stored with the outputs in forward_small.npz.
The reference never travels: only inputs and outputs are stored.
Run: python tests/golden/make_golden_forward.py
"""
import importlib.util
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
from oracle import eval_oracle as E  # noqa: E402
from oracle import oracle as O  # noqa: E402
from oracle import track_oracle as T  # noqa: E402
from tests.golden import make_golden_track as MT  # noqa: E402
from pcseqlearning_amd import synthetic  # noqa: E402

KEYS = ["component_rad1x25", "component_rad0x75", "component_rad0x25"]
RADII = [1.25, 0.75, 0.25]


def small_inputs(seq_seed=3, num_frames=12, n_target=9000):
    """Everything the three stages read, as numpy (see the module docstring)."""
    batch = synthetic.gen_sequence(seq_seed, num_frames, n_target, n_static=14, n_vehicle=8, n_ped=4)
    full_fxyz = np.concatenate([batch["point_sweep"].astype(np.float32), batch["point_bxyz"][:, 1:]], 1)
    full_seg, full_inst = batch["segmentation_label"], batch["instance_label"]
    _, _, order, seg = O.grid_sample(full_fxyz, [0.08, 0.08, 0.08])
    pick = order[seg[1:] - 1]
    keep = full_seg[pick] != 17
    pick = pick[keep]
    inp = dict(point_fxyz=full_fxyz[pick], point_sweep=batch["point_sweep"][pick],
               segmentation_label=full_seg[pick], instance_label=full_inst[pick],
               full_point_fxyz=full_fxyz, full_point_sweep=batch["point_sweep"],
               full_point_height=np.where(full_seg == 17, -1.0, 1.0).astype(np.float32),
               full_segmentation_label=full_seg, full_instance_label=full_inst)
    for key, r in zip(KEYS, RADII):
        inp[f"point_{key}"] = T.propose_clusters(inp["point_fxyz"], inp["point_sweep"].reshape(-1), r)
    b = E.format_boxes(batch["gt_box_attr"][0], batch["gt_box_cls_label"][0], batch["obj_ids"][0], num_frames)
    inp.update(gt_box_attr=b["attr"], gt_box_cls_label=b["cls_label"], gt_box_frame=b["frame"],
               gt_box_track_label=b["trace_id"], gt_box_velo=b["velo"], moving=b["moving"])
    return inp, batch["frame_id"][0]


def load_reference_proposal():
    spec = importlib.util.spec_from_file_location(
        "refpre.cluster_proposal", f"{REF}/pcdet/models/registration/preprocessors/cluster_proposal.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules["refpre.cluster_proposal"] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    track_mod, EasyDict = MT.load_reference_tracking()
    sys.modules["torch_cluster"].knn = lambda *a, **k: None   # imported by name, never called here
    prop_mod = load_reference_proposal()
    inp, frame_ids = small_inputs()
    t = torch.from_numpy
    tmp = tempfile.mkdtemp()
    gold = {f"in_{k}": v for k, v in inp.items()}

    def seq_dict():
        d = EasyDict({k: t(np.ascontiguousarray(v).copy()) for k, v in inp.items()})
        d.frame_id = list(frame_ids)
        return d

    # ---- a8: evaluate_proposal
    pcfg = EasyDict(COMPONENT_KEYS=KEYS, DIR=tmp,
                    GRAPH=EasyDict(TYPE="RadiusGraph", RADIUS=RADII, MAX_NUM_NEIGHBORS=32, SORT_BY_DIST=True,
                                   RELATIVE_KEY="fxyz"))
    proposal = prop_mod.ClusterProposal(pcfg, {})
    sd = proposal.evaluate_proposal(seq_dict())
    for k in ("point_gt_box_id", "point_gt_trace_id", "point_pred_trace_id", "point_pred_box_id",
              "gt_box_best_iou", "gt_trace_best_iou"):
        gold[f"eval_{k}"] = sd[k].numpy()
    print("evaluate_proposal: boxes", sd["gt_box_best_iou"].shape[0], "mean best IoU",
          float(sd["gt_box_best_iou"].mean()), "points in boxes", int((sd["point_gt_box_id"] >= 0).sum()))
    point_gt_box_id = sd["point_gt_box_id"]

    # ---- a15: extract_traces_and_update_boxes on the reference's own extraction of anchor 8
    cfg = EasyDict({k: (EasyDict(v) if isinstance(v, dict) else v) for k, v in MT.MODEL_CFG.items()})
    cfg.REGISTRATION = EasyDict(MT.MODEL_CFG["REGISTRATION"])
    cfg.COMPONENT_KEYS = KEYS
    cfg.DIR = tmp
    tracker = track_mod.ClusterTracking(cfg, {"visualize": False})
    sd = seq_dict()
    sd.point_gt_box_id = point_gt_box_id
    sp = EasyDict(fxyz=sd.point_fxyz, frame=sd.point_sweep, gt_box_id=sd.point_gt_box_id,
                  segmentation_label=sd.segmentation_label, instance_label=sd.instance_label,
                  component=sd[f"point_{KEYS[0]}"])
    sp.component_diameter = track_mod.component_diameter(sp)[sp.component]
    sp.stationary = sp.component_diameter > 12.5
    sp.extracted = torch.zeros_like(sp.fxyz[:, 0]).bool()
    mask = (sp.fxyz[:, 0] == 8).reshape(-1)
    fp = EasyDict({k: v[mask] for k, v in sp.items()})
    fp.component = fp.component - fp.component.min()
    ex = tracker.track_frame(sp, fp, None)
    for k in ("fxyz", "component", "moving", "transforms", "original_indices", "frame_indices"):
        gold[f"ext_in_{k}"] = ex[k].numpy().copy()
    hm = sd.full_point_height > 0.0
    all_points = EasyDict(fxyz=sd.full_point_fxyz[hm], frame=sd.full_point_sweep[hm],
                          height=sd.full_point_height[hm],
                          full_segmentation_label=sd.full_segmentation_label[hm],
                          full_instance_label=sd.full_instance_label[hm])
    boxes = tracker.format_boxes(sd, 12)
    boxes.best_iou = torch.zeros_like(boxes.attr[:, 0])
    full, boxes = tracker.extract_traces_and_update_boxes(all_points, ex, boxes)
    for k in ("fxyz", "component", "segmentation_label", "instance_label", "original_indices", "frame_indices",
              "moving", "component_hit", "component_size"):
        gold[f"ext_out_{k}"] = full[k].numpy()
    gold["ext_out_best_iou"] = boxes.best_iou.numpy()
    print("extract_traces: harvested", full.fxyz.shape[0], "full-resolution points from", ex.fxyz.shape[0],
          "voxels; boxes touched", int((boxes.best_iou > 0).sum()), "hits", int(full.component_hit.sum()))

    # ---- a16: forward (what it saves)
    saved = {}
    orig_save = torch.save
    torch.save = lambda obj, path, *a, **k: saved.__setitem__(os.path.basename(path)[:-4], obj)
    try:
        sd = seq_dict()
        sd.point_gt_box_id = point_gt_box_id
        tracker.forward(sd)
    finally:
        torch.save = orig_save
    units = sorted(k for k in saved if k != "all")
    gold["fwd_units"] = np.array(units)
    for u in units:
        for k in ("fxyz", "component", "original_indices", "frame_indices", "moving", "component_hit",
                  "component_size"):
            if k in saved[u]:
                gold[f"fwd_{u}_{k}"] = saved[u][k].numpy()
        print("forward unit", u, "points", saved[u]["fxyz"].shape[0])
    gold["fwd_all_best_iou"] = saved["all"].best_iou.numpy()
    print("forward: box mIoU", float(saved["all"].best_iou.mean()))
    np.savez_compressed(os.path.join(HERE, "forward_small.npz"), **gold)


if __name__ == "__main__":
    main()
