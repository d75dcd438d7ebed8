"""numpy restatement of the evaluation / full-resolution harvest / orchestration around the tracker:

  evaluate_proposal                  cluster_proposal.py:142-285 (+ assign_instances_to_boxes :63-88)
  extract_traces_and_update_boxes    cluster_tracking.py:287-428
  forward                            cluster_tracking.py:789-915 (what it saves: `<frame:03d>_<key>` and `all`)

TEST INFRASTRUCTURE ONLY (see oracle/oracle.py). Written as the reference's own loops (per frame, per
component) so that the product's vectorised forms are checked against the plain statement. Pinned by
tests/golden/forward_small.npz, produced by the reference's own code (tests/golden/make_golden_forward.py).
Leaf ops: oracle.points_in_boxes (roiaware_pool3d.cpp:160-165), oracle.radius_graph, track_oracle.
"""
import numpy as np

from . import oracle as O
from . import track_oracle as T

F32 = np.float32


def trace_table(boxes):
    """The per-trace placeholders both stages build (cluster_proposal.py:167-178)."""
    num_traces = int(boxes["trace_id"].max()) + 1
    return dict(best_iou=np.zeros(num_traces, F32))


def assign_instances_to_boxes(labels, bp):
    """cluster_proposal.py:63-88: every instance goes to the box holding most of its points (first box
    on ties, as argmax), or to none. Returns (dict label -> box, point_box_id [n] i64)."""
    inst2box = {}
    point_box = np.full(len(labels), -1, np.int64)
    for lab in np.unique(labels).tolist():
        m = labels == lab
        bi = bp[:, m]
        if not bi.any():
            continue
        b = int(bi.astype(np.int64).sum(-1).argmax())
        inst2box[lab] = b
        point_box[m] = b
    return inst2box, point_box


def evaluate_proposal(fxyz, components, boxes):
    """fxyz [N,4]; components: list of [N] i64 (one per component key, in key order); boxes: dict(attr
    [B,7] f32, frame [B] i64, trace_id [B] i64). Returns the reference's outputs: point_gt_box_id,
    point_gt_trace_id, point_pred_trace_id, point_pred_box_id (those of the LAST key, as the reference
    overwrites them per key), gt_box_best_iou [B] f32, gt_trace_best_iou [T] f32, and
    best_iou_after[key index] [B] f32."""
    n = fxyz.shape[0]
    B = boxes["attr"].shape[0]
    out = {}
    if B == 0:
        for k in ("gt_box_id", "gt_trace_id", "pred_trace_id", "pred_box_id"):
            out[f"point_{k}"] = np.full(n, -1, np.int64)
        return out
    best = np.zeros(B, F32)
    traces = trace_table(boxes)
    frame_of = fxyz[:, 0]
    num_frames = int(frame_of.max()) + 1
    after = []
    for comp in components:
        ids = {k: np.full(n, -1, np.int64) for k in ("gt_box_id", "pred_box_id", "gt_trace_id", "pred_trace_id")}
        for f in range(num_frames):
            fm = frame_of == f
            if not fm.any():
                continue
            bm = boxes["frame"] == f
            if not bm.any():
                continue
            pts, c = fxyz[fm], comp[fm]
            tr = boxes["trace_id"][bm]
            fb_best = best[bm].copy()
            bp = O.points_in_boxes(pts[:, 1:], boxes["attr"][bm]).astype(np.int64)
            gt_box = np.full(len(c), -1, np.int64)
            gt_trace = np.full(len(c), -1, np.int64)
            inb = (bp == 1).any(0)
            gt_box[inb] = bp[:, inb].argmax(0)
            gt_trace[inb] = tr[gt_box[inb]]
            comp2box, pred_box = assign_instances_to_boxes(c, bp)
            pred_trace = np.full(len(c), -1, np.int64)
            v = pred_box >= 0
            pred_trace[v] = tr[pred_box[v]]
            for cc in np.unique(c).tolist():
                b = comp2box.get(cc, -1)
                if b < 0:
                    continue
                m1, m2 = gt_box == b, c == cc
                # float32 sums of 0/1 flags (exact), python-float division
                iou = float((m1 & m2).sum()) / (float((m1 | m2).sum()) + 1e-6)
                if iou > fb_best[b]:
                    fb_best[b] = iou
                if iou > traces["best_iou"][tr[b]]:
                    traces["best_iou"][tr[b]] = iou
            ids["gt_box_id"][fm], ids["gt_trace_id"][fm] = gt_box, gt_trace
            ids["pred_box_id"][fm], ids["pred_trace_id"][fm] = pred_box, pred_trace
            best[bm] = fb_best
        after.append(best.copy())
    for k, v in ids.items():
        out[f"point_{k}"] = v
    out["gt_box_best_iou"] = best
    out["gt_trace_best_iou"] = traces["best_iou"]
    out["best_iou_after"] = after
    return out


def extract_traces_and_update_boxes(all_points, extracted, boxes, nn_radius=0.5):
    """all_points: dict(fxyz [M,4], frame [M] i64, full_segmentation_label, full_instance_label);
    extracted: what track_frame returned (fxyz, component, moving, transforms, ...); boxes: dict(attr,
    frame, best_iou [B] f32 -- updated in place). Returns the reference's `full_extracted`."""
    transforms = extracted["transforms"]
    ecomp = extracted["component"]
    C = int(ecomp.max()) + 1
    component_hit = np.zeros(C, np.int64)
    ef = np.round(extracted["fxyz"][:, 0]).astype(np.int64)
    fmin = np.full(C, np.iinfo(np.int64).max)
    fmax = np.full(C, np.iinfo(np.int64).min)
    np.minimum.at(fmin, ecomp, ef)
    np.maximum.at(fmax, ecomp, ef)
    seen = fmax >= fmin
    # scatter(reduce='min'/'max') leaves untouched rows at 0
    component_size = np.where(seen, fmax - fmin + 1, 1)
    full = {k: [] for k in ("fxyz", "component", "segmentation_label", "instance_label", "original_indices",
                            "frame_indices", "moving")}
    aframe = all_points["frame"].reshape(-1)
    radius = float(F32(nn_radius) * F32(1.732))
    for fid in np.unique(ef).tolist():
        bm = boxes["frame"] == fid
        rm = aframe == fid
        ref = all_points["fxyz"][rm]
        ridx = np.nonzero(rm)[0]
        if bm.any():
            ref_bp = O.points_in_boxes(ref[:, 1:], boxes["attr"][bm]).astype(np.int64)
            ref_gt = ref_bp.argmax(0)
            ref_gt[ref_bp.max(0) == 0] = -1
        else:
            ref_gt = np.full(len(ref), -1, np.int64)
        em = ef == fid
        one_fxyz, one_comp, one_moving = extracted["fxyz"][em], ecomp[em], extracted["moving"][em]
        e_ext, e_ref = O.radius_graph(one_fxyz, ref, radius, 1, True)
        # robust_mean / scatter-max per component over this frame's tracked voxels (xy only)
        sg = T.Segs(one_comp, C)
        center = T.seg_sum(one_fxyz[:, 1:3], sg, mean=True)
        spread = norm2(one_fxyz[:, 1:3] - center[one_comp])
        diameter = np.zeros(C, F32)
        np.maximum.at(diameter, one_comp, spread)
        dz = one_fxyz[e_ext, -1] - ref[e_ref, -1]
        keep = dz < F32(0.5)
        dist = norm2(ref[e_ref, 1:3] - center[one_comp[e_ext]])
        keep &= dist < diameter[one_comp[e_ext]] + F32(0.05)
        keep &= dz > F32(-0.05)
        e_ext, e_ref = e_ext[keep], e_ref[keep]
        o = dict(fxyz=ref[e_ref], component=one_comp[e_ext],
                 segmentation_label=all_points["full_segmentation_label"][rm][e_ref],
                 instance_label=all_points["full_instance_label"][rm][e_ref],
                 frame_indices=e_ref.astype(np.int64), original_indices=ridx[e_ref].reshape(-1, 1),
                 moving=one_moving[e_ext])
        for k in full:
            full[k].append(o[k])
        if bm.any():
            fb_best = boxes["best_iou"][bm].copy()
            bp = O.points_in_boxes(o["fxyz"][:, 1:], boxes["attr"][bm]).astype(np.int64)
            for c in np.unique(o["component"]).tolist():
                cm = o["component"] == c
                if not bp[:, cm].any():
                    continue
                b = int(bp[:, cm].astype(F32).sum(-1).argmax())
                m1 = ref_gt == b
                mask = np.zeros(len(ref), bool)
                mask[o["frame_indices"][cm]] = True
                # long / (long + 1e-6) -> float32 tensor in torch
                iou = F32(F32((mask & m1).sum()) / F32(F32((mask | m1).sum()) + F32(1e-6)))
                if iou > F32(0.7):
                    component_hit[c] += 1
                if iou > fb_best[b]:
                    fb_best[b] = iou
            boxes["best_iou"][bm] = fb_best
    out = {k: np.concatenate(v) for k, v in full.items()}
    out["component_hit"] = component_hit
    out["component_size"] = component_size
    out["transforms"] = transforms
    return out


def norm2(x):
    x = x.astype(F32)
    return np.sqrt((x * x).sum(-1, dtype=F32)).astype(F32)


def forward(seq, tracker=None, track_interval=8):
    """cluster_tracking.py:789-915 on numpy inputs. seq: dict(point_fxyz [N,4], point_sweep [N],
    point_<key> [N] for every key of seq["component_keys"], full_point_fxyz [M,4], full_point_sweep [M],
    full_point_height [M], full_segmentation_label, full_instance_label, boxes dict(attr, frame, trace_id,
    moving)). Returns (saved, boxes): saved["<frame:03d>_<key>"] = the dict the reference torch.saves for
    that unit, boxes["best_iou"] the final per-box IoU (`all.pth`)."""
    tracker = tracker or T.Tracker(track_interval=track_interval)
    boxes = dict(seq["boxes"])
    saved = {}
    if boxes["attr"].shape[0] == 0:
        return saved, boxes
    boxes["best_iou"] = np.zeros(boxes["attr"].shape[0], F32)
    hm = seq["full_point_height"] > 0.0
    all_points = dict(fxyz=seq["full_point_fxyz"][hm], frame=seq["full_point_sweep"].reshape(-1)[hm],
                      full_segmentation_label=seq["full_segmentation_label"][hm],
                      full_instance_label=seq["full_instance_label"][hm])
    fxyz = seq["point_fxyz"]
    frame = seq["point_sweep"].reshape(-1).astype(np.int64)
    num_frames = int(frame.max()) + 1
    for key in seq["component_keys"]:
        comp = seq[f"point_{key}"]
        C = int(comp.max()) + 1
        diam = T.component_diameter(fxyz, comp, C)
        sp = dict(fxyz=fxyz.copy(), frame=frame, component=comp, stationary=diam[comp] > 12.5,
                  extracted=np.zeros(len(comp), bool), segmentation_label=seq["segmentation_label"])
        for frame_id in range(0, num_frames, tracker.track_interval):
            if not (fxyz[:, 0] == frame_id).any():
                continue
            ex = tracker.track_frame(sp, frame_id)
            ex["segmentation_label"] = sp["segmentation_label"][ex["original_indices"]]
            if ex["fxyz"].shape[0] > 0:
                ex = extract_traces_and_update_boxes(all_points, ex, boxes, tracker.nn_radius)
            saved[f"{frame_id:03d}_{key}"] = ex
    return saved, boxes


def format_boxes(attr, cls, obj_ids, num_frames):
    """simple_reg.py:35-101: frame ids of the frame-major boxes, empty boxes dropped, track labels =
    rank of the object id string, velocity = mean corner displacement to the trace's previous box (the
    first takes the second's), moving = velo > 0.05."""
    attr = attr.reshape(-1, 7).astype(F32)
    cls = cls.reshape(-1)
    assert attr.shape[0] % num_frames == 0
    frame = np.repeat(np.arange(num_frames), attr.shape[0] // num_frames).astype(cls.dtype)
    keep = norm2(attr[:, 3:6]) > F32(1e-5)
    attr, cls, frame = attr[keep], cls[keep], frame[keep]
    obj_ids = np.asarray(obj_ids).astype(str).reshape(-1)[keep]
    track = np.unique(obj_ids, return_inverse=True)[1].astype(np.int64)
    velo = np.zeros(attr.shape[0], F32)
    for t in np.unique(track).tolist():
        idx = np.nonzero(track == t)[0]
        order = np.argsort(frame[idx], kind="stable")
        corners = boxes_to_corners_3d(attr[idx][order])
        tv = np.zeros(len(idx), F32)
        if len(idx) > 1:
            d = corners[1:] - corners[:-1]
            tv[1:] = np.sqrt((d * d).sum(-1, dtype=F32)).mean(-1, dtype=F32)
            tv[0] = tv[1]
        velo[idx[order]] = tv
    return dict(attr=attr, cls_label=cls, frame=frame.astype(np.int64), trace_id=track, velo=velo,
                moving=velo > F32(5e-2), obj_ids=obj_ids)


def boxes_to_corners_3d(boxes):
    """pcdet/utils/box_utils.py boxes_to_corners_3d: 8 corners of (x,y,z,dx,dy,dz,heading) boxes."""
    t = np.array([[1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1],
                  [1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1]], F32) / F32(2)
    c = boxes[:, None, 3:6] * t[None]
    ang = boxes[:, 6]
    cos, sin = np.cos(ang).astype(F32), np.sin(ang).astype(F32)
    x = c[..., 0] * cos[:, None] - c[..., 1] * sin[:, None]
    y = c[..., 0] * sin[:, None] + c[..., 1] * cos[:, None]
    return np.stack([x, y, c[..., 2]], -1).astype(F32) + boxes[:, None, :3]
