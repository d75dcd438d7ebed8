"""CPU restatement of the reference's sequence loader. TEST INFRASTRUCTURE ONLY:
imported by tests/, never by the product path.

Follows, for the registration configs (tools/cfgs/dataset_configs/waymo/PCsequence/registration/
all_sequence.yaml: NUM_SWEEPS > 1, SYNC_MOVING_POINTS False, LOAD_SEG True,
DROP_POINTS_BY_LIDAR_INDEX, test mode, absolute_coordinates_encoding, DRY voxelisation):

  get_lidar        pcdet/datasets/waymo/waymo_dataset.py:334-343
  load_data        pcdet/datasets/waymo/waymo_dataset.py:366-500
  getitem          pcdet/datasets/waymo/waymo_dataset.py:502-682
  prepare_data     pcdet/datasets/dataset.py:143-191, processor/point_feature_encoder.py:41-55
  collate_batch    pcdet/datasets/dataset.py:193-298
  boxes_to_corners pcdet/utils/box_utils.py:28-53, common_utils.py:137-159 (the torch fp32 form)

Pinned: tests/test_loader_oracle.py checks every array against tests/golden/loader_fwd.npz and
loader_bwd.npz, which hold the outputs of the reference's own WaymoDataset / collate_batch imported
and run in the authoring container (tests/golden/make_golden_loader.py).
"""
import numpy as np
import torch


def get_lidar(raw):
    pts = np.array(raw, dtype=np.float32, copy=True)
    pts[:, 3] = np.tanh(pts[:, 3])
    pts[:, 5] /= 75.
    pts[:, 7] *= 64
    pts[:, 6] *= 2650
    return pts


def boxes_to_corners_3d(boxes):
    """box_utils.boxes_to_corners_3d on a numpy input: the arithmetic runs in torch fp32."""
    b = torch.from_numpy(np.asarray(boxes)).float()
    template = b.new_tensor(([1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1],
                             [1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1])) / 2
    corners = b[:, None, 3:6].repeat(1, 8, 1) * template[None, :, :]
    angle = b[:, 6]
    cosa, sina = torch.cos(angle), torch.sin(angle)
    zeros, ones = angle.new_zeros(b.shape[0]), angle.new_ones(b.shape[0])
    rot = torch.stack((cosa, sina, zeros, -sina, cosa, zeros, zeros, zeros, ones), dim=1).view(-1, 3, 3).float()
    corners = torch.matmul(corners.view(-1, 8, 3), rot).view(-1, 8, 3)
    corners += b[:, None, 0:3]
    return corners.numpy()


def points_in_boxes_any(points, boxes):
    """points_in_boxes_cpu(points, boxes).sum(0) > 0 (roiaware_pool3d.cpp:121-168)."""
    from . import oracle as O
    if boxes.shape[0] == 0:
        return np.zeros(points.shape[0], dtype=bool)
    m = O.points_in_boxes(np.ascontiguousarray(points[:, :3], dtype=np.float32),
                          np.ascontiguousarray(boxes[:, :7], dtype=np.float32))
    return m.sum(0) > 0


def load_data(info, raw, seg, drop_lidars):
    points = get_lidar(raw).astype(np.float32)
    pw = dict(point_xyz=points[:, :3], point_feat=points[:, 3:-2],
              segmentation_label=seg[:, 1], instance_label=seg[:, 0])
    if drop_lidars is not None:
        # NB the reference never advances `offset` (waymo_dataset.py:395-402): lidar i contributes
        # rows [0, n_i), which is right when only lidar 0 is kept
        keep = [np.arange(0, n) for i, n in enumerate(info["num_points_of_each_lidar"]) if i not in drop_lidars]
        idx = np.concatenate(keep, axis=0)
        pw = {k: v[idx] for k, v in pw.items()}
    annos = info["annos"]
    for drop in ("unknown", "Sign"):
        keep = [i for i, x in enumerate(annos["name"]) if x != drop]
        annos = {k: (v[keep] if isinstance(v, np.ndarray) else [v[i] for i in keep]) for k, v in annos.items()}
    boxes = annos["gt_boxes_lidar"]
    pw["is_foreground"] = points_in_boxes_any(pw["point_xyz"], boxes)
    ow = dict(gt_box_cls_label=annos["name"].astype(str), gt_box_attr=boxes,
              augmented=np.zeros(annos["name"].shape[0], dtype=bool), obj_ids=annos["obj_ids"],
              num_points_in_gt=annos["num_points_in_gt"])
    sw = dict(frame_id=info["frame_id"], pose=info["pose"].reshape(4, 4))
    return dict(point_wise=pw, scene_wise=sw, object_wise=ow)


def _pad_stack(dicts, pad_to):
    out = {}
    for key in dicts[0].keys():
        vals = []
        for d in dicts:
            v = d[key]
            if pad_to is not None and v.shape[0] < pad_to:
                v = np.concatenate([v, np.zeros((pad_to - v.shape[0], *v.shape[1:]), dtype=v.dtype)], axis=0)
            vals.append(v)
        out[key] = np.stack(vals, axis=0)
    return out


def getitem(infos, frames, segs, index, num_sweeps, sweep_dir, drop_lidars, box_classes, used_features,
            src_features):
    """infos: list of info dicts of ONE sequence ordered by sample_idx; frames / segs: raw arrays by
    sample_idx. Returns the sample dict of WaymoDataset.__getitem__(index) in test mode."""
    import copy
    pool = {i["point_cloud"]["sample_idx"]: i for i in infos}
    info = copy.deepcopy(infos[index])
    cur = info["point_cloud"]["sample_idx"]
    dicts = [load_data(info, frames[cur], segs[cur], drop_lidars)]
    if num_sweeps > 1:
        for ci in range(cur + sweep_dir, cur + sweep_dir * num_sweeps, sweep_dir):
            if ci not in pool:
                continue
            d = load_data(copy.deepcopy(pool[ci]), frames[ci], segs[ci], drop_lidars)
            dicts = [d] + dicts if sweep_dir == -1 else dicts + [d]
    T0_inv = np.linalg.inv(dicts[-1]["scene_wise"]["pose"].reshape(4, 4))
    max_obj = 0
    for d in dicts:
        T = T0_inv @ d["scene_wise"]["pose"].reshape(4, 4)
        pts = d["point_wise"]["point_xyz"]
        pts[:, :3] = pts[:, :3] @ T[:3, :3].T + T[:3, 3]
        n = pts.shape[0]
        d["point_wise"]["point_sweep"] = np.zeros((n, 1), dtype=np.int32) + int(d["scene_wise"]["frame_id"][-3:])
        boxes = d["object_wise"]["gt_box_attr"]
        corners = boxes_to_corners_3d(boxes)
        corners = (corners @ T[:3, :3].T + T[:3, 3]).reshape(-1, 24)
        boxes[:, :3] = boxes[:, :3] @ T[:3, :3].T + T[:3, 3]
        theta = boxes[:, -1]
        heading = np.stack([np.cos(theta), np.sin(theta), np.zeros_like(theta)], axis=-1)
        heading = heading @ T[:3, :3].T
        heading[:, :2] = heading[:, :2] / np.linalg.norm(heading[:, :2], ord=2, axis=-1)[:, np.newaxis]
        boxes[:, -1] = np.arctan2(heading[:, 1], heading[:, 0])
        d["object_wise"]["gt_box_attr"] = boxes
        d["object_wise"]["gt_box_corners_3d"] = corners
        max_obj = max(max_obj, boxes.shape[0])
    F = len(dicts)
    pw = {k: np.concatenate([d["point_wise"][k] for d in dicts], axis=0) for k in dicts[0]["point_wise"]}
    ow = _pad_stack([d["object_wise"] for d in dicts], max_obj)
    ow = {k: v.reshape(F * max_obj, *v.shape[2:]) for k, v in ow.items()}
    sw = _pad_stack([{k: np.asarray(v) for k, v in d["scene_wise"].items()} for d in dicts], None)
    # prepare_data, test mode
    names = ow["gt_box_cls_label"]
    cls = np.array([box_classes.index(n) + 1 if len(n) > 0 else 0 for n in names], dtype=np.int32)
    ow["gt_box_cls_label"] = cls
    ow["gt_boxes"] = np.concatenate((ow["gt_box_attr"], cls.reshape(-1, 1).astype(np.float32)), axis=1)
    if len(used_features) == 0:
        pw["point_feat"] = pw["point_feat"][:, :0]
    else:
        pw["point_feat"] = np.concatenate([pw["point_feat"][:, src_features.index(x):src_features.index(x) + 1]
                                           for x in used_features], axis=1)
    sw["num_sweeps"] = F
    return dict(point_wise=pw, object_wise=ow, scene_wise=sw)


def collate_batch(samples):
    B = len(samples)
    ret = {}
    concat = {"point_sweep", "point_feat", "segmentation_label", "is_foreground", "instance_label"}
    boxy = {"gt_boxes", "gt_box_attr", "gt_box_cls_label", "num_points_in_gt", "augmented", "gt_box_corners_3d"}
    keys = {}
    for s in samples:
        for group in s.values():
            for k, v in group.items():
                keys.setdefault(k, []).append(v)
    for k, val in keys.items():
        if k in concat:
            ret[k] = np.concatenate(val, axis=0)
        elif k == "point_xyz":
            ret["point_bxyz"] = np.concatenate([np.pad(c, ((0, 0), (1, 0)), mode="constant", constant_values=i)
                                                for i, c in enumerate(val)], axis=0)
        elif k in boxy:
            if k in ("gt_box_cls_label", "num_points_in_gt"):
                val, dtype = [v.reshape(-1, 1) for v in val], np.int32
            elif k == "augmented":
                val, dtype = [v.reshape(-1, 1) for v in val], bool
            else:
                dtype = np.float32
            m = max(len(x) for x in val)
            out = np.zeros((B, m, val[0].shape[-1]), dtype=dtype)
            for i in range(B):
                out[i, :len(val[i]), :] = val[i]
            ret[k] = out
        elif k == "obj_ids":
            ret["obj_ids"] = val
        else:
            ret[k] = np.stack(val, axis=0)
    ret["batch_size"] = B
    return ret
