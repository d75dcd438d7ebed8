"""`WaymoDataset` for the registration path: pre-extracted `%04d.npy` frames of a sequence ->
multi-sweep sample in the coordinate frame of the last sweep.

Mirrors pcdet/datasets/waymo/waymo_dataset.py (constructor + `include_waymo_data` :25-108,186-245;
`get_seg_label` :352-358; `load_data` :366-500; `__getitem__` :502-682) for the options of
tools/cfgs/dataset_configs/waymo/PCsequence/registration/*.yaml in test mode. What differs is
where the work happens: the raw rows of all sweeps are copied once into HBM and ONE HIP pass
(`pcs_assemble_sweeps`) normalises the features, applies the f64 pose products and tags the sweep;
`is_foreground` comes from `pcs_points_in_frame_boxes`. `load_sample_to_gpu` / `load_batch_to_gpu`
leave the point arrays on the device (what `SimpleReg` reads after `load_data_to_gpu`);
`__getitem__` returns the reference's host dictionaries. Per-box arithmetic (a few dozen boxes per
sweep) stays on the host in the reference's own precision. Options that change the sample and are
not implemented are refused in the constructor.
"""
import copy
import os
import pickle
from pathlib import Path

import numpy as np
import torch

from .... import ops
from ..dataset import DatasetTemplate

_UNSUPPORTED = {
    "SPHERICAL_RESAMPLING": False,
    "DROP_POINTS_BY_SEG_LEN": False, "MIX3D": None, "INFO_WITH_FAKELIDAR": False,
}


def boxes_to_corners_3d(boxes):
    """box_utils.boxes_to_corners_3d (pcdet/utils/box_utils.py:28-53) for a numpy [k,7] input: torch
    fp32 arithmetic on the host, corner order as in the reference's drawing."""
    b = torch.from_numpy(np.asarray(boxes)).float()
    template = b.new_tensor(([1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1],
                             [1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1])) / 2
    corners = b[:, None, 3:6].repeat(1, 8, 1) * template[None, :, :]
    cosa, sina = torch.cos(b[:, 6]), torch.sin(b[:, 6])
    zeros, ones = cosa.new_zeros(b.shape[0]), cosa.new_ones(b.shape[0])
    rot = torch.stack((cosa, sina, zeros, -sina, cosa, zeros, zeros, zeros, ones), dim=1).view(-1, 3, 3).float()
    corners = torch.matmul(corners.view(-1, 8, 3), rot).view(-1, 8, 3)
    corners += b[:, None, 0:3]
    return corners.numpy()


class WaymoDataset(DatasetTemplate):
    def __init__(self, dataset_cfg, training=True, root_path=None, logger=None):
        super().__init__(dataset_cfg=dataset_cfg, training=training, root_path=root_path, logger=logger)
        cfg = dataset_cfg
        for key, off in _UNSUPPORTED.items():
            if cfg.get(key, off) not in (off, None, False):
                raise NotImplementedError(f"WaymoDataset option {key}={cfg.get(key)} is not implemented")
        if training:
            raise NotImplementedError("training mode (augmentation, box filtering) is out of scope")
        self.data_path = self.root_path / cfg["PROCESSED_DATA_TAG"]
        self.split = cfg["DATA_SPLIT"][self.mode]
        with open(self.root_path / "ImageSets" / (self.split + ".txt")) as fh:
            self.sample_sequence_list = [x.strip() for x in fh.readlines()]
        if cfg.get("ONE_SEQUENCE", None) is not None:
            seq = cfg["ONE_SEQUENCE"] + ".tfrecord"
            if seq not in self.sample_sequence_list:
                raise ValueError(f"{seq} not in Sequence List")
            self.sample_sequence_list = [seq]
        if cfg.get("EXTERNAL_SEQUENCE_LIST", None) is not None:
            with open(cfg["EXTERNAL_SEQUENCE_LIST"]) as fh:
                wanted = [line.strip() + ".tfrecord" for line in fh.readlines()]
            for s in wanted:
                if s not in self.sample_sequence_list:
                    raise ValueError(f"{s} not in Sequence List")
            self.sample_sequence_list = wanted
        self.num_sweeps = cfg.get("NUM_SWEEPS", 1)
        self.with_time_feat = cfg.get("WITH_TIME_FEAT", False)  # waymo_dataset.py:59
        self.sync_moving_points = cfg.get("SYNC_MOVING_POINTS", False)  # waymo_dataset.py:61
        self.remove_ground = cfg.get("REMOVE_GROUND", False)  # waymo_dataset.py:62 (POINT_HEIGHT_PATH, TRUNCATE_GROUND_HEIGHT)
        ff = cfg.get("FILTER_FOREGROUND", None)  # waymo_dataset.py:65-69
        self.filter_foreground = ff is not None
        if ff is not None:
            self.pred_label_path, self.filter_class = ff["PRED_LABEL_PATH"], list(ff["FILTER_CLASS"])
        self.sweep_dir = cfg.get("SWEEP_DIR", -1)
        self.max_num_points *= self.num_sweeps
        self.infos = []
        self.include_waymo_data(self.mode)

    # ------------------------------------------------------------------ index of samples
    @staticmethod
    def check_sequence_name_with_all_version(sequence_file):
        """waymo_dataset.py:296-310: older extractions prefix / suffix the sequence directory."""
        if not sequence_file.exists():
            found = sequence_file
            for pre in ("training", "validation", "testing"):
                cand = Path(str(sequence_file).replace("segment", pre + "_segment"))
                if cand.exists():
                    found = cand
                    break
            if not found.exists():
                found = Path(str(sequence_file).replace("_with_camera_labels", ""))
            if found.exists():
                sequence_file = found
        return sequence_file

    def include_waymo_data(self, mode):
        cfg = self.dataset_cfg
        offset = cfg.get("SAMPLED_SEQUENCE_OFFSET_TRAIN" if mode == "train" else "SAMPLED_SEQUENCE_OFFSET_TEST", 0)
        waymo_infos, skipped = [], 0
        for k in range(offset, len(self.sample_sequence_list), cfg["SAMPLED_SEQUENCE_INTERVAL"][mode]):
            name = os.path.splitext(self.sample_sequence_list[k])[0]
            info_path = self.check_sequence_name_with_all_version(self.data_path / name / f"{name}.pkl")
            if not info_path.exists():
                skipped += 1
                continue
            with open(info_path, "rb") as fh:
                waymo_infos.extend(pickle.load(fh))
        self.infos.extend(waymo_infos)
        self.info_pool = {(i["point_cloud"]["lidar_sequence"], i["point_cloud"]["sample_idx"]): i for i in self.infos}
        rng = cfg.get("SAMPLE_IDX_RANGE", None)
        if rng is not None:
            lo, hi = rng[mode]
            self.infos = [i for i in waymo_infos if lo <= i["point_cloud"]["sample_idx"] < hi]
        if self.use_only_samples_with_seg_labels:
            keep = [i for i in self.infos if i["annos"].get("seg_label_path", None) is not None]
            keep = [i for i in keep if "_propseg.npy" not in i["annos"].get("seg_label_path", None)]
            self.infos = [i for i in keep if i["point_cloud"]["sample_idx"] >= self.num_sweeps - 1]
        step = cfg["SAMPLED_INTERVAL"][mode]
        if step > 1:
            self.infos = self.infos[::step]
        if self.logger is not None:
            self.logger.info(f"Waymo sequences: {len(waymo_infos)} frames indexed ({skipped} sequences skipped), "
                             f"{len(self.infos)} samples, {self.num_sweeps} sweeps each")

    def __len__(self):
        if self._merge_all_iters_to_one_epoch:
            return len(self.infos) * self.total_epochs
        return len(self.infos)

    # ------------------------------------------------------------------ files
    def read_lidar_file(self, sequence_name, sample_idx):
        """Raw rows of `%04d.npy`: (N, 8) [x, y, z, intensity, elongation, range, rimage_w, rimage_h]
        (waymo_dataset.py:334-336), memory-mapped; the normalisation of get_lidar happens on the device."""
        return np.load(self.data_path / sequence_name / ("%04d.npy" % sample_idx), mmap_mode="r")

    def get_seg_label(self, sequence_name, sample_idx):
        seg_file = str(self.data_path / sequence_name / ("%04d_seg.npy" % sample_idx))
        if not os.path.exists(seg_file):
            seg_file = seg_file.replace("_seg.npy", "_propseg.npy")
        return np.load(seg_file, mmap_mode="r")  # (N, 2): [instance_label, segmentation_label]

    def kept_rows(self, info, n_rows):
        """Row selection of DROP_POINTS_BY_LIDAR_INDEX (waymo_dataset.py:393-403). The reference takes
        rows [0, n_i) for every kept lidar i (its running offset never advances); mirrored as is.
        Returns a slice when that is one leading run, else an index array."""
        if self.drop_points_by_lidar_index is None:
            return slice(0, n_rows)
        kept = [n for i, n in enumerate(info["num_points_of_each_lidar"]) if i not in self.drop_points_by_lidar_index]
        if len(kept) == 1:
            return slice(0, int(kept[0]))
        return np.concatenate([np.arange(0, n) for n in kept], axis=0)

    def filter_rows(self, info, sel, n_rows):
        """FILTER_FOREGROUND then REMOVE_GROUND (waymo_dataset.py:479-492): both drop rows of the frame's points
        AFTER the lidar selection, by a per-frame file with one entry per remaining row -- predicted labels
        (`<PRED_LABEL_PATH>/<sequence>/<idx:03d>_pred.npy`, rows of the FILTER_CLASS classes go) and point heights
        (`<POINT_HEIGHT_PATH>/<sequence>/<idx:04d>_point_height.npy`, rows below TRUNCATE_GROUND_HEIGHT go; its
        length is that of the rows the foreground filter left). A host-side row selection like kept_rows."""
        if not (self.filter_foreground or self.remove_ground):
            return sel
        seq, idx = info["point_cloud"]["lidar_sequence"], info["point_cloud"]["sample_idx"]
        rows = np.arange(n_rows)[sel]
        if self.filter_foreground:
            pred = np.load(f"{self.pred_label_path}/{seq}/{idx:03d}_pred.npy").astype(np.int64)
            mask = np.ones(pred.shape[0], dtype=bool)
            for cls in self.filter_class:
                mask[pred == cls] = 0
            rows = rows[mask]
        if self.remove_ground:
            height = np.load(Path(self.dataset_cfg["POINT_HEIGHT_PATH"]) / seq / ("%04d_point_height.npy" % idx))
            rows = rows[~(height < self.dataset_cfg["TRUNCATE_GROUND_HEIGHT"])]
        return rows

    @staticmethod
    def frame_annos(info):
        """Boxes of one frame as load_data keeps them in test mode (waymo_dataset.py:418-468)."""
        if "annos" not in info:
            return None
        annos = info["annos"]
        for name in ("unknown", "Sign"):
            keep = [i for i, x in enumerate(annos["name"]) if x != name]
            annos = {k: (v[keep] if isinstance(v, np.ndarray) else [v[i] for i in keep])
                     for k, v in annos.items() if isinstance(v, (np.ndarray, list))}
        return annos

    def sweep_infos(self, index):
        """The sweeps of sample `index` in output order (waymo_dataset.py:506-575)."""
        if self._merge_all_iters_to_one_epoch:
            index = index % len(self.infos)
        info = self.infos[index]
        cur, seq = info["point_cloud"]["sample_idx"], info["point_cloud"]["lidar_sequence"]
        sweeps = [info]
        if self.num_sweeps > 1:
            for ci in range(cur + self.sweep_dir, cur + self.sweep_dir * self.num_sweeps, self.sweep_dir):
                if (seq, ci) not in self.info_pool:
                    continue
                other = self.info_pool[(seq, ci)]
                sweeps = [other] + sweeps if self.sweep_dir == -1 else sweeps + [other]
        return sweeps

    def sweep_transforms(self, sweeps):
        """Pose products of __getitem__ (waymo_dataset.py:576-585): T_s = inv(pose_last) @ pose_s, f64."""
        poses = [np.asarray(i["pose"], dtype=np.float64).reshape(4, 4) for i in sweeps]
        T0_inv = np.linalg.inv(poses[-1])
        return poses, [T0_inv @ T1 for T1 in poses]

    def sync_transforms(self, index, sweeps):
        """SYNC_MOVING_POINTS, the rigid motion of every box of every sweep onto the box of the same object in
        the sample's own frame (waymo_dataset.py:513-569): Kabsch on the eight corners (numpy SVD in f64, the
        reference's own calls), identity for the sample's frame, a 1e4 m translation -- the points are dropped
        afterwards -- for objects the sample's frame does not have. One [k_s, 4, 4] f64 array per sweep."""
        cur = self.infos[index]
        cur_annos = self.frame_annos(cur)
        cur_boxes = np.array(cur_annos["gt_boxes_lidar"], copy=True)
        box_corners = boxes_to_corners_3d(cur_boxes)
        obj_id_to_box = {obj_id: k for k, obj_id in enumerate(cur_annos["obj_ids"])}
        T0_inv = np.linalg.inv(np.asarray(cur["pose"], dtype=np.float64).reshape(4, 4))
        out = []
        for info in sweeps:
            annos = self.frame_annos(info)
            if info is cur or info["point_cloud"]["sample_idx"] == cur["point_cloud"]["sample_idx"]:
                k = len(annos["obj_ids"])
                out.append(np.stack([np.eye(4).astype(np.float64)] * k, axis=0) if k > 0
                           else np.zeros((0, 4, 4), dtype=np.float32))
                continue
            T_this = T0_inv @ np.asarray(info["pose"], dtype=np.float64).reshape(4, 4)
            corners = boxes_to_corners_3d(np.array(annos["gt_boxes_lidar"], copy=True))
            corners = (corners @ T_this[:3, :3].T + T_this[:3, 3]).reshape(-1, 8, 3)
            global_T = []
            for obj_idx, obj_id in enumerate(annos["obj_ids"]):
                T_t = np.eye(4).astype(np.float64)
                if obj_id not in obj_id_to_box:
                    T_t[:3, 3] = 1e4
                else:
                    last_box_corner = corners[obj_idx]
                    box_corner = box_corners[obj_id_to_box[obj_id]]
                    b0 = box_corner.mean(0)
                    l0 = last_box_corner.mean(0)
                    q = box_corner - b0
                    pp = last_box_corner - l0
                    M = pp.T @ q
                    U, S, VT = np.linalg.svd(M)
                    V = VT.T
                    sign = np.linalg.det(V @ U.T)
                    R = V @ np.diag([1, 1, sign]) @ U.T
                    T_t[:3, :3] = R
                    T_t[:3, 3] = b0 - R @ l0
                global_T.append(T_t)
            out.append(np.stack(global_T, axis=0) if global_T else np.zeros((0, 4, 4), dtype=np.float32))
        return out

    def assemble_objects(self, sweeps, Ts):
        """Per-sweep boxes into the last sweep's frame, padded and stacked (waymo_dataset.py:632-665):
        a few dozen boxes per sweep, on the host in the reference's own precision. Also returns every
        sweep's boxes in its OWN coordinates (for is_foreground)."""
        F = len(sweeps)
        object_dicts, box_list = [], []
        for info, T in zip(sweeps, Ts):
            annos = self.frame_annos(info)
            if annos is None:
                raise NotImplementedError("frames without annotations")
            boxes = np.array(annos["gt_boxes_lidar"], copy=True)
            box_list.append(boxes[:, :7].astype(np.float32))  # the frame's own coordinates: is_foreground
            corners = boxes_to_corners_3d(boxes)
            corners = (corners @ T[:3, :3].T + T[:3, 3]).reshape(-1, 24)
            boxes[:, :3] = boxes[:, :3] @ T[:3, :3].T + T[:3, 3]
            theta = boxes[:, -1]
            heading = np.stack([np.cos(theta), np.sin(theta), np.zeros_like(theta)], axis=-1) @ T[:3, :3].T
            heading[:, :2] = heading[:, :2] / np.linalg.norm(heading[:, :2], ord=2, axis=-1)[:, np.newaxis]
            boxes[:, -1] = np.arctan2(heading[:, 1], heading[:, 0])
            object_dicts.append(dict(
                gt_box_cls_label=annos["name"].astype(str), gt_box_attr=boxes,
                augmented=np.zeros(annos["name"].shape[0], dtype=bool), obj_ids=annos["obj_ids"],
                num_points_in_gt=annos["num_points_in_gt"], gt_box_corners_3d=corners))
        max_obj = max(d["gt_box_attr"].shape[0] for d in object_dicts)
        object_wise = {}
        for key in object_dicts[0]:
            padded = []
            for d in object_dicts:
                v = d[key]
                if v.shape[0] < max_obj:
                    v = np.concatenate([v, np.zeros((max_obj - v.shape[0], *v.shape[1:]), dtype=v.dtype)], axis=0)
                padded.append(v)
            stacked = np.stack(padded, axis=0)
            object_wise[key] = stacked.reshape(F * max_obj, *stacked.shape[2:])
        return object_wise, box_list

    # ------------------------------------------------------------------ one sample
    def load_sample_to_gpu(self, index, device="cuda", batch_index=0):
        """Sample `index` with the point-wise arrays as device tensors:
        point_wise = {point_bxyz [N,4] f32 (batch_index, aligned xyz), point_feat, point_sweep [N,1]
        int32, segmentation_label / instance_label [N] int32, is_foreground [N] bool};
        object_wise / scene_wise are the reference's host arrays (after prepare_data)."""
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("WaymoDataset assembles sweeps with the HIP library; a GPU device is required "
                               "(no CPU fallback)")
        sweeps = [copy.deepcopy(i) for i in self.sweep_infos(index)]
        F = len(sweeps)
        # ---- host: file rows of every sweep into one pinned buffer, one copy to HBM
        rows, raws, segs = [], [], []
        for info in sweeps:
            seq, idx = info["point_cloud"]["lidar_sequence"], info["point_cloud"]["sample_idx"]
            raw = self.read_lidar_file(seq, idx)
            sel = self.filter_rows(info, self.kept_rows(info, raw.shape[0]), raw.shape[0])
            raws.append((raw, sel))
            if self.load_seg:
                segs.append((self.get_seg_label(seq, idx), sel))
            rows.append(sel.stop - sel.start if isinstance(sel, slice) else sel.shape[0])
        offsets = np.zeros(F + 1, dtype=np.int64)
        np.cumsum(rows, out=offsets[1:])
        N = int(offsets[-1])
        host_raw = torch.empty((N, 8), dtype=torch.float32).pin_memory()
        host_seg = torch.empty((N, 2), dtype=torch.int32).pin_memory() if self.load_seg else None
        for s, (raw, sel) in enumerate(raws):
            host_raw.numpy()[offsets[s]:offsets[s + 1]] = raw[sel]
            if self.load_seg:
                host_seg.numpy()[offsets[s]:offsets[s + 1]] = segs[s][0][sel]
        d_raw = host_raw.to(device, non_blocking=True)
        d_seg = host_seg.to(device, non_blocking=True) if self.load_seg else None
        # ---- host: pose products in f64 (4x4, the reference's numpy calls)
        poses, Ts = self.sweep_transforms(sweeps)
        sweep_ids = np.array([int(i["frame_id"][-3:]) for i in sweeps], dtype=np.int32)
        d_T = torch.from_numpy(np.ascontiguousarray(np.stack([T[:3, :4] for T in Ts]))).to(device)
        d_off = torch.from_numpy(offsets).to(device)
        d_ids = torch.from_numpy(sweep_ids).to(device)
        object_wise, box_list = self.assemble_objects(sweeps, Ts)
        scene_wise = dict(frame_id=np.stack([np.asarray(i["frame_id"]) for i in sweeps], axis=0),
                          pose=np.stack(poses, axis=0))
        # ---- device: one pass over the rows; foreground flags against each sweep's own boxes
        bxyz, feat, sweep, seg_label, inst_label = ops.assemble_sweeps(d_raw, d_seg, d_off, d_T, d_ids, batch_index)
        if self.num_sweeps > 1 and self.with_time_feat:
            # waymo_dataset.py:625-630: the sweep's frame index / (sweeps of this sample - 1), formed in f64 like
            # numpy's int / int, in front of the features; the concatenation is f64 and rounded to f32 once
            t = sweep.reshape(-1, 1).to(torch.float64) / (F - 1)
            feat = torch.cat([t, feat.to(torch.float64)], dim=-1).to(torch.float32)
        point_wise = dict(point_bxyz=bxyz, point_feat=feat)
        if self.load_seg:
            point_wise["segmentation_label"] = seg_label
            point_wise["instance_label"] = inst_label
        box_counts = np.array([b.shape[0] for b in box_list], dtype=np.int32)
        box_off = np.zeros(F + 1, dtype=np.int32)
        np.cumsum(box_counts, out=box_off[1:])
        if box_off[-1] > 0 and N > 0:
            slot = torch.repeat_interleave(torch.arange(F, device=device), torch.from_numpy(np.asarray(rows)).to(device))
            _, first = ops.points_in_frame_boxes(
                d_raw[:, :3], slot, torch.from_numpy(np.concatenate(box_list, axis=0)).to(device),
                torch.from_numpy(box_off).to(device), int(box_counts.max()), want_member=False, want_first=True)
            point_wise["is_foreground"] = first >= 0
            if self.sync_moving_points and self.num_sweeps > 1:
                # waymo_dataset.py:587-618: a point inside a box of its own sweep (the FIRST such box) moves with
                # that box's global_T -- in f64, on the aligned coordinates as stored (f32) --, then every point of
                # a sweep that has boxes is kept only if all its coordinates are below 1e3. The sample's
                # coordinates are f64 from here on, as the reference's are (its concatenation of f32 and f64
                # sweeps is f64). Term by term, left to right, like numpy's small-matrix product.
                gT = np.concatenate([g.astype(np.float64).reshape(-1, 4, 4) for g in self.sync_transforms(index, sweeps)], axis=0)
                d_gT = torch.from_numpy(gT).to(device)
                has_boxes = torch.from_numpy(box_counts > 0).to(device)[slot]
                gidx = torch.from_numpy(box_off[:-1].astype(np.int64)).to(device)[slot] + first.long().clamp(min=0)
                Tp = d_gT[gidx.clamp(max=max(gT.shape[0] - 1, 0))]
                inside = (first >= 0) & has_boxes
                xyz = bxyz[:, 1:].to(torch.float64)
                x, y, z = xyz[:, 0], xyz[:, 1], xyz[:, 2]
                moved = torch.stack([Tp[:, i, 0] * x + Tp[:, i, 1] * y + Tp[:, i, 2] * z + Tp[:, i, 3] for i in range(3)], dim=1)
                xyz = torch.where(inside[:, None], moved, xyz)
                keep = ~has_boxes | (xyz < 1e3).all(-1)
                bxyz = torch.cat([bxyz[:, :1].to(torch.float64), xyz], dim=1)[keep]
                point_wise = {k: v[keep] for k, v in point_wise.items()}
                point_wise["point_bxyz"] = bxyz
                sweep = sweep[keep]
        else:
            point_wise["is_foreground"] = torch.zeros(N, dtype=torch.bool, device=device)
        point_wise["point_sweep"] = sweep
        data_dict = self.prepare_data(dict(point_wise=point_wise, object_wise=object_wise, scene_wise=scene_wise))
        data_dict["scene_wise"]["num_sweeps"] = F
        return data_dict

    def __getitem__(self, index):
        """The reference's sample (host numpy, `point_xyz` [N,3] without the batch column)."""
        sample = self.load_sample_to_gpu(index, "cuda", 0)
        pw = {k: v.cpu().numpy() for k, v in sample["point_wise"].items()}
        pw["point_xyz"] = np.ascontiguousarray(pw.pop("point_bxyz")[:, 1:])
        sample["point_wise"] = pw
        return sample

    def load_batch_to_gpu(self, indices, device="cuda"):
        """collate_batch of the samples `indices` with the point arrays never leaving the device: the
        dictionary `SimpleReg.forward` takes (what `load_data_to_gpu(collate_batch(...))` gives)."""
        from ...models import load_data_to_gpu
        samples = [self.load_sample_to_gpu(i, device, b) for b, i in enumerate(indices)]
        return load_data_to_gpu(self.collate_batch(samples), device)
